/* rtr_post.h — launchers of the post passes that follow the ray-gen dispatch (internal to librtr_hip.so). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtrdev {

/* One a-trous step: reference src/shaders/denoise.comp:36-116 (5x5 taps spaced step_width apart, colour / normal /
 * position edge-stopping weights), applied to the two sampled images the reference dispatches one after the other
 * (unshadowed, then shadowed; application.cppm:399-432) in one fused pass.  All images are RGBA8, width x height. */
hipError_t launch_denoise_pair(const uint32_t* inA, uint32_t* outA, const uint32_t* inB, uint32_t* outB, const uint32_t* normal,
                               const uint32_t* position, uint32_t width, uint32_t height, int step_width, float c_phi, float n_phi,
                               float p_phi, hipStream_t stream);

/* reference src/shaders/combine.comp:20-37: final = analytic * shadowed / max(unshadowed, 0.001) */
hipError_t launch_combine(const uint32_t* analytic, const uint32_t* shadowed, const uint32_t* unshadowed, uint32_t* finalImage,
                          uint32_t width, uint32_t height, hipStream_t stream);

}  // namespace rtrdev
