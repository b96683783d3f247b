/* rtr_post.h — launchers of the post passes that follow the ray-gen dispatch (internal to librtr_hip.so). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtrdev {

/* One a-trous pass: reference src/shaders/denoise.comp:36-116 (5x5 taps spaced step_width apart, colour / normal /
 * position edge-stopping weights).  in/out/normal/position are RGBA8 images of width x height. */
hipError_t launch_denoise(const uint32_t* in, uint32_t* out, const uint32_t* normal, const uint32_t* position,
                          uint32_t width, uint32_t height, int step_width, float c_phi, float n_phi, float p_phi,
                          hipStream_t stream);

/* reference src/shaders/combine.comp:20-37: final = analytic * shadowed / max(unshadowed, 0.001) */
hipError_t launch_combine(const uint32_t* analytic, const uint32_t* shadowed, const uint32_t* unshadowed, uint32_t* finalImage,
                          uint32_t width, uint32_t height, hipStream_t stream);

}  // namespace rtrdev
