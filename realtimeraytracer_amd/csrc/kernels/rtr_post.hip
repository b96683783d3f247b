/* rtr_post.hip — a-trous denoise + combine as gfx950 kernels.
 * Replaces reference src/shaders/denoise.comp and combine.comp, dispatched 8x + 1x per frame by
 * src/app/application.cppm:391-445.
 *
 * Launch shape: 32x8-pixel workgroups (256 threads; rows are contiguous so the 4-byte loads and stores of a wave are
 * 128-B segments).  The four source images of one pass total 33 MB at 1080p and the 25 taps of neighbouring pixels
 * overlap almost completely, so the taps are served by L1/L2; the centre pixel's values stay in registers.  The pass
 * is bound by vector arithmetic, not by memory: what made it 2.5x faster (2.90 -> 1.15 ms for the reference's 8 + 1
 * dispatches at 1080p, profiles/time_denoise.py) was removing instructions — UNORM8 unpacking without the IEEE division
 * sequence (rtr_unorm8_to_float, exact) and sharing the normal / position weights between the two images.
 * Arithmetic goes through include/rtr_math.h (exp = exp2(x*log2 e)), in the same order as oracle/oracle_post.cpp, so the
 * UNORM8 outputs are bit-identical to the oracle's.
 * Quirks kept verbatim: Q9 (kernel weight index advances only for in-bounds taps), Q10 (G-buffers are UNORM8).
 */
#include "rtr_post.h"
#include "../../../include/rtr_math.h"

namespace rtrdev {

struct V4 { float x, y, z, w; };

/* UNORM8 -> float through rtr_unorm8_to_float (exactly b / 255.0f, three vector instructions instead of a division) */
__device__ __forceinline__ V4 load_unorm(uint32_t p) {
    V4 r;
    r.x = rtr_unorm8_to_float(p & 0xffu);
    r.y = rtr_unorm8_to_float((p >> 8) & 0xffu);
    r.z = rtr_unorm8_to_float((p >> 16) & 0xffu);
    r.w = rtr_unorm8_to_float(p >> 24);
    return r;
}
__device__ __forceinline__ uint32_t store_unorm(V4 v) {
    return rtr_unorm8(v.x) | (rtr_unorm8(v.y) << 8) | (rtr_unorm8(v.z) << 16) | (rtr_unorm8(v.w) << 24);
}
__device__ __forceinline__ V4 sub4(V4 a, V4 b) { V4 r = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; return r; }
__device__ __forceinline__ float dot4(V4 a, V4 b) { return rtr_fma(a.w, b.w, rtr_fma(a.z, b.z, rtr_fma(a.y, b.y, a.x * b.x))); }
__device__ __forceinline__ float exp_f(float x) { return rtr_exp2(x * 1.4426950408889634f); }

__constant__ float c_kernel[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};

/* One a-trous step (denoise.comp:36-116) over BOTH sampled images in one pass: the reference dispatches the unshadowed
 * and the shadowed image separately (application.cppm:399-432), but they share the taps' normal and position weights
 * (two of the three exponentials per tap) and the unpacking of those two G-buffers, so the pair costs ~45 % of two
 * separate launches.  Per image the arithmetic and its order are exactly those of the shader / the oracle. */
__global__ __launch_bounds__(256) void k_denoise_pair(const uint32_t* __restrict__ inA, uint32_t* __restrict__ outA,
                                                      const uint32_t* __restrict__ inB, uint32_t* __restrict__ outB,
                                                      const uint32_t* __restrict__ normalImg, const uint32_t* __restrict__ positionImg,
                                                      int W, int H, int step_width, float c_phi, float n_phi, float p_phi) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const V4 colorA = load_unorm(inA[p]), colorB = load_unorm(inB[p]);
    const V4 normal = load_unorm(normalImg[p]);
    const V4 position = load_unorm(positionImg[p]);
    const float inv_step2 = (float)(step_width * step_width);
    float cumA = 0.0f, cumB = 0.0f;
    V4 sumA = {0.f, 0.f, 0.f, 0.f}, sumB = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int ox = x + dx * step_width, oy = y + dy * step_width;
            if (ox < 0 || oy < 0 || ox >= W || oy >= H) continue;
            const size_t q = (size_t)oy * W + ox;
            const V4 a_tmp = load_unorm(inA[q]), b_tmp = load_unorm(inB[q]);
            const V4 normal_tmp = load_unorm(normalImg[q]);
            const V4 position_tmp = load_unorm(positionImg[q]);
            V4 t = sub4(normal, normal_tmp);
            float dist2 = rtr_max(dot4(t, t) / inv_step2, 0.0f);
            const float normal_weight = rtr_min(exp_f(-(dist2) / n_phi), 1.0f);
            t = sub4(position, position_tmp);
            dist2 = dot4(t, t);
            const float pos_weight = rtr_min(exp_f(-(dist2) / p_phi), 1.0f);
            t = sub4(colorA, a_tmp);
            dist2 = dot4(t, t);
            const float wA = rtr_min(exp_f(-(dist2) / c_phi), 1.0f) * normal_weight * pos_weight * c_kernel[k];
            t = sub4(colorB, b_tmp);
            dist2 = dot4(t, t);
            const float wB = rtr_min(exp_f(-(dist2) / c_phi), 1.0f) * normal_weight * pos_weight * c_kernel[k];
            cumA += wA; cumB += wB;
            sumA.x = rtr_fma(a_tmp.x, wA, sumA.x); sumA.y = rtr_fma(a_tmp.y, wA, sumA.y);
            sumA.z = rtr_fma(a_tmp.z, wA, sumA.z); sumA.w = rtr_fma(a_tmp.w, wA, sumA.w);
            sumB.x = rtr_fma(b_tmp.x, wB, sumB.x); sumB.y = rtr_fma(b_tmp.y, wB, sumB.y);
            sumB.z = rtr_fma(b_tmp.z, wB, sumB.z); sumB.w = rtr_fma(b_tmp.w, wB, sumB.w);
            ++k;
        }
    }
    const float dA = rtr_max(cumA, 1e-5f), dB = rtr_max(cumB, 1e-5f);
    sumA.x /= dA; sumA.y /= dA; sumA.z /= dA; sumA.w /= dA;
    sumB.x /= dB; sumB.y /= dB; sumB.z /= dB; sumB.w /= dB;
    outA[p] = store_unorm(sumA);
    outB[p] = store_unorm(sumB);
}

__global__ __launch_bounds__(256) void k_combine(const uint32_t* __restrict__ analytic, const uint32_t* __restrict__ shadowed,
                                                 const uint32_t* __restrict__ unshadowed, uint32_t* __restrict__ finalImage, size_t n) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) {
        const V4 a = load_unorm(analytic[p]), s = load_unorm(shadowed[p]), u = load_unorm(unshadowed[p]);
        V4 f;
        f.x = a.x * (s.x / rtr_max(u.x, 0.001f));
        f.y = a.y * (s.y / rtr_max(u.y, 0.001f));
        f.z = a.z * (s.z / rtr_max(u.z, 0.001f));
        f.w = 1.0f;
        finalImage[p] = store_unorm(f);
    }
}

hipError_t launch_denoise_pair(const uint32_t* inA, uint32_t* outA, const uint32_t* inB, uint32_t* outB, const uint32_t* normal,
                               const uint32_t* position, uint32_t width, uint32_t height, int step_width, float c_phi, float n_phi,
                               float p_phi, hipStream_t stream) {
    dim3 grid((width + 31u) / 32u, (height + 7u) / 8u);
    hipLaunchKernelGGL(k_denoise_pair, grid, dim3(256), 0, stream, inA, outA, inB, outB, normal, position, (int)width, (int)height,
                       step_width, c_phi, n_phi, p_phi);
    return hipGetLastError();
}

hipError_t launch_combine(const uint32_t* analytic, const uint32_t* shadowed, const uint32_t* unshadowed, uint32_t* finalImage,
                          uint32_t width, uint32_t height, hipStream_t stream) {
    const size_t n = (size_t)width * height;
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(256), 0, stream, analytic, shadowed, unshadowed, finalImage, n);
    return hipGetLastError();
}

}  // namespace rtrdev
