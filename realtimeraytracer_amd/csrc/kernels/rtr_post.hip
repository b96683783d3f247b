/* rtr_post.hip — a-trous denoise + combine as gfx950 kernels.
 * Replaces reference src/shaders/denoise.comp and combine.comp, dispatched 8x + 1x per frame by
 * src/app/application.cppm:391-445.
 *
 * Launch shape: 32x8-pixel workgroups (256 threads, one wave per 32x2 strip... rows are contiguous so the 4-byte
 * loads and stores of a wave are 128-B segments).  The three source images of one pass total 25 MB at 1080p and the
 * 25 taps of neighbouring pixels overlap almost completely, so the taps are served by L1/L2; the centre pixel's
 * colour / normal / position stay in registers.  Arithmetic goes through include/rtr_math.h (exp = exp2(x*log2 e)),
 * in the same order as oracle/oracle_post.cpp, so the UNORM8 outputs are bit-identical to the oracle's.
 * Quirks kept verbatim: Q9 (kernel weight index advances only for in-bounds taps), Q10 (G-buffers are UNORM8).
 */
#include "rtr_post.h"
#include "../../../include/rtr_math.h"

namespace rtrdev {

struct V4 { float x, y, z, w; };

__device__ __forceinline__ V4 load_unorm(uint32_t p) {
    V4 r;
    r.x = (float)(p & 0xffu) / 255.0f;
    r.y = (float)((p >> 8) & 0xffu) / 255.0f;
    r.z = (float)((p >> 16) & 0xffu) / 255.0f;
    r.w = (float)((p >> 24) & 0xffu) / 255.0f;
    return r;
}
__device__ __forceinline__ uint32_t store_unorm(V4 v) {
    return rtr_unorm8(v.x) | (rtr_unorm8(v.y) << 8) | (rtr_unorm8(v.z) << 16) | (rtr_unorm8(v.w) << 24);
}
__device__ __forceinline__ V4 sub4(V4 a, V4 b) { V4 r = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; return r; }
__device__ __forceinline__ float dot4(V4 a, V4 b) { return rtr_fma(a.w, b.w, rtr_fma(a.z, b.z, rtr_fma(a.y, b.y, a.x * b.x))); }
__device__ __forceinline__ float exp_f(float x) { return rtr_exp2(x * 1.4426950408889634f); }

__constant__ float c_kernel[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};

__global__ __launch_bounds__(256) void k_denoise(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                 const uint32_t* __restrict__ normalImg, const uint32_t* __restrict__ positionImg,
                                                 int W, int H, int step_width, float c_phi, float n_phi, float p_phi) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const V4 color = load_unorm(in[p]);
    const V4 normal = load_unorm(normalImg[p]);
    const V4 position = load_unorm(positionImg[p]);
    const float inv_step2 = (float)(step_width * step_width);
    float cum_weight = 0.0f;
    V4 sum = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int ox = x + dx * step_width, oy = y + dy * step_width;
            if (ox < 0 || oy < 0 || ox >= W || oy >= H) continue;
            const size_t q = (size_t)oy * W + ox;
            const V4 color_tmp = load_unorm(in[q]);
            const V4 normal_tmp = load_unorm(normalImg[q]);
            const V4 position_tmp = load_unorm(positionImg[q]);
            V4 t = sub4(color, color_tmp);
            float dist2 = dot4(t, t);
            const float color_weight = rtr_min(exp_f(-(dist2) / c_phi), 1.0f);
            t = sub4(normal, normal_tmp);
            dist2 = rtr_max(dot4(t, t) / inv_step2, 0.0f);
            const float normal_weight = rtr_min(exp_f(-(dist2) / n_phi), 1.0f);
            t = sub4(position, position_tmp);
            dist2 = dot4(t, t);
            const float pos_weight = rtr_min(exp_f(-(dist2) / p_phi), 1.0f);
            const float weight = color_weight * normal_weight * pos_weight * c_kernel[k];
            cum_weight += weight;
            sum.x = rtr_fma(color_tmp.x, weight, sum.x); sum.y = rtr_fma(color_tmp.y, weight, sum.y);
            sum.z = rtr_fma(color_tmp.z, weight, sum.z); sum.w = rtr_fma(color_tmp.w, weight, sum.w);
            ++k;
        }
    }
    const float d = rtr_max(cum_weight, 1e-5f);
    sum.x /= d; sum.y /= d; sum.z /= d; sum.w /= d;
    out[p] = store_unorm(sum);
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

hipError_t launch_denoise(const uint32_t* in, uint32_t* out, const uint32_t* normal, const uint32_t* position,
                          uint32_t width, uint32_t height, int step_width, float c_phi, float n_phi, float p_phi, hipStream_t stream) {
    dim3 grid((width + 31u) / 32u, (height + 7u) / 8u);
    hipLaunchKernelGGL(k_denoise, grid, dim3(256), 0, stream, in, out, normal, position, (int)width, (int)height, step_width, c_phi, n_phi, p_phi);
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
