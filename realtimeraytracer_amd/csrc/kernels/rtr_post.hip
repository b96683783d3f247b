/* rtr_post.hip — a-trous denoise + combine as gfx950 kernels.
 * Replaces reference src/shaders/denoise.comp and combine.comp, dispatched 8x + 1x per frame by
 * src/app/application.cppm:391-445.
 *
 * Launch shape: 32x8-pixel workgroups (256 threads; rows are contiguous so the 4-byte loads and stores of a wave are
 * 128-B segments).  The four source images of one pass total 33 MB at 1080p and the 25 taps of neighbouring pixels
 * overlap almost completely, so the taps are served by L1/L2; the centre pixel's values stay in registers.  The pass
 * is bound by vector arithmetic, not by memory: what made it faster (2.90 -> 1.15 ms for the reference's 8 + 1 dispatches at 1080p
 * in round 2, profiles/time_denoise.py; see k_denoise_pair for round 4's step) was removing instructions — UNORM8 unpacking without
 * the IEEE division sequence (rtr_unorm8_to_float, exact) and sharing the normal / position weights between the two images.
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
/* the interior form: every tap is in bounds, so Q9's index is the tap's number (wave-uniform: a scalar load) */
__device__ __forceinline__ float kernel_weight(int i) { return c_kernel[i]; }

/* min(exp(x), 1) for x <= 0 (x = -(a squared distance) / phi): rtr_exp2 without the branch that cannot be taken (z > 127) and with
 * the underflow case selected at the end; the same operations in the same order otherwise, so the same bits. */
__device__ __forceinline__ float weight_exp(float x) {
    const float z = x * 1.4426950408889634f;
    const float nf = __builtin_floorf(z + 0.5f);
    const float r = (z - nf) * 0.6931471805599453f;
    float p = 2.48015873015873e-05f;
    p = rtr_fma(p, r, 1.984126984126984e-04f);
    p = rtr_fma(p, r, 1.388888888888889e-03f);
    p = rtr_fma(p, r, 8.333333333333333e-03f);
    p = rtr_fma(p, r, 4.1666666666666664e-02f);
    p = rtr_fma(p, r, 0.16666666666666666f);
    p = rtr_fma(p, r, 0.5f);
    p = rtr_fma(p, r, 1.0f);
    p = rtr_fma(p, r, 1.0f);
    const float e = p * rtr_u2f((uint32_t)((int32_t)nf + 127) << 23);
    return rtr_hwmin(z < -126.0f ? 0.0f : e, 1.0f);
}

struct DenoiseArgs {
    const uint32_t *inA, *inB, *normalImg, *positionImg;
    uint32_t *outA, *outB;
    int W, H, step_width;
    float step2, r_step2;           /* step_width^2 and fl(1 / it) */
    float c_phi, r_c, n_phi, r_n, p_phi, r_p;
};

/* One a-trous step (denoise.comp:36-116) over BOTH sampled images in one pass: the reference dispatches the unshadowed
 * and the shadowed image separately (application.cppm:399-432), but they share the taps' normal and position weights
 * (two of the three exponentials per tap) and the unpacking of those two G-buffers, so the pair costs ~45 % of two
 * separate launches.  Per image the arithmetic and its order are exactly those of the shader / the oracle.
 * The pass is bound by vector issue, and on gfx950 an instruction costs 2 cycles (plain fp32 fma / mul / add) or 4 (everything else:
 * profiles/r04/inst_rates.json), so what is kept out of the tap loop is divisions (~34 cycles each: the divisors are launch
 * constants -> rtr_div_by, five 2-cycle operations, exact; a divisor of 1 is skipped), the compare + select pairs of rtr_min /
 * rtr_max (hardware min; the max against 0 of a sum of squares is the identity), the UNORM8 unpack's third operation, and — INTERIOR:
 * a workgroup whose every tap is inside the image — the bounds tests and the per-lane index into the 5x5 weights.
 * 1.15 -> 0.90 ms for the reference's 8 + 1 dispatches at 1080p (profiles/r04/time_denoise_r04.log, ab_denoise.log); outputs unchanged. */
template <bool INTERIOR, bool CPHI_ONE>
__device__ __forceinline__ void denoise_pixel(const DenoiseArgs& g, const int x, const int y) {
    const size_t p = (size_t)y * g.W + x;
    const V4 colorA = load_unorm(g.inA[p]), colorB = load_unorm(g.inB[p]);
    const V4 normal = load_unorm(g.normalImg[p]);
    const V4 position = load_unorm(g.positionImg[p]);
    const bool step_one = g.step2 == 1.0f;
    float cumA = 0.0f, cumB = 0.0f;
    V4 sumA = {0.f, 0.f, 0.f, 0.f}, sumB = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    /* the rows of taps stay a loop: with all 25 taps unrolled the compiler hoists their hundred loads and the kernel takes 248 VGPRs
     * (two waves per SIMD); five taps at a time it takes 117 (four waves) and is 7 % faster; capping it further spills
     * (profiles/r04/ab_denoise.log) */
#pragma unroll 1
    for (int dy = -2; dy <= 2; ++dy) {
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int ox = x + dx * g.step_width, oy = y + dy * g.step_width;
            if (!INTERIOR && (ox < 0 || oy < 0 || ox >= g.W || oy >= g.H)) continue;
            const size_t q = (size_t)oy * g.W + ox;
            const V4 a_tmp = load_unorm(g.inA[q]), b_tmp = load_unorm(g.inB[q]);
            const V4 normal_tmp = load_unorm(g.normalImg[q]);
            const V4 position_tmp = load_unorm(g.positionImg[q]);
            V4 t = sub4(normal, normal_tmp);
            float dist2 = dot4(t, t);                                       /* >= +0: the shader's max(.., 0) is the identity */
            if (!step_one) dist2 = rtr_div_by(dist2, g.step2, g.r_step2);
            const float normal_weight = weight_exp(rtr_div_by(-dist2, g.n_phi, g.r_n));
            t = sub4(position, position_tmp);
            const float pos_weight = weight_exp(rtr_div_by(-dot4(t, t), g.p_phi, g.r_p));
            const float kw = INTERIOR ? kernel_weight((dy + 2) * 5 + dx + 2) : c_kernel[k];
            t = sub4(colorA, a_tmp);
            dist2 = dot4(t, t);
            const float wA = weight_exp(CPHI_ONE ? -dist2 : rtr_div_by(-dist2, g.c_phi, g.r_c)) * normal_weight * pos_weight * kw;
            t = sub4(colorB, b_tmp);
            dist2 = dot4(t, t);
            const float wB = weight_exp(CPHI_ONE ? -dist2 : rtr_div_by(-dist2, g.c_phi, g.r_c)) * normal_weight * pos_weight * kw;
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
    g.outA[p] = store_unorm(sumA);
    g.outB[p] = store_unorm(sumB);
}

template <bool CPHI_ONE>
__global__ __launch_bounds__(256) void k_denoise_pair(const DenoiseArgs g) {
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8, reach = 2 * g.step_width;
    const int x = x0 + (threadIdx.x & 31), y = y0 + (threadIdx.x >> 5);
    if (x0 >= reach && y0 >= reach && x0 + 31 + reach < g.W && y0 + 7 + reach < g.H) { denoise_pixel<true, CPHI_ONE>(g, x, y); return; }
    if (x >= g.W || y >= g.H) return;
    denoise_pixel<false, CPHI_ONE>(g, x, y);
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
    DenoiseArgs g;
    g.inA = inA; g.inB = inB; g.normalImg = normal; g.positionImg = position; g.outA = outA; g.outB = outB;
    g.W = (int)width; g.H = (int)height; g.step_width = step_width;
    g.step2 = (float)(step_width * step_width); g.r_step2 = 1.0f / g.step2;          /* host divisions: IEEE, correctly rounded */
    g.c_phi = c_phi; g.r_c = 1.0f / c_phi; g.n_phi = n_phi; g.r_n = 1.0f / n_phi; g.p_phi = p_phi; g.r_p = 1.0f / p_phi;
    if (c_phi == 1.0f) hipLaunchKernelGGL((k_denoise_pair<true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((k_denoise_pair<false>), grid, dim3(256), 0, stream, g);
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
