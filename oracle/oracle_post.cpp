/* oracle_post.cpp — CPU ORACLE (test infrastructure) for the passes that follow the ray-gen dispatch:
 * the à-trous denoiser and the combine pass.  Restates reference src/shaders/denoise.comp:36-116 and
 * src/shaders/combine.comp:20-37 with the host protocol of src/app/application.cppm:391-445
 * (NUM_DENOISING_ITERATIONS=4, step=(i+1)*DENOISING_STRENGTH, c_phi=1, n_phi=p_phi=1e-3, ping-pong
 * flag starting at 1, unshadowed pass then shadowed pass per iteration).  Quirks Q8 (wasted 4th pass),
 * Q9 (kernel index k only advances for in-bounds taps) and Q10 (UNORM8 G-buffers) are kept verbatim.
 * exp() is rtr_exp2(x*log2(e)) from the shared numerical contract. */
#include "oracle.h"
#include "../include/rtr_math.h"

#include <vector>

namespace {

struct V4 { float x, y, z, w; };

inline V4 load_unorm(uint32_t p) {                 /* imageLoad on rgba8 */
    V4 r;
    r.x = (float)(p & 0xffu) / 255.0f;
    r.y = (float)((p >> 8) & 0xffu) / 255.0f;
    r.z = (float)((p >> 16) & 0xffu) / 255.0f;
    r.w = (float)((p >> 24) & 0xffu) / 255.0f;
    return r;
}
inline uint32_t store_unorm(V4 v) {                /* imageStore on rgba8 */
    return rtr_unorm8(v.x) | (rtr_unorm8(v.y) << 8) | (rtr_unorm8(v.z) << 16) | (rtr_unorm8(v.w) << 24);
}
inline V4 sub4(V4 a, V4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline float dot4(V4 a, V4 b) { return rtr_fma(a.w, b.w, rtr_fma(a.z, b.z, rtr_fma(a.y, b.y, a.x * b.x))); }
inline float exp_f(float x) { return rtr_exp2(x * 1.4426950408889634f); }

const float kKernel[25] = {1, 4, 7, 4, 1, 4, 16, 26, 16, 4, 7, 26, 41, 26, 7, 4, 16, 26, 16, 4, 1, 4, 7, 4, 1};  /* denoise.comp:28-34 */

void denoise_pass(int W, int H, const uint32_t* in, uint32_t* out, const uint32_t* normalImg, const uint32_t* positionImg,
                  int step_width, float c_phi, float n_phi, float p_phi) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            const V4 color = load_unorm(in[p]);                                           /* :61-63 */
            const V4 normal = load_unorm(normalImg[p]);
            const V4 position = load_unorm(positionImg[p]);
            float cum_weight = 0.0f;
            V4 sum = {0, 0, 0, 0};
            int k = 0;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int ox = x + dx * step_width, oy = y + dy * step_width;         /* :70 */
                    if (ox < 0 || oy < 0 || ox >= W || oy >= H) continue;                 /* :72-73 */
                    const size_t q = (size_t)oy * W + ox;
                    const V4 color_tmp = load_unorm(in[q]);
                    const V4 normal_tmp = load_unorm(normalImg[q]);
                    const V4 position_tmp = load_unorm(positionImg[q]);
                    V4 t = sub4(color, color_tmp);                                        /* :79-81 */
                    float dist2 = dot4(t, t);
                    const float color_weight = rtr_min(exp_f(-(dist2) / c_phi), 1.0f);
                    t = sub4(normal, normal_tmp);                                         /* :83-85 */
                    dist2 = rtr_max(dot4(t, t) / (float)(step_width * step_width), 0.0f);
                    const float normal_weight = rtr_min(exp_f(-(dist2) / n_phi), 1.0f);
                    t = sub4(position, position_tmp);                                     /* :87-90 (POSITION_SCALE = 1) */
                    dist2 = dot4(t, t);
                    const float pos_weight = rtr_min(exp_f(-(dist2) / p_phi), 1.0f);
                    const float weight = color_weight * normal_weight * pos_weight * kKernel[k];   /* :92 */
                    cum_weight += weight;
                    sum.x = rtr_fma(color_tmp.x, weight, sum.x); sum.y = rtr_fma(color_tmp.y, weight, sum.y);
                    sum.z = rtr_fma(color_tmp.z, weight, sum.z); sum.w = rtr_fma(color_tmp.w, weight, sum.w);
                    ++k;                                                                  /* :96 (Q9) */
                }
            const float d = rtr_max(cum_weight, 1e-5f);                                   /* :100 */
            sum.x /= d; sum.y /= d; sum.z /= d; sum.w /= d;
            out[p] = store_unorm(sum);
        }
}

}  // namespace

extern "C" int oracle_denoise_combine(uint32_t width, uint32_t height, const uint32_t* analytic, uint32_t* shadowed,
                                      uint32_t* unshadowed, const uint32_t* normal, const uint32_t* position,
                                      uint32_t* denoisedShadowed, uint32_t* denoisedUnshadowed, uint32_t* finalImage,
                                      int iterations) {
    if (!analytic || !shadowed || !unshadowed || !normal || !position || !denoisedShadowed || !denoisedUnshadowed || !finalImage)
        return -1;
    const int W = (int)width, H = (int)height;
    int denoisingOutput = 1;                                                              /* application.cppm:392 */
    for (int i = 0; i < iterations; ++i) {                                                /* :395-434 */
        const int step = (i + 1) * 1;
        if (denoisingOutput == 1) {
            denoise_pass(W, H, unshadowed, denoisedUnshadowed, normal, position, step, 1.0f, 0.001f, 0.001f);
            denoise_pass(W, H, shadowed, denoisedShadowed, normal, position, step, 1.0f, 0.001f, 0.001f);
        } else {
            denoise_pass(W, H, denoisedUnshadowed, unshadowed, normal, position, step, 1.0f, 0.001f, 0.001f);
            denoise_pass(W, H, denoisedShadowed, shadowed, normal, position, step, 1.0f, 0.001f, 0.001f);
        }
        denoisingOutput = 1 - denoisingOutput;
    }
    /* combine.comp:20-37 with the flag as left by the loop (:444) */
    const uint32_t* sh = denoisingOutput == 0 ? shadowed : denoisedShadowed;
    const uint32_t* un = denoisingOutput == 0 ? unshadowed : denoisedUnshadowed;
    for (size_t p = 0; p < (size_t)W * H; ++p) {
        const V4 a = load_unorm(analytic[p]), s = load_unorm(sh[p]), u = load_unorm(un[p]);
        V4 f;
        f.x = a.x * (s.x / rtr_max(u.x, 0.001f));
        f.y = a.y * (s.y / rtr_max(u.y, 0.001f));
        f.z = a.z * (s.z / rtr_max(u.z, 0.001f));
        f.w = 1.0f;
        finalImage[p] = store_unorm(f);
    }
    return 0;
}
