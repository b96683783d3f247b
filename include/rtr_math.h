/* rtr_math.h — the numerical contract of the renderer.
 *
 * Every float operation on the hot path is spelled here in terms of IEEE-754 binary32
 * + - * / sqrt and explicit fma(), each of which is correctly rounded on x86-64 and on
 * gfx950 (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  Both compilers are
 * run with -ffp-contract=off and without fast-math, so an expression means the same bits
 * on the host cores and on the GPU.  That is what makes "RGBA8 framebuffer bit-exact
 * between the HIP path and the CPU oracle" a testable statement.
 *
 * pow/exp2/log2 are our own polynomial forms (libm and ocml differ by ULPs, and
 * pow(x,1/2.2) feeds the 8-bit quantiser); they are pinned against libm in
 * tests/test_math.py with a stated ULP tolerance.
 *
 * Used by: the HIP kernels (device), the host scene layer, and the CPU oracle.
 */
#ifndef RTR_MATH_H
#define RTR_MATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RTR_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define RTR_HD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RTR_DEVICE_CODE 1
#else
#define RTR_DEVICE_CODE 0
#endif

typedef struct rtr_v3 { float x, y, z; } rtr_v3;

/* ---- scalar helpers ---------------------------------------------------------------- */
RTR_HD float rtr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RTR_HD float rtr_sqrt(float a) { return __builtin_sqrtf(a); }
RTR_HD float rtr_abs(float a) { return __builtin_fabsf(a); }

/* select-form min/max/clamp: identical on every target, NaN in 'a' yields 'b' */
RTR_HD float rtr_min(float a, float b) { return a < b ? a : b; }
RTR_HD float rtr_max(float a, float b) { return a > b ? a : b; }
RTR_HD float rtr_clamp(float x, float lo, float hi) { return rtr_min(rtr_max(x, lo), hi); }

/* hardware min/max: only for NaN-free operands whose zero sign is never observed
 * (the slab test).  On the device these are single v_min/v_max instructions. */
RTR_HD float rtr_hwmin(float a, float b) {
#if RTR_DEVICE_CODE
    return __builtin_fminf(a, b);
#else
    return a < b ? a : b;
#endif
}
RTR_HD float rtr_hwmax(float a, float b) {
#if RTR_DEVICE_CODE
    return __builtin_fmaxf(a, b);
#else
    return a > b ? a : b;
#endif
}

RTR_HD uint32_t rtr_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RTR_HD float    rtr_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---- PCG hash: reference src/shaders/raycommon.glsl:22-27 ------------------------------ */
RTR_HD uint32_t rtr_pcg_hash(uint32_t seed) {
    uint32_t state = seed * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
/* float(hash) / 2^32: one round-to-nearest-even convert, then an exact scaling.
 * May return exactly 1.0 (hash >= 0xFFFFFF80), as the reference does (quirk Q2). */
RTR_HD float rtr_random(uint32_t seed) {
    return (float)rtr_pcg_hash(seed) * 2.3283064365386963e-10f;
}

/* ---- vec3 ------------------------------------------------------------------------------ */
RTR_HD rtr_v3 rtr_mk(float x, float y, float z) { rtr_v3 r; r.x = x; r.y = y; r.z = z; return r; }
RTR_HD rtr_v3 rtr_ld3(const float* p) { return rtr_mk(p[0], p[1], p[2]); }
RTR_HD rtr_v3 rtr_add(rtr_v3 a, rtr_v3 b) { return rtr_mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTR_HD rtr_v3 rtr_sub(rtr_v3 a, rtr_v3 b) { return rtr_mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTR_HD rtr_v3 rtr_mul(rtr_v3 a, rtr_v3 b) { return rtr_mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RTR_HD rtr_v3 rtr_scale(rtr_v3 a, float s) { return rtr_mk(a.x * s, a.y * s, a.z * s); }
RTR_HD rtr_v3 rtr_neg(rtr_v3 a) { return rtr_mk(-a.x, -a.y, -a.z); }
/* a + b*s, one fma per component */
RTR_HD rtr_v3 rtr_madd(rtr_v3 a, rtr_v3 b, float s) {
    return rtr_mk(rtr_fma(b.x, s, a.x), rtr_fma(b.y, s, a.y), rtr_fma(b.z, s, a.z));
}
/* dot = fma(az,bz, fma(ay,by, ax*bx)) */
RTR_HD float rtr_dot(rtr_v3 a, rtr_v3 b) {
    return rtr_fma(a.z, b.z, rtr_fma(a.y, b.y, a.x * b.x));
}
/* cross: each component fma(p, q, -(r*s)) */
RTR_HD rtr_v3 rtr_cross(rtr_v3 a, rtr_v3 b) {
    return rtr_mk(rtr_fma(a.y, b.z, -(a.z * b.y)),
                  rtr_fma(a.z, b.x, -(a.x * b.z)),
                  rtr_fma(a.x, b.y, -(a.y * b.x)));
}
RTR_HD float rtr_length(rtr_v3 a) { return rtr_sqrt(rtr_dot(a, a)); }
/* GLSL normalize(): v * (1/sqrt(dot)); normalize(0) = NaN exactly as in GLSL (0 * inf). */
RTR_HD rtr_v3 rtr_normalize(rtr_v3 a) {
    float inv = 1.0f / rtr_sqrt(rtr_dot(a, a));
    return rtr_scale(a, inv);
}

/* ---- transforms ------------------------------------------------------------------------ */
/* row-major 3x4 (vk::TransformMatrixKHR / gl_ObjectToWorldEXT) times (p,1) */
RTR_HD rtr_v3 rtr_xform_point34(const float* m, rtr_v3 p) {
    return rtr_mk(rtr_fma(m[2],  p.z, rtr_fma(m[1], p.y, m[0] * p.x)) + m[3],
                  rtr_fma(m[6],  p.z, rtr_fma(m[5], p.y, m[4] * p.x)) + m[7],
                  rtr_fma(m[10], p.z, rtr_fma(m[9], p.y, m[8] * p.x)) + m[11]);
}
/* column-major mat4 (GLSL `mat4 * vec4(p,1)`, raygen.rgen:187-189) */
RTR_HD rtr_v3 rtr_xform_point44cm(const float* m, rtr_v3 p) {
    return rtr_mk(rtr_fma(m[8],  p.z, rtr_fma(m[4], p.y, m[0] * p.x)) + m[12],
                  rtr_fma(m[9],  p.z, rtr_fma(m[5], p.y, m[1] * p.x)) + m[13],
                  rtr_fma(m[10], p.z, rtr_fma(m[6], p.y, m[2] * p.x)) + m[14]);
}
/* 3x3 row-major times vector */
RTR_HD rtr_v3 rtr_mul33(const float* m, rtr_v3 p) {
    return rtr_mk(rtr_fma(m[2], p.z, rtr_fma(m[1], p.y, m[0] * p.x)),
                  rtr_fma(m[5], p.z, rtr_fma(m[4], p.y, m[3] * p.x)),
                  rtr_fma(m[8], p.z, rtr_fma(m[7], p.y, m[6] * p.x)));
}
/* transpose(inverse(mat3(O2W))) of a row-major 3x4, as closesthit.rchit:74 builds it.
 * = cofactor matrix / determinant; out is row-major 3x3. */
RTR_HD void rtr_normal_matrix(const float* m, float* out) {
    rtr_v3 r0 = rtr_mk(m[0], m[1], m[2]);
    rtr_v3 r1 = rtr_mk(m[4], m[5], m[6]);
    rtr_v3 r2 = rtr_mk(m[8], m[9], m[10]);
    rtr_v3 c0 = rtr_cross(r1, r2);
    rtr_v3 c1 = rtr_cross(r2, r0);
    rtr_v3 c2 = rtr_cross(r0, r1);
    float det = rtr_dot(r0, c0);
    float inv = 1.0f / det;
    out[0] = c0.x * inv; out[1] = c0.y * inv; out[2] = c0.z * inv;
    out[3] = c1.x * inv; out[4] = c1.y * inv; out[5] = c1.z * inv;
    out[6] = c2.x * inv; out[7] = c2.y * inv; out[8] = c2.z * inv;
}

/* ---- log2 / exp2 / pow ----------------------------------------------------------------- */
/* log2(x) for normal x > 0.  x = m * 2^e with m in [sqrt(1/2), sqrt(2)); with s=(m-1)/(m+1),
 * ln(m) = 2s(1 + s^2/3 + s^4/5 + s^6/7 + s^8/9 + s^10/11); |s| <= 0.1716 so the truncation
 * error is < 2e-10 relative.  The integer part is added last to keep it exact. */
RTR_HD float rtr_log2(float x) {
    uint32_t ix = rtr_f2u(x);
    /* move the split point to sqrt(2)/2: add (1.0 - sqrt(.5)) in bit space */
    uint32_t t = ix + (0x3f800000u - 0x3f3504f3u);
    int32_t e = (int32_t)(t >> 23) - 127;
    uint32_t im = (t & 0x007fffffu) + 0x3f3504f3u;
    float m = rtr_u2f(im);
    float s = (m - 1.0f) / (m + 1.0f);
    float z = s * s;
    float p = 0.09090909090909091f;             /* 1/11 */
    p = rtr_fma(p, z, 0.1111111111111111f);     /* 1/9  */
    p = rtr_fma(p, z, 0.14285714285714285f);    /* 1/7  */
    p = rtr_fma(p, z, 0.2f);                    /* 1/5  */
    p = rtr_fma(p, z, 0.3333333333333333f);     /* 1/3  */
    p = rtr_fma(p, z, 1.0f);
    float lnm2 = (s + s) * p;                   /* ln(m) */
    return rtr_fma(lnm2, 1.4426950408889634f, (float)e);
}
/* 2^z.  z = n + f, f in [-0.5, 0.5]; 2^f = e^(f ln2) by a degree-8 Taylor form
 * (|f ln2| <= 0.3466 -> truncation < 2e-10 relative).  Result 0 below 2^-126. */
RTR_HD float rtr_exp2(float z) {
    if (z < -126.0f) return 0.0f;
    if (z > 127.0f) z = 127.0f;
    float nf = __builtin_floorf(z + 0.5f);
    float f = z - nf;
    float r = f * 0.6931471805599453f;
    float p = 2.48015873015873e-05f;            /* 1/8! */
    p = rtr_fma(p, r, 1.984126984126984e-04f);  /* 1/7! */
    p = rtr_fma(p, r, 1.388888888888889e-03f);  /* 1/6! */
    p = rtr_fma(p, r, 8.333333333333333e-03f);  /* 1/5! */
    p = rtr_fma(p, r, 4.1666666666666664e-02f); /* 1/4! */
    p = rtr_fma(p, r, 0.16666666666666666f);    /* 1/3! */
    p = rtr_fma(p, r, 0.5f);
    p = rtr_fma(p, r, 1.0f);
    p = rtr_fma(p, r, 1.0f);
    int32_t n = (int32_t)nf;
    float scale = rtr_u2f((uint32_t)(n + 127) << 23);
    return p * scale;
}
/* GLSL pow(x,y) for the uses on this path (x >= 0, y > 0: 2.2, 1/2.2, 5.0).
 * x below FLT_MIN (incl. 0, negatives, NaN) -> 0, documented divergence from GLSL's
 * "undefined for x < 0". */
RTR_HD float rtr_pow(float x, float y) {
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    return rtr_exp2(y * rtr_log2(x));
}

/* ---- atan2 / acos (miss.rmiss:19-22 equirect lookup) ------------------------------------------ */
/* atan on [0, tan(pi/8)]: odd degree-9 polynomial (the classic single-precision form; |err| < 2e-7) */
RTR_HD float rtr_atan_small(float x) {
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = rtr_fma(p, z, -1.38776856032e-1f);
    p = rtr_fma(p, z, 1.99777106478e-1f);
    p = rtr_fma(p, z, -3.33329491539e-1f);
    return rtr_fma(p * z, x, x);
}
/* atan2(y, x) in (-pi, pi]; atan2(0,0) = 0.  Range reduction: t = min/max in [0,1], pi/8 split,
 * then octant / quadrant / sign fix-ups. */
RTR_HD float rtr_atan2(float y, float x) {
    const float ax = rtr_abs(x), ay = rtr_abs(y);
    const float mx = ax > ay ? ax : ay, mn = ax > ay ? ay : ax;
    if (!(mx > 0.0f)) return 0.0f;                 /* (0,0) and NaN */
    const float t = mn / mx;
    float r;
    if (t > 0.41421356237f) r = 0.78539816339f + rtr_atan_small((t - 1.0f) / (t + 1.0f));
    else r = rtr_atan_small(t);
    if (ay > ax) r = 1.57079632679f - r;
    if (x < 0.0f) r = 3.14159265359f - r;
    return y < 0.0f ? -r : r;
}
/* acos(x), x in [-1,1]: 2*atan2(sqrt(1-x), sqrt(1+x)) — accurate at both ends */
RTR_HD float rtr_acos(float x) {
    return 2.0f * rtr_atan2(rtr_sqrt(1.0f - x), rtr_sqrt(1.0f + x));
}

/* b / 255.0f for b = 0..255 without the division sequence: fl(1/255) split into a head and a tail, b * tail folded into ONE fma with
 * the head — a single rounding of b/255 to within 2^-50.  Equal to the correctly rounded IEEE quotient for all 256 inputs
 * (tests/test_math.py checks it exhaustively), so the denoise kernels unpack UNORM8 with a conversion + two plain fp32 operations
 * per channel (2-cycle instructions on gfx950) while the oracle divides. */
RTR_HD float rtr_unorm8_to_float(uint32_t b) {
    const float x = (float)b;
    return rtr_fma(x, 0.0039215688593685627f, x * -2.3191758e-10f);      /* fl(1/255), fl(1/255 - fl(1/255)) */
}
/* a / b for a divisor known ahead, r = fl(1 / b): the quotient estimate a * r corrected twice by its exact remainder.  The first
 * correction makes it faithful, and a faithful quotient corrected once more by its remainder times the correctly rounded reciprocal
 * is the correctly rounded quotient (Markstein) — for finite a, b with no overflow / underflow in a * r and the remainders, which is
 * where the denoise pass uses it (|a| <= 4, b = 0.001, 1, 4, 9, 16 ...).  Five plain fp32 operations instead of the IEEE division
 * sequence (v_div_scale x 2, v_rcp, four fma, v_div_fmas, v_div_fixup); a zero keeps its value, not always its sign.
 * tests/test_math.py holds it against the division over every float of the reachable range for the divisors in use. */
RTR_HD float rtr_div_by(float a, float b, float r) {
    const float q0 = a * r;
    const float q1 = rtr_fma(rtr_fma(-b, q0, a), r, q0);
    return rtr_fma(rtr_fma(-b, q1, a), r, q1);
}

/* ---- ray / box / triangle ---------------------------------------------------------------- */
/* Direction components are kept away from 0 so 1/d is finite and the slab test never
 * forms inf*0 or inf-inf (no NaN reaches rtr_hwmin/rtr_hwmax). */
RTR_HD float rtr_safe_rcp_dir(float d) {
    const float tiny = 1e-20f;
    float a = rtr_abs(d);
    float c = a < tiny ? tiny : a;
    float r = 1.0f / c;
    return d < 0.0f ? -r : r;
}

/* Slab test against one box.  t = b*idir + ood (one fma per plane), ood = -o*idir.
 * Returns entry distance through *t_entry; hit iff max(entry, tmin) <= min(exit, tmax)*(1+2^-21).
 * The 1+2^-21 widening (Ize, "Robust BVH ray traversal", 2013, uses 1+2ulp) together with the
 * builder's outward box padding makes the test conservative w.r.t. rtr_mt_intersect's t. */
#define RTR_BOX_WIDEN 1.0000004768371582f
RTR_HD int rtr_slab(const float* bmin, const float* bmax, rtr_v3 idir, rtr_v3 ood,
                    float tmin, float tmax, float* t_entry) {
    float tx0 = rtr_fma(bmin[0], idir.x, ood.x), tx1 = rtr_fma(bmax[0], idir.x, ood.x);
    float ty0 = rtr_fma(bmin[1], idir.y, ood.y), ty1 = rtr_fma(bmax[1], idir.y, ood.y);
    float tz0 = rtr_fma(bmin[2], idir.z, ood.z), tz1 = rtr_fma(bmax[2], idir.z, ood.z);
    float lo = rtr_hwmax(rtr_hwmax(rtr_hwmin(tx0, tx1), rtr_hwmin(ty0, ty1)),
                         rtr_hwmax(rtr_hwmin(tz0, tz1), tmin));
    float hi = rtr_hwmin(rtr_hwmin(rtr_hwmax(tx0, tx1), rtr_hwmax(ty0, ty1)),
                         rtr_hwmin(rtr_hwmax(tz0, tz1), tmax));
    *t_entry = lo;
    return lo <= hi * RTR_BOX_WIDEN;
}

/* ---- 16-bit planes on the scene grid (BVH layout version 3, include/rtr_types.h) ----------------------------------
 * Grid from the (padded) scene bounds: 65532 steps span the extent; the origin sits 1.5 steps below the minimum so every
 * plane of the tree lands strictly inside [0, 65535]. */
RTR_HD void rtr_grid_from_bounds(const float* bmin, const float* bmax, float* origin, float* scale) {
    for (int k = 0; k < 3; ++k) {
        float ext = bmax[k] - bmin[k];
        float s = ext / 65532.0f;
        float floor_s = rtr_max(rtr_abs(bmin[k]), rtr_abs(bmax[k])) * 9.094947e-13f + 1e-30f;   /* 2^-40 of the magnitude: keeps s > 0 for flat scenes */
        if (!(s > floor_s)) s = floor_s;
        scale[k] = s;
        origin[k] = bmin[k] - 1.5f * s;
    }
}
/* Outward quantisation: origin + qlo*scale <= v <= origin + qhi*scale.  Evaluated in double: x is then within 2e-11 of
 * the real quotient, so floor/ceil can be off by one only when v sits within ~1e-11 grid steps of a grid plane — a
 * slip far inside the builder's padding.  (A whole guard step on either side, as a float evaluation needs, made flat
 * boxes thick enough that every shadow ray started inside its own wall's leaves: +14 % triangle tests.)  floor/ceil
 * are exact and the division is IEEE: host and device agree bit for bit. */
RTR_HD uint32_t rtr_quant_lo(float v, float origin, float scale) {
    double q = __builtin_floor(((double)v - (double)origin) / (double)scale);
    q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
    return (uint32_t)q;
}
RTR_HD uint32_t rtr_quant_hi(float v, float origin, float scale) {
    double q = __builtin_ceil(((double)v - (double)origin) / (double)scale);
    q = q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q);
    return (uint32_t)q;
}
/* Per-ray constants of the quantised slab test: t(q) = q * ga + gb with ga = scale * idir, gb = (origin - o) * idir.
 * Same operation count per plane as the fp32 form (one fma) once q is converted.  Rounding: |t(q) - t_exact| corresponds
 * to moving the plane by at most ~(2 extent + 3 |origin - o| + |o + t d|) * 2^-24, far inside the builder's outward
 * padding of 2^-18 * max|coordinate| (ray origins up to ~8 scene sizes away), so the test stays conservative with
 * respect to rtr_mt_intersect's t exactly as rtr_slab was. */
RTR_HD void rtr_ray_grid(rtr_v3 o, rtr_v3 idir, const float* origin, const float* scale, rtr_v3* ga, rtr_v3* gb) {
    ga->x = scale[0] * idir.x; ga->y = scale[1] * idir.y; ga->z = scale[2] * idir.z;
    gb->x = (origin[0] - o.x) * idir.x; gb->y = (origin[1] - o.y) * idir.y; gb->z = (origin[2] - o.z) * idir.z;
}
/* The same constants with the plane coordinate counted from the scene's wide centre c (RtrBvhGrid::wideCentreXY / Z, grid steps): what
 * the half-float planes of the 4-wide records (RtrWideNode) are offsets from.  ga is unchanged; gbc = (origin + c scale - o) * idir. */
/* the wide centre in world units: the part of rtr_ray_grid_centre that does not depend on the ray (the any-hit kernel computes it once per workgroup) */
RTR_HD rtr_v3 rtr_wide_centre_world(const float* origin, const float* scale, uint32_t centreXY, uint32_t centreZ) {
    return rtr_mk(rtr_fma((float)(centreXY & 0xffffu), scale[0], origin[0]), rtr_fma((float)(centreXY >> 16), scale[1], origin[1]),
                  rtr_fma((float)(centreZ & 0xffffu), scale[2], origin[2]));
}
RTR_HD void rtr_ray_grid_about(rtr_v3 o, rtr_v3 idir, rtr_v3 scale, rtr_v3 centreWorld, rtr_v3* ga, rtr_v3* gbc) {
    ga->x = scale.x * idir.x; ga->y = scale.y * idir.y; ga->z = scale.z * idir.z;
    gbc->x = (centreWorld.x - o.x) * idir.x; gbc->y = (centreWorld.y - o.y) * idir.y; gbc->z = (centreWorld.z - o.z) * idir.z;
}
RTR_HD void rtr_ray_grid_centre(rtr_v3 o, rtr_v3 idir, const float* origin, const float* scale, uint32_t centreXY, uint32_t centreZ, rtr_v3* ga, rtr_v3* gbc) {
    rtr_ray_grid_about(o, idir, rtr_mk(scale[0], scale[1], scale[2]), rtr_wide_centre_world(origin, scale, centreXY, centreZ), ga, gbc);
}
/* Slab test of one child box given its six grid coordinates (already widened to 32 bits). */
RTR_HD int rtr_slab_q(uint32_t qminx, uint32_t qminy, uint32_t qminz, uint32_t qmaxx, uint32_t qmaxy, uint32_t qmaxz,
                      rtr_v3 ga, rtr_v3 gb, float tmin, float tmax, float* t_entry) {
    float tx0 = rtr_fma((float)qminx, ga.x, gb.x), tx1 = rtr_fma((float)qmaxx, ga.x, gb.x);
    float ty0 = rtr_fma((float)qminy, ga.y, gb.y), ty1 = rtr_fma((float)qmaxy, ga.y, gb.y);
    float tz0 = rtr_fma((float)qminz, ga.z, gb.z), tz1 = rtr_fma((float)qmaxz, ga.z, gb.z);
    float lo = rtr_hwmax(rtr_hwmax(rtr_hwmin(tx0, tx1), rtr_hwmin(ty0, ty1)),
                         rtr_hwmax(rtr_hwmin(tz0, tz1), tmin));
    float hi = rtr_hwmin(rtr_hwmin(rtr_hwmax(tx0, tx1), rtr_hwmax(ty0, ty1)),
                         rtr_hwmin(rtr_hwmax(tz0, tz1), tmax));
    *t_entry = lo;
    return lo <= hi * RTR_BOX_WIDEN;
}

/* Moeller-Trumbore, following the only in-repo statement of the ray-triangle arithmetic,
 * reference src/shaders/intersect.rint:18-41 (EPSILON, the u / v / u+v rejections, t > tmin),
 * on a precomputed {v0, e1, e2} record.  Returns 1 and (t,u,v) when tmin < t. */
#define RTR_MT_EPSILON 0.00001f
RTR_HD int rtr_mt_intersect(rtr_v3 o, rtr_v3 d, rtr_v3 v0, rtr_v3 e1, rtr_v3 e2,
                            float tmin, float* t_out, float* u_out, float* v_out) {
    rtr_v3 h = rtr_cross(d, e2);
    float a = rtr_dot(e1, h);
    if (rtr_abs(a) < RTR_MT_EPSILON) return 0;
    float f = 1.0f / a;
    rtr_v3 s = rtr_sub(o, v0);
    float u = f * rtr_dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    rtr_v3 q = rtr_cross(s, e1);
    float v = f * rtr_dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * rtr_dot(e2, q);
    if (!(t > tmin)) return 0;
    *t_out = t; *u_out = u; *v_out = v;
    return 1;
}

/* ---- tone map: reference src/shaders/raygen.rgen:45-59 ----------------------------------- */
RTR_HD float rtr_aces(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float num = x * rtr_fma(a, x, b);
    float den = rtr_fma(x, rtr_fma(c, x, d), e);
    return rtr_clamp(num / den, 0.0f, 1.0f);
}
RTR_HD float rtr_to_srgb(float x) { return rtr_pow(x, 0.45454545454545453f); }
RTR_HD float rtr_to_linear(float x) { return rtr_pow(x, 2.2f); }

/* Vulkan UNORM8 store: clamp to [0,1], scale by 255, round to nearest even (NaN -> 0). */
RTR_HD uint32_t rtr_unorm8(float x) {
    float c = rtr_clamp(x, 0.0f, 1.0f);       /* NaN: max(NaN,0)->0 via select form */
    float s = c * 255.0f;
    float r = __builtin_rintf(s);
    return (uint32_t)(int32_t)r;
}
/* imageStore(vec4(b,g,r,1)) into an rgba8 image: bytes in memory are B,G,R,255 (quirk Q14). */
RTR_HD uint32_t rtr_pack_bgra8(float r, float g, float b) {
    return rtr_unorm8(b) | (rtr_unorm8(g) << 8) | (rtr_unorm8(r) << 16) | 0xff000000u;
}

#endif /* RTR_MATH_H */
