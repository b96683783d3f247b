// Issue cost of the packed-fp16 / mixed-precision vector instructions a half-float slab test could be made of (round 5: is there a cheaper
// form of the any-hit kernel's 4-wide visit than 24 v_fma_mix_f32 + 16 fp32 min / max?).  Same method as inst_rates.hip: eight
// independent chains per wave, 8 waves per SIMD, the LAUNCH's duration, t(2n) - t(n).
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/inst_rates_f16.hip -o profiles/microbench/inst_rates_f16 && profiles/microbench/inst_rates_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Stamp { unsigned long long cyc, rt; };

#define REP8(x) x x x x x x x x
#define OPS8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define IO8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define KERNEL(name, body, ...)                                                                                               \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void name(unsigned* out, Stamp* st, int iters, unsigned sel) { \
        unsigned a0 = 0x3c003c00u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;    \
        unsigned b = 0x3c013c01u, c = 0x38003800u;  float fb = 1.0001f, fc = 0.5f;                                            \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                    \
        for (int i = 0; i < iters; ++i) { REP8(asm volatile(OPS8(body) : IO8 : "v"(b), "v"(c), "v"(sel), "v"(fb), "v"(fc) : __VA_ARGS__);) }   \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                    \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) *st = Stamp{c1 - c0, r1 - r0};                                               \
    }

#define I_PKFMA16(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_pk_fma_f16, I_PKFMA16, "memory")
#define I_PKMAX16(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_pk_max_f16, I_PKMAX16, "memory")
#define I_PKMIN16(i) "v_pk_min_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_pk_min_f16, I_PKMIN16, "memory")
#define I_PKADD16(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_pk_add_f16, I_PKADD16, "memory")
#define I_PKMUL16(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_pk_mul_f16, I_PKMUL16, "memory")
#define I_MIXLO(i) "v_fma_mixlo_f16 %" #i ", %10, %11, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
KERNEL(k_fma_mixlo_f16, I_MIXLO, "memory")
#define I_MIXHI(i) "v_fma_mixhi_f16 %" #i ", %10, %11, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
KERNEL(k_fma_mixhi_f16, I_MIXHI, "memory")
#define I_MIX32(i) "v_fma_mix_f32 %" #i ", %10, %11, %12 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
KERNEL(k_fma_mix_f32, I_MIX32, "memory")
#define I_CMP16(i) "v_cmp_le_f16 vcc, %" #i ", %8\n"
KERNEL(k_cmp_le_f16, I_CMP16, "vcc")
#define I_CMP16SEL(i) "v_cmp_le_f16_sdwa s[52:53], %" #i ", -%" #i " src0_sel:WORD_0 src1_sel:WORD_1\n"
KERNEL(k_cmp_le_f16_opsel_neg, I_CMP16SEL, "s52", "s53")
#define I_PKADDSEL(i) "v_pk_add_f16 %" #i ", %" #i ", %" #i " op_sel:[0,1] op_sel_hi:[1,0]\n"
KERNEL(k_pk_add_f16_opsel, I_PKADDSEL, "memory")
#define I_MAX16(i) "v_max_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_max_f16, I_MAX16, "memory")
#define I_CVT16(i) "v_cvt_f16_f32 %" #i ", %" #i "\n"
KERNEL(k_cvt_f16_f32, I_CVT16, "memory")
#define I_CVT32(i) "v_cvt_f32_f16 %" #i ", %" #i "\n"
KERNEL(k_cvt_f32_f16, I_CVT32, "memory")
#define I_PKMAXI16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
KERNEL(k_pk_max_i16, I_PKMAXI16, "memory")
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %10\n"
KERNEL(k_perm_b32, I_PERM, "memory")
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %11, %12\n"
KERNEL(k_med3_f32, I_MED3, "memory")
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %11, %12\n"
KERNEL(k_fma_f32, I_FMA32, "memory")

#define I_MAXI32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
KERNEL(k_max_i32, I_MAXI32, "memory")
#define I_MINI32(i) "v_min_i32 %" #i ", %" #i ", %8\n"
KERNEL(k_min_i32, I_MINI32, "memory")
#define I_MAXU32(i) "v_max_u32 %" #i ", %" #i ", %8\n"
KERNEL(k_max_u32, I_MAXU32, "memory")
#define I_MAX3I32(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_max3_i32, I_MAX3I32, "memory")
#define I_MIN3I32(i) "v_min3_i32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_min3_i32, I_MIN3I32, "memory")
#define I_CMPI32(i) "v_cmp_le_i32 vcc, %" #i ", %8\n"
KERNEL(k_cmp_le_i32, I_CMPI32, "vcc")
#define I_MAXF32(i) "v_max_f32 %" #i ", %" #i ", %11\n"
KERNEL(k_max_f32, I_MAXF32, "memory")
#define I_MINF16(i) "v_min_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_min_f16, I_MINF16, "memory")
#define I_FMAF16(i) "v_fma_f16 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_fma_f16, I_FMAF16, "memory")
#define I_ADDF16(i) "v_add_f16 %" #i ", %" #i ", %8\n"
KERNEL(k_add_f16, I_ADDF16, "memory")
#define I_MAXI16(i) "v_max_i16 %" #i ", %" #i ", %8\n"
KERNEL(k_max_i16, I_MAXI16, "memory")
#define I_SUBU32(i) "v_sub_u32 %" #i ", %" #i ", %10\n"
KERNEL(k_sub_u32, I_SUBU32, "memory")
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %10\n"
KERNEL(k_and_or_b32, I_ANDOR, "memory")
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
KERNEL(k_cndmask, I_CNDMASK, "memory")

template <class K> static void run(const char* name, K k, int perBlock, int cus, unsigned* d, Stamp* ds) {
    int fit = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k, 256, 0);
    if (fit < 8) { fprintf(stderr, "%s: only %d workgroups fit a CU\n", name, fit); exit(1); }
    const int iters = 4096, blocks = cus * 8;
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ds, 64, 0x3c004000u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ds, iters, 0x3c004000u);
    (void)hipEventRecord(e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, ds, 2 * iters, 0x3c004000u);
    (void)hipEventRecord(e2); (void)hipEventSynchronize(e2);
    float m1 = 0, m2 = 0; (void)hipEventElapsedTime(&m1, e0, e1); (void)hipEventElapsedTime(&m2, e1, e2);
    Stamp s; (void)hipMemcpy(&s, ds, sizeof s, hipMemcpyDeviceToHost);
    const double mhz = (double)s.cyc / (double)s.rt * 100.0;
    const double insts = (double)iters * 8 * perBlock * 8;
    const double cyc = (m2 - m1) * 1e-3 * mhz * 1e6 / insts;
    printf("  {\"inst\": \"%s\", \"cycles_per_inst_per_simd\": %.3f, \"lanes_per_clock_per_simd\": %.1f, \"clock_mhz\": %.0f},\n", name, cyc, 64.0 / cyc, mhz);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned* d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    Stamp* ds; (void)hipMalloc(&ds, sizeof(Stamp));
    printf("{\"what\": \"cycles per vector instruction per SIMD at 8 resident waves, independent chains, by the launch's duration\", \"device\": \"%s\", \"rates\": [\n", p.gcnArchName);
    run("v_fma_f32", k_fma_f32, 8, cus, d, ds);
    run("v_fma_f32", k_fma_f32, 8, cus, d, ds);
    run("v_fma_mix_f32", k_fma_mix_f32, 8, cus, d, ds);
    run("v_fma_mixlo_f16", k_fma_mixlo_f16, 8, cus, d, ds);
    run("v_fma_mixhi_f16", k_fma_mixhi_f16, 8, cus, d, ds);
    run("v_pk_fma_f16", k_pk_fma_f16, 8, cus, d, ds);
    run("v_pk_max_f16", k_pk_max_f16, 8, cus, d, ds);
    run("v_pk_min_f16", k_pk_min_f16, 8, cus, d, ds);
    run("v_pk_add_f16", k_pk_add_f16, 8, cus, d, ds);
    run("v_pk_mul_f16", k_pk_mul_f16, 8, cus, d, ds);
    run("v_max_f16", k_max_f16, 8, cus, d, ds);
    run("v_cmp_le_f16 -> vcc", k_cmp_le_f16, 8, cus, d, ds);
    run("v_cmp_le_f16_sdwa WORD_0 <= -WORD_1 -> sgpr pair", k_cmp_le_f16_opsel_neg, 8, cus, d, ds);
    run("v_pk_add_f16 op_sel (lo + hi)", k_pk_add_f16_opsel, 8, cus, d, ds);
    run("v_cvt_f16_f32", k_cvt_f16_f32, 8, cus, d, ds);
    run("v_cvt_f32_f16", k_cvt_f32_f16, 8, cus, d, ds);
    run("v_pk_max_i16", k_pk_max_i16, 8, cus, d, ds);
    run("v_perm_b32", k_perm_b32, 8, cus, d, ds);
    run("v_med3_f32", k_med3_f32, 8, cus, d, ds);
    run("v_max_f32", k_max_f32, 8, cus, d, ds);
    run("v_max_i32", k_max_i32, 8, cus, d, ds);
    run("v_min_i32", k_min_i32, 8, cus, d, ds);
    run("v_max_u32", k_max_u32, 8, cus, d, ds);
    run("v_max3_i32", k_max3_i32, 8, cus, d, ds);
    run("v_min3_i32", k_min3_i32, 8, cus, d, ds);
    run("v_cmp_le_i32 -> vcc", k_cmp_le_i32, 8, cus, d, ds);
    run("v_min_f16", k_min_f16, 8, cus, d, ds);
    run("v_fma_f16", k_fma_f16, 8, cus, d, ds);
    run("v_add_f16", k_add_f16, 8, cus, d, ds);
    run("v_max_i16", k_max_i16, 8, cus, d, ds);
    run("v_sub_u32", k_sub_u32, 8, cus, d, ds);
    run("v_and_or_b32", k_and_or_b32, 8, cus, d, ds);
    run("v_cndmask_b32 (vcc)", k_cndmask, 8, cus, d, ds);
    printf("  {}]}\n");
    return 0;
}
