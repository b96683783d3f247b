// Issue cost of single vector instructions on gfx950 with every wave slot filled — measured by the KERNEL's duration, not a wave's.
// Eight independent chains per wave, 8 waves per SIMD (the host refuses to run if the workgroups are not all resident), HIP events
// around a launch of `iters` and one of 2 x `iters` trips: the difference is free of launch and ramp costs.
//   cycles per instruction per SIMD = (t(2 iters) - t(iters)) x shader clock / (instructions a wave issues in `iters` trips x 8 waves)
// the shader clock taken inside the launch (s_memtime ticks over s_memrealtime's 100-MHz ticks, lane 0 of workgroup 0).
// Why not a wave's own s_memtime span (profiles/microbench/valu_rates2.hip, round 2; visit_mix.hip's first version): the SIMD's issue
// arbitration is not fair between its eight waves — the older ones are served first and leave early — so the MEDIAN wave spans about
// half of the launch and the per-wave figure comes out ~2x too low (v_fma_f32: 1.30 "cycles" by a wave's span, 2.65 by the launch).
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/inst_rates.hip -o profiles/microbench/inst_rates && profiles/microbench/inst_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Stamp { unsigned long long cyc, rt; };
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define OPS8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define IO8 "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define KERNEL(name, T, body, ...)                                                                                            \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void name(float* out, Stamp* st, int iters, unsigned sel) { \
        T a0 = (T)(float)threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;  \
        T b = (T)1.0001f, c = (T)0.5f;                                                                                        \
        const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                    \
        for (int i = 0; i < iters; ++i) { REP8(asm volatile(OPS8(body) : IO8    : "v"(b), "v"(c), "v"(sel) : __VA_ARGS__);) }           \
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                    \
        const T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                    \
        out[blockIdx.x * 256 + threadIdx.x] = *(const float*)&s;                                                              \
        if (blockIdx.x == 0 && threadIdx.x == 0) *st = Stamp{c1 - c0, r1 - r0};                                               \
    }

#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_fma, float, I_FMA, "memory")
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_mul, float, I_MUL, "memory")
#define I_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_add, float, I_ADD, "memory")
#define I_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_pkfma, f2, I_PKFMA, "memory")
#define I_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_pkmul, f2, I_PKMUL, "memory")
#define I_FMAMIX(i) "v_fma_mix_f32 %" #i ", %10, %" #i ", %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
KERNEL(k_fmamix, float, I_FMAMIX, "memory")
#define I_MAX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
KERNEL(k_max, float, I_MAX, "memory")
#define I_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_max3, float, I_MAX3, "memory")
#define I_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_min3, float, I_MIN3, "memory")
#define I_CMPVCC(i) "v_cmp_le_f32 vcc, %" #i ", %8\n"
KERNEL(k_cmp_vcc, float, I_CMPVCC, "vcc")
#define I_CMPSG(i) "v_cmp_le_f32 s[52:53], %" #i ", %8\n"
KERNEL(k_cmp_sgpr, float, I_CMPSG, "s52", "s53")
#define I_CNDVCC(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
KERNEL(k_cndmask_vcc, float, I_CNDVCC, "memory")
#define I_CNDSG(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[52:53]\n"
KERNEL(k_cndmask_sgpr, float, I_CNDSG, "memory")
#define I_MOV(i) "v_mov_b32 %" #i ", %8\n"
KERNEL(k_mov, float, I_MOV, "memory")
#define I_ADDU(i) "v_add_u32 %" #i ", %" #i ", %10\n"
KERNEL(k_add_u32, float, I_ADDU, "memory")
#define I_UBYTE(i) "v_cvt_f32_ubyte1 %" #i ", %" #i "\n"
KERNEL(k_cvt_ubyte, float, I_UBYTE, "memory")
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
KERNEL(k_exp, float, I_EXP, "memory")
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
KERNEL(k_rcp, float, I_RCP, "memory")
#define I_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
KERNEL(k_floor, float, I_FLOOR, "memory")
#define I_CMPCND(i) "v_cmp_le_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
KERNEL(k_cmp_then_cndmask, float, I_CMPCND, "vcc")          /* 16 instructions per block */

#define I_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %10\n"
KERNEL(k_mul_lo_u32, float, I_MULLO, "memory")
#define I_XOR(i) "v_xor_b32 %" #i ", %" #i ", %10\n"
KERNEL(k_xor, float, I_XOR, "memory")
#define I_LSHR(i) "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
KERNEL(k_lshr, float, I_LSHR, "memory")
#define I_LSHRV(i) "v_lshrrev_b32 %" #i ", %10, %" #i "\n"
KERNEL(k_lshr_var, float, I_LSHRV, "memory")
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %10\n"
KERNEL(k_and, float, I_AND, "memory")
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %10\n"
KERNEL(k_lshl_add, float, I_LSHLADD, "memory")
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
KERNEL(k_sqrt, float, I_SQRT, "memory")
#define I_DIVFIX(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_div_fixup, float, I_DIVFIX, "memory")
#define I_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
KERNEL(k_div_fmas, float, I_DIVFMAS, "vcc")
#define I_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9\n"
KERNEL(k_div_scale, float, I_DIVSCALE, "vcc")
#define I_CVTU(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
KERNEL(k_cvt_f32_u32, float, I_CVTU, "memory")
#define I_LSHL64(i) "v_lshlrev_b64 %" #i ", 3, %" #i "\n"
KERNEL(k_lshl_b64, f2, I_LSHL64, "memory")
#define I_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 2, %8\n"
KERNEL(k_lshl_add_u64, f2, I_LSHLADD64, "memory")
#define I_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
KERNEL(k_mov_dpp, float, I_MOVDPP, "memory")

template <class K> static void run(const char* name, K k, int perBlock, int cus, float* d, Stamp* ds) {
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
    const double insts = (double)iters * 8 * perBlock * 8;                       /* per SIMD: 8 waves x iters x 8 blocks x perBlock */
    const double cyc = (m2 - m1) * 1e-3 * mhz * 1e6 / insts, waveSpan = (double)s.cyc / ((double)2 * iters * 8 * perBlock) / 8;
    printf("  {\"inst\": \"%s\", \"cycles_per_inst_per_simd\": %.3f, \"lanes_per_clock_per_simd\": %.1f, \"clock_mhz\": %.0f, \"by_one_waves_span\": %.3f},\n", name, cyc, 64.0 / cyc, mhz, waveSpan);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 8);
    Stamp* ds; (void)hipMalloc(&ds, sizeof(Stamp));
    printf("{\"what\": \"cycles per vector instruction per SIMD at 8 resident waves, independent chains, by the launch's duration (events, t(2n) - t(n)) x the clock measured in the launch\", \"device\": \"%s\", \"rates\": [\n", p.gcnArchName);
    run("v_fma_f32", k_fma, 8, cus, d, ds);
    run("v_fma_f32", k_fma, 8, cus, d, ds);
    run("v_mul_f32", k_mul, 8, cus, d, ds);
    run("v_add_f32", k_add, 8, cus, d, ds);
    run("v_pk_fma_f32", k_pkfma, 8, cus, d, ds);
    run("v_pk_mul_f32", k_pkmul, 8, cus, d, ds);
    run("v_fma_mix_f32", k_fmamix, 8, cus, d, ds);
    run("v_max_f32", k_max, 8, cus, d, ds);
    run("v_max3_f32", k_max3, 8, cus, d, ds);
    run("v_min3_f32", k_min3, 8, cus, d, ds);
    run("v_cmp_le_f32 -> vcc", k_cmp_vcc, 8, cus, d, ds);
    run("v_cmp_le_f32 -> sgpr pair", k_cmp_sgpr, 8, cus, d, ds);
    run("v_cndmask_b32 (vcc)", k_cndmask_vcc, 8, cus, d, ds);
    run("v_cndmask_b32_e64 (sgpr pair)", k_cndmask_sgpr, 8, cus, d, ds);
    run("v_cmp + v_cndmask (dependent through vcc)", k_cmp_then_cndmask, 16, cus, d, ds);
    run("v_mov_b32", k_mov, 8, cus, d, ds);
    run("v_add_u32", k_add_u32, 8, cus, d, ds);
    run("v_cvt_f32_ubyte1", k_cvt_ubyte, 8, cus, d, ds);
    run("v_floor_f32", k_floor, 8, cus, d, ds);
    run("v_exp_f32", k_exp, 8, cus, d, ds);
    run("v_rcp_f32", k_rcp, 8, cus, d, ds);
    run("v_mul_lo_u32", k_mul_lo_u32, 8, cus, d, ds);
    run("v_xor_b32", k_xor, 8, cus, d, ds);
    run("v_lshrrev_b32 (constant)", k_lshr, 8, cus, d, ds);
    run("v_lshrrev_b32 (register)", k_lshr_var, 8, cus, d, ds);
    run("v_and_b32", k_and, 8, cus, d, ds);
    run("v_lshl_add_u32", k_lshl_add, 8, cus, d, ds);
    run("v_cvt_f32_u32", k_cvt_f32_u32, 8, cus, d, ds);
    run("v_sqrt_f32", k_sqrt, 8, cus, d, ds);
    run("v_div_scale_f32", k_div_scale, 8, cus, d, ds);
    run("v_div_fmas_f32", k_div_fmas, 8, cus, d, ds);
    run("v_div_fixup_f32", k_div_fixup, 8, cus, d, ds);
    run("v_lshlrev_b64", k_lshl_b64, 8, cus, d, ds);
    run("v_lshl_add_u64", k_lshl_add_u64, 8, cus, d, ds);
    run("v_mov_b32 dpp quad_perm", k_mov_dpp, 8, cus, d, ds);
    printf("  {}]}\n");
    return 0;
}
