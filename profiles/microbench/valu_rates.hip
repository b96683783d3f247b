// Issue rate of the vector instructions the traversal loop is made of, on gfx950 (one wave64 per SIMD saturating the VALU with
// 8 independent chains).  Build: hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define KERNEL(name, body)                                                                                  \
    __global__ __launch_bounds__(256) void name(float* out, int iters, unsigned sel) {                       \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;  \
        float b = 1.0001f, c = 0.5f;                                                                         \
        for (int i = 0; i < iters; ++i) { REP8(body) }                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                        \
    }

KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_max, asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                           "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_min3, asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                            "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_cvt, asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n"
                           "v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_cvt_sdwa, asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                "v_cvt_f32_u32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                "v_cvt_f32_u32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                "v_cvt_f32_u32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %0, %8\n v_alignbit_b32 %1, %1, %1, %8\n v_alignbit_b32 %2, %2, %2, %8\n v_alignbit_b32 %3, %3, %3, %8\n"
                                "v_alignbit_b32 %4, %4, %4, %8\n v_alignbit_b32 %5, %5, %5, %8\n v_alignbit_b32 %6, %6, %6, %8\n v_alignbit_b32 %7, %7, %7, %8\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sel), "v"(c));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %0, %8\n v_perm_b32 %1, %1, %1, %8\n v_perm_b32 %2, %2, %2, %8\n v_perm_b32 %3, %3, %3, %8\n"
                            "v_perm_b32 %4, %4, %4, %8\n v_perm_b32 %5, %5, %5, %8\n v_perm_b32 %6, %6, %6, %8\n v_perm_b32 %7, %7, %7, %8\n"
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sel), "v"(c));)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)

KERNEL(k_cndmask64, asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
                               "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s10", "s11");)
KERNEL(k_cndmask64vcc, asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n"
                               "v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_mix, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n"
                           "v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                           "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
KERNEL(k_lshlor, asm volatile("v_lshl_or_b32 %0, %0, 3, %8\n v_lshl_or_b32 %1, %1, 3, %8\n v_lshl_or_b32 %2, %2, 3, %8\n v_lshl_or_b32 %3, %3, 3, %8\n"
                              "v_lshl_or_b32 %4, %4, 3, %8\n v_lshl_or_b32 %5, %5, 3, %8\n v_lshl_or_b32 %6, %6, 3, %8\n v_lshl_or_b32 %7, %7, 3, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)

__global__ __launch_bounds__(256) void k_pkfma(float* out, int iters, unsigned sel) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <class K> static void run(const char* name, K k, float* d) {
    const int iters = 4096, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 16, 0x01000302u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0x01000302u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)blocks * 4 / 1024.0 * iters * 64;     // wave-instructions each SIMD issues
    printf("%-14s %8.3f ms  -> %.2f clocks per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run("warm-up fma", k_fma, d); run("v_fma_f32", k_fma, d); run("v_mul_f32", k_mul, d); run("v_lshl_or_b32", k_lshlor, d);
    run("v_cndmask e64", k_cndmask64, d); run("cndmask e64 vcc", k_cndmask64vcc, d); run("1 cnd_vcc+7 fma", k_mix, d); run("v_cmp_lt_f32", k_cmp, d); run("v_max_f32", k_max, d); run("v_min3_f32", k_min3, d); run("v_cvt_f32_u32", k_cvt, d);
    run("v_cvt.._sdwa", k_cvt_sdwa, d); run("v_alignbit_b32", k_alignbit, d); run("v_perm_b32", k_perm, d);
    run("v_cndmask_b32", k_cndmask, d); run("v_pk_fma_f32", k_pkfma, d);
    return 0;
}
