// Does packed fp32 (v_pk_fma_f32: two fp32 FMAs per lane and instruction) buy vector throughput on gfx950, or does it take two passes of
// the datapath?  Eight independent accumulator chains per wave, 8 waves per SIMD (and 1), plain v_fma_f32 against v_pk_fma_f32,
// cycles per instruction per SIMD from s_memtime — as profiles/microbench/visit_mix.hip measures the any-hit kernel's mix.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/pk_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R8(x) x x x x x x x x
__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = 1.0001f, c = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk(float* out, unsigned long long* cyc, int iters) {
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = {1.0001f, 1.0002f}, c = {0.5f, 0.25f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                        "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <class K> static double run(K k, int wgsPerCu, int cus, float* d, unsigned long long* dc) {
    const int iters = 2048, blocks = cus * wgsPerCu;
    int fit = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k, 256, 0);
    if (fit < wgsPerCu) { fprintf(stderr, "only %d workgroups fit a CU, %d asked for\n", fit, wgsPerCu); exit(1); }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, dc, 16);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, dc, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    fprintf(stderr, "  %d waves per SIMD: %.3f ms by HIP events = %.3f ns per instruction per SIMD\n", wgsPerCu, ms, ms * 1e6 / ((double)iters * 64 * wgsPerCu));
    std::vector<unsigned long long> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2] / ((double)iters * 64) / wgsPerCu;      /* cycles per instruction per SIMD */
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
    unsigned long long* dc; (void)hipMalloc(&dc, (size_t)cus * 8 * 4 * 8);
    const double f8 = run(k_fma, 8, cus, d, dc), p8 = run(k_pk, 8, cus, d, dc), f1 = run(k_fma, 1, cus, d, dc), p1 = run(k_pk, 1, cus, d, dc);
    printf("{\"what\": \"cycles per vector instruction per SIMD, independent chains\", \"v_fma_f32_8_waves\": %.3f, \"v_pk_fma_f32_8_waves\": %.3f, "
           "\"v_fma_f32_1_wave\": %.3f, \"v_pk_fma_f32_1_wave\": %.3f, \"fp32_fma_per_clock_per_simd_plain\": %.1f, \"fp32_fma_per_clock_per_simd_packed\": %.1f, \"device\": \"%s\"}\n",
           f8, p8, f1, p1, 64.0 / f8, 128.0 / p8, p.gcnArchName);
    return 0;
}
