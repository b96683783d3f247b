// What one visit of k_shadow_trace4's node loop costs to ISSUE, measured: the vector-instruction mix of the visit block (octant form,
// from the kernel's own assembly: 24 v_fma_mix_f32, 4 v_max3 + 5 v_max, 4 v_min3 + 4 v_min, 4 v_mul (the exit widening), 8 compares
// into SGPR pairs, 6 v_cndmask, 2 v_mov, 3 v_add for the pushes = 64 VALU) with its LDS traffic (one ds_read_b32: the speculative top;
// three ds_write_b32: the pushes) and its mask arithmetic on the scalar unit (18 s_and / s_or / s_xor), issued back to back with the
// dependencies a visit has INSIDE itself (plane times -> min / max tree -> compare -> select) and none between visits — i.e. the rate
// the SIMD could sustain if no load ever stalled a wave.  Eight waves per SIMD (the kernel's occupancy) and one.
//   cycles per VALU instruction per SIMD = median over waves of (s_memtime ticks) / (VALU instructions one wave issued) / (waves per SIMD)
// (Round 4's first version of this file clobbered v40-v80, which made the kernel 88 VGPRs = FIVE resident waves per SIMD while the rate was
// divided by eight: it reported 1.90 where its own event-time check said 3.48.  The block now lives in v16-v56, the kernel is capped at 64
// VGPRs, the host refuses to run unless all workgroups are resident, and the two clocks must agree.)
// SQ_ACTIVE_INST_VALU charges 4 cycles per instruction (roofline.frac's numerator); the guide prices a wave64 instruction at 4 cycles
// of one wave's issue and 2 cycles of the SIMD-32 datapath for plain fp32.  This says which of the two this mix lands on.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/visit_mix.hip -o profiles/microbench/visit_mix && profiles/microbench/visit_mix > profiles/r04/visit_mix_issue.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Stamp { unsigned long long cyc, rt; };

#define FM(d, sel) "v_fma_mix_f32 v" #d ", %1, %2, %3 op_sel:[" #sel ",0,0] op_sel_hi:[1,0,0]\n"
/* one child box: entry = max3(near planes) against tmin, exit = min3(far planes) against tmax, widened, compared */
#define CHILD(p0, p1, p2, p3, p4, p5, lo, hi, sg, sg1) \
    "v_max3_f32 v" #lo ", v" #p0 ", v" #p2 ", v" #p4 "\n v_max_f32 v" #lo ", v" #lo ", %4\n" \
    "v_min3_f32 v" #hi ", v" #p1 ", v" #p3 ", v" #p5 "\n v_min_f32 v" #hi ", v" #hi ", %5\n v_mul_f32 v" #hi ", v" #hi ", %6\n" \
    "v_cmp_le_f32 s[" #sg ":" #sg1 "], v" #lo ", v" #hi "\n"
#define VISIT \
    FM(16, 0) FM(17, 1) FM(18, 0) FM(19, 1) FM(20, 0) FM(21, 1) FM(22, 0) FM(23, 1) FM(24, 0) FM(25, 1) FM(26, 0) FM(27, 1) \
    FM(28, 0) FM(29, 1) FM(30, 0) FM(31, 1) FM(32, 0) FM(33, 1) FM(34, 0) FM(35, 1) FM(36, 0) FM(37, 1) FM(38, 0) FM(39, 1) \
    "ds_read_b32 v56, %7\n" \
    CHILD(16, 17, 18, 19, 20, 21, 40, 41, 52, 53) CHILD(22, 23, 24, 25, 26, 27, 42, 43, 54, 55) CHILD(28, 29, 30, 31, 32, 33, 44, 45, 56, 57) CHILD(34, 35, 36, 37, 38, 39, 46, 47, 58, 59) \
    /* the nearest hit child: three displacement tests, the running nearest t, the code entered */ \
    "v_cndmask_b32_e64 v48, %5, v40, s[52:53]\n" \
    "v_cmp_lt_f32 s[60:61], v42, v48\n s_and_b64 s[60:61], s[60:61], s[54:55]\n v_cndmask_b32_e64 v48, v48, v42, s[60:61]\n" \
    "v_cmp_lt_f32 s[62:63], v44, v48\n s_and_b64 s[62:63], s[62:63], s[56:57]\n v_cndmask_b32_e64 v48, v48, v44, s[62:63]\n" \
    "v_cmp_lt_f32 s[64:65], v46, v48\n s_and_b64 s[64:65], s[64:65], s[58:59]\n" \
    "v_cndmask_b32_e64 v49, %8, %9, s[60:61]\n v_cndmask_b32_e64 v49, v49, %8, s[62:63]\n v_cndmask_b32_e64 v49, v49, %9, s[64:65]\n" \
    /* which slots are stacked: mask arithmetic on the comparison results (scalar unit) */ \
    "s_or_b64 s[66:67], s[60:61], s[62:63]\n s_or_b64 s[66:67], s[66:67], s[64:65]\n s_and_b64 s[68:69], s[52:53], s[66:67]\n" \
    "s_xor_b64 s[70:71], s[62:63], s[64:65]\n s_or_b64 s[70:71], s[70:71], s[60:61]\n s_and_b64 s[72:73], s[54:55], s[70:71]\n" \
    "s_xor_b64 s[74:75], s[64:65], s[62:63]\n s_and_b64 s[74:75], s[56:57], s[74:75]\n s_or_b64 s[76:77], s[52:53], s[54:55]\n" \
    "s_or_b64 s[76:77], s[76:77], s[56:57]\n s_or_b64 s[76:77], s[76:77], s[58:59]\n s_and_b64 s[78:79], s[58:59], s[64:65]\n" \
    "s_or_b64 s[78:79], s[78:79], s[74:75]\n s_or_b64 s[78:79], s[78:79], s[72:73]\n s_or_b64 s[78:79], s[78:79], s[68:69]\n" \
    /* up to three pushes: a store and an add each */ \
    "ds_write_b32 %7, v49 offset:1024\n v_add_u32 v50, %7, %8\n ds_write_b32 %7, v48 offset:2048\n v_add_u32 v51, v50, %8\n ds_write_b32 %7, v49 offset:3072\n v_add_u32 v52, v51, %8\n" \
    "v_mov_b32 v53, v49\n v_mov_b32 v54, v56\n"
constexpr int kValuPerVisit = 24 + 4 * 6 + 3 + 6 + 3 + 2 + 2;      /* 64: fma_mix, the four child blocks (5 VALU + a compare each), three v_cmp_lt, six v_cndmask, three v_add, two v_mov */
static_assert(kValuPerVisit == 64, "count the block");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_visit(float* out, Stamp* st, int iters, unsigned planes) {
    __shared__ int lds[256 * 5];
    float acc = threadIdx.x, b = 1.0001f, c = 0.5f, tmin = 0.001f, tmax = 1.0e4f, widen = 1.0000005f;
    const unsigned addr = threadIdx.x * 4u;
    unsigned c0 = 7u, c1 = 9u;
    lds[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(VISIT VISIT VISIT VISIT
                     : "+v"(acc) : "v"(planes), "v"(b), "v"(c), "v"(tmin), "v"(tmax), "v"(widen), "v"(addr), "v"(c0), "v"(c1)
                     : "memory", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35",
                       "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v56",
                       "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
                       "s74", "s75", "s76", "s77", "s78", "s79");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc + (float)lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};
}

/* cycles per VALU instruction per SIMD by the LAUNCH's duration: HIP events around a launch of `iters` trips and one of 2 x iters, the
 * difference (free of launch and ramp costs) x the shader clock measured inside (s_memtime over s_memrealtime's 100-MHz ticks) over the
 * instructions a SIMD's waves issue.  A wave's own s_memtime span is reported beside it and must NOT be used for the rate: the SIMD serves
 * its older waves first, they leave early, and the median wave spans little more than half of the launch. */
static void run(int wgsPerCu, int cus, float* d, Stamp* dst, double* cyclesPerInst, double* clockMHz, double* byWaveSpan) {
    const int iters = 4096, blocks = cus * wgsPerCu;       /* wgsPerCu workgroups of 4 waves per CU = wgsPerCu waves per SIMD */
    int fit = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, k_visit, 256, 0);
    if (fit < wgsPerCu) { fprintf(stderr, "only %d workgroups of k_visit fit a CU, %d asked for: the per-SIMD rate would be wrong\n", fit, wgsPerCu); exit(1); }
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    hipLaunchKernelGGL(k_visit, dim3(blocks), dim3(256), 0, 0, d, dst, 64, 0x3c004000u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_visit, dim3(blocks), dim3(256), 0, 0, d, dst, iters, 0x3c004000u);
    (void)hipEventRecord(e1);
    hipLaunchKernelGGL(k_visit, dim3(blocks), dim3(256), 0, 0, d, dst, 2 * iters, 0x3c004000u);
    (void)hipEventRecord(e2); (void)hipEventSynchronize(e2);
    float m1 = 0, m2 = 0; (void)hipEventElapsedTime(&m1, e0, e1); (void)hipEventElapsedTime(&m2, e1, e2);
    std::vector<Stamp> h((size_t)blocks * 4);
    (void)hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto& s : h) { cyc.push_back((double)s.cyc); clk.push_back(s.rt ? (double)s.cyc / (double)s.rt * 100.0 : 0.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double insts = (double)iters * 4 * kValuPerVisit;              /* one wave, `iters` trips */
    *clockMHz = clk[clk.size() / 2];
    *cyclesPerInst = (m2 - m1) * 1e-3 * *clockMHz * 1e6 / (insts * wgsPerCu);
    *byWaveSpan = cyc[cyc.size() / 2] / (2 * insts) / wgsPerCu;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; (void)hipMalloc(&d, (size_t)cus * 8 * 256 * sizeof(float));
    Stamp* st; (void)hipMalloc(&st, (size_t)cus * 8 * 4 * sizeof(Stamp));
    double c8, k8, s8, c1, k1, s1, w;
    run(8, cus, d, st, &w, &k8, &s8);      /* warm-up */
    run(8, cus, d, st, &c8, &k8, &s8);
    run(1, cus, d, st, &c1, &k1, &s1);
    printf("{\"what\": \"issue cost of k_shadow_trace4's visit block as a stream: 64 VALU (24 v_fma_mix_f32, 9 v_max / v_max3, 8 v_min / v_min3, 4 v_mul, 7 compares, 6 v_cndmask, 3 v_add, 2 v_mov) "
           "+ 1 ds_read_b32 + 3 ds_write_b32 + 18 scalar mask operations per visit, dependencies inside a visit only, no loads; by the launch's duration (events, t(2n) - t(n)) x the clock measured in the launch\", "
           "\"cycles_per_valu_inst_per_simd_8_waves\": %.3f, \"cycles_per_valu_inst_per_simd_1_wave\": %.3f, \"shader_clock_mhz\": %.1f, "
           "\"by_the_median_waves_own_span_8_waves\": %.3f, \"valu_per_visit\": %d, \"cycles_per_visit_per_simd_8_waves\": %.1f, "
           "\"counter_convention_cycles_per_inst\": 4, \"lands_on\": \"%s\", \"source\": \"profiles/microbench/visit_mix.hip on %s\"}\n",
           c8, c1, k8, s8, kValuPerVisit, c8 * kValuPerVisit,
           c8 >= 3.3 ? "the 4-cycle figure: all of the block but its v_mul / v_add / v_mov issue at 4 cycles (profiles/r04/inst_rates.json), so the counter's convention over-states this mix by ~1.1x only" :
           (c8 <= 2.5 ? "the 2-cycle figure (SIMD-32 datapath): the counter's 4 cycles over-state this mix by up to 2x" : "between the guide's 2-cycle and 4-cycle figures"),
           p.gcnArchName);
    return 0;
}
