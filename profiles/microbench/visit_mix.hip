// What one visit of k_shadow_trace4's node loop costs to ISSUE, measured: the vector-instruction mix of the visit block (octant form,
// from the kernel's own assembly: 24 v_fma_mix_f32, 4 v_max3 + 5 v_max, 4 v_min3 + 4 v_min, 4 v_mul (the exit widening), 8 compares
// into SGPR pairs, 6 v_cndmask, 2 v_mov, 3 v_add for the pushes = 64 VALU) with its LDS traffic (one ds_read_b32: the speculative top;
// three ds_write_b32: the pushes) and its mask arithmetic on the scalar unit (18 s_and / s_or / s_xor), issued back to back with the
// dependencies a visit has INSIDE itself (plane times -> min / max tree -> compare -> select) and none between visits — i.e. the rate
// the SIMD could sustain if no load ever stalled a wave.  Eight waves per SIMD (the kernel's occupancy) and one.
//   cycles per VALU instruction per SIMD = median over waves of (s_memtime ticks) / (VALU instructions one wave issued) / (waves per SIMD)
// SQ_ACTIVE_INST_VALU charges 4 cycles per instruction (roofline.frac's numerator); the guide prices a wave64 instruction at 4 cycles
// of one wave's issue and 2 cycles of the SIMD-32 datapath for plain fp32.  This says which of the two this mix lands on.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/visit_mix.hip -o profiles/microbench/visit_mix && profiles/microbench/visit_mix > profiles/r04/visit_mix_issue.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long cyc, rt; };

#define FM(d, sel) "v_fma_mix_f32 v" #d ", %1, %2, %3 op_sel:[" #sel ",0,0] op_sel_hi:[1,0,0]\n"
/* one child box: entry = max3(near planes) against tmin, exit = min3(far planes) against tmax, widened, compared */
#define CHILD(p0, p1, p2, p3, p4, p5, lo, hi, sg, sg1) \
    "v_max3_f32 v" #lo ", v" #p0 ", v" #p2 ", v" #p4 "\n v_max_f32 v" #lo ", v" #lo ", %4\n" \
    "v_min3_f32 v" #hi ", v" #p1 ", v" #p3 ", v" #p5 "\n v_min_f32 v" #hi ", v" #hi ", %5\n v_mul_f32 v" #hi ", v" #hi ", %6\n" \
    "v_cmp_le_f32 s[" #sg ":" #sg1 "], v" #lo ", v" #hi "\n"
#define VISIT \
    FM(40, 0) FM(41, 1) FM(42, 0) FM(43, 1) FM(44, 0) FM(45, 1) FM(46, 0) FM(47, 1) FM(48, 0) FM(49, 1) FM(50, 0) FM(51, 1) \
    FM(52, 0) FM(53, 1) FM(54, 0) FM(55, 1) FM(56, 0) FM(57, 1) FM(58, 0) FM(59, 1) FM(60, 0) FM(61, 1) FM(62, 0) FM(63, 1) \
    "ds_read_b32 v80, %7\n" \
    CHILD(40, 41, 42, 43, 44, 45, 64, 65, 52, 53) CHILD(46, 47, 48, 49, 50, 51, 66, 67, 54, 55) CHILD(52, 53, 54, 55, 56, 57, 68, 69, 56, 57) CHILD(58, 59, 60, 61, 62, 63, 70, 71, 58, 59) \
    /* the nearest hit child: three displacement tests, the running nearest t, the code entered */ \
    "v_cndmask_b32_e64 v72, %5, v64, s[52:53]\n" \
    "v_cmp_lt_f32 s[60:61], v66, v72\n s_and_b64 s[60:61], s[60:61], s[54:55]\n v_cndmask_b32_e64 v72, v72, v66, s[60:61]\n" \
    "v_cmp_lt_f32 s[62:63], v68, v72\n s_and_b64 s[62:63], s[62:63], s[56:57]\n v_cndmask_b32_e64 v72, v72, v68, s[62:63]\n" \
    "v_cmp_lt_f32 s[64:65], v70, v72\n s_and_b64 s[64:65], s[64:65], s[58:59]\n" \
    "v_cndmask_b32_e64 v73, %8, %9, s[60:61]\n v_cndmask_b32_e64 v73, v73, %8, s[62:63]\n v_cndmask_b32_e64 v73, v73, %9, s[64:65]\n" \
    /* which slots are stacked: mask arithmetic on the comparison results (scalar unit) */ \
    "s_or_b64 s[66:67], s[60:61], s[62:63]\n s_or_b64 s[66:67], s[66:67], s[64:65]\n s_and_b64 s[68:69], s[52:53], s[66:67]\n" \
    "s_xor_b64 s[70:71], s[62:63], s[64:65]\n s_or_b64 s[70:71], s[70:71], s[60:61]\n s_and_b64 s[72:73], s[54:55], s[70:71]\n" \
    "s_xor_b64 s[74:75], s[64:65], s[62:63]\n s_and_b64 s[74:75], s[56:57], s[74:75]\n s_or_b64 s[76:77], s[52:53], s[54:55]\n" \
    "s_or_b64 s[76:77], s[76:77], s[56:57]\n s_or_b64 s[76:77], s[76:77], s[58:59]\n s_and_b64 s[78:79], s[58:59], s[64:65]\n" \
    "s_or_b64 s[78:79], s[78:79], s[74:75]\n s_or_b64 s[78:79], s[78:79], s[72:73]\n s_or_b64 s[78:79], s[78:79], s[68:69]\n" \
    /* up to three pushes: a store and an add each */ \
    "ds_write_b32 %7, v73 offset:1024\n v_add_u32 v74, %7, %8\n ds_write_b32 %7, v72 offset:2048\n v_add_u32 v75, v74, %8\n ds_write_b32 %7, v73 offset:3072\n v_add_u32 v76, v75, %8\n" \
    "v_mov_b32 v77, v73\n v_mov_b32 v78, v80\n"
constexpr int kValuPerVisit = 24 + 4 * 6 + 3 + 6 + 3 + 2 + 2;      /* 64: fma_mix, the four child blocks (5 VALU + a compare each), three v_cmp_lt, six v_cndmask, three v_add, two v_mov */
static_assert(kValuPerVisit == 64, "count the block");

__global__ __launch_bounds__(256) void k_visit(float* out, Stamp* st, int iters, unsigned planes) {
    __shared__ int lds[256 * 5];
    float acc = threadIdx.x, b = 1.0001f, c = 0.5f, tmin = 0.001f, tmax = 1.0e4f, widen = 1.0000005f;
    const unsigned addr = threadIdx.x * 4u;
    unsigned c0 = 7u, c1 = 9u;
    lds[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile(VISIT VISIT VISIT VISIT
                     : "+v"(acc) : "v"(planes), "v"(b), "v"(c), "v"(tmin), "v"(tmax), "v"(widen), "v"(addr), "v"(c0), "v"(c1)
                     : "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59",
                       "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v80",
                       "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
                       "s74", "s75", "s76", "s77", "s78", "s79");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc + (float)lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, r1 - r0};
}

static void run(int wgsPerCu, int cus, float* d, Stamp* dst, double* cyclesPerInst, double* clockMHz, double* eventCheck) {
    const int iters = 4096, blocks = cus * wgsPerCu;       /* wgsPerCu workgroups of 4 waves per CU = wgsPerCu waves per SIMD */
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_visit, dim3(blocks), dim3(256), 0, 0, d, dst, 64, 0x3c004000u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_visit, dim3(blocks), dim3(256), 0, 0, d, dst, iters, 0x3c004000u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h((size_t)blocks * 4);
    hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (auto& s : h) { cyc.push_back((double)s.cyc); clk.push_back(s.rt ? (double)s.cyc / (double)s.rt * 100.0 : 0.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double insts = (double)iters * 4 * kValuPerVisit;
    *cyclesPerInst = cyc[cyc.size() / 2] / insts / wgsPerCu; *clockMHz = clk[clk.size() / 2];
    *eventCheck = ms * 1e-3 * *clockMHz * 1e6 / (insts * wgsPerCu);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* d; hipMalloc(&d, (size_t)cus * 8 * 256 * sizeof(float));
    Stamp* st; hipMalloc(&st, (size_t)cus * 8 * 4 * sizeof(Stamp));
    double c8, k8, e8, c1, k1, e1, w;
    run(8, cus, d, st, &w, &k8, &e8);      /* warm-up */
    run(8, cus, d, st, &c8, &k8, &e8);
    run(1, cus, d, st, &c1, &k1, &e1);
    printf("{\"what\": \"issue cost of k_shadow_trace4's visit block as a stream: 64 VALU (24 v_fma_mix_f32, 9 v_max / v_max3, 8 v_min / v_min3, 4 v_mul, 7 compares, 6 v_cndmask, 3 v_add, 2 v_mov) "
           "+ 1 ds_read_b32 + 3 ds_write_b32 + 18 scalar mask operations per visit, dependencies inside a visit only, no loads\", "
           "\"cycles_per_valu_inst_per_simd_8_waves\": %.3f, \"cycles_per_valu_inst_per_simd_1_wave\": %.3f, \"shader_clock_mhz\": %.1f, "
           "\"event_time_check_8_waves\": %.3f, \"valu_per_visit\": %d, \"cycles_per_visit_per_simd_8_waves\": %.1f, "
           "\"counter_convention_cycles_per_inst\": 4, \"lands_on\": \"%s\", \"source\": \"profiles/microbench/visit_mix.hip on %s\"}\n",
           c8, c1, k8, e8, kValuPerVisit, c8 * kValuPerVisit,
           c8 >= 3.5 ? "the 4-cycle figure (one wave-instruction per SIMD per 4 cycles): the counter's convention is this mix's real rate" :
           (c8 <= 2.5 ? "the 2-cycle figure (SIMD-32 datapath): the counter's 4 cycles over-state this mix by up to 2x" : "between the guide's 2-cycle and 4-cycle figures"),
           p.gcnArchName);
    return 0;
}
