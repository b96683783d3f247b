// What a DEPENDENT gather of 64-B records costs a wave on gfx950, by how the 64 lanes ask for them — the memory shape of one visit of
// k_shadow_trace4 (every lane walks its own ray: its next 64-B four-wide record comes from a different cache line than its
// neighbours', and which record comes next is known only when this one has arrived).
//   own    : today's shape.  Lane l loads the four 16-B pieces of ITS record with four buffer_load_dwordx4: every instruction
//            touches 64 different lines (one L1 tag look-up per lane and instruction: 256 per visit of a wave).
//   quad   : the four lanes of a quad load ONE record per instruction, lane j its piece j (instruction i: the record of the quad's
//            lane i): every instruction touches 16 lines, each asked for by four neighbouring lanes, 16 B apiece = the whole line —
//            64 look-ups per visit of a wave if the L1 merges the four.  Lane j then holds piece j of all four records of its quad.
//   pair32 : 32-B records (the BVH2 node of the camera rays) as two pieces, own / lane pairs.
// The table is `mb` MB of records (17 = the Sponza-class tree: lives in the 4-MiB L2s + Infinity Cache), the walk a random
// permutation cycle per lane (so the next index is data of the record just loaded), `waves` waves per SIMD.
// Prints nanoseconds per visit of a wave and the rate per CU; TCP counters can be collected around it with rocprofv3 --pmc.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/gather64.hip -o profiles/microbench/gather64 && profiles/microbench/gather64
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0xffffffff, 0x00020000); }

/* every lane its own record: 4 x 16 B */
__global__ __launch_bounds__(256) void k_own(const uint4* __restrict__ tab, uint32_t* out, int visits, uint32_t mask) {
    const __amdgpu_buffer_rsrc_t b = rsrc(tab);
    uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask, acc = 0;
    for (int v = 0; v < visits; ++v) {
        const int off = (int)(cur << 6);
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(b, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(b, off + 16, 0, 0);
        const u32x4 q2 = __builtin_amdgcn_raw_buffer_load_b128(b, off + 32, 0, 0), q3 = __builtin_amdgcn_raw_buffer_load_b128(b, off + 48, 0, 0);
        acc += q0.y ^ q1.z ^ q2.w ^ q3.y;
        cur = q3.x & mask;                       /* the next record: data of this one (word 12) */
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + cur;
}

/* the quad's four records, one per instruction, lane j its piece j; every lane then needs ITS record's word 12 (piece 3, held by
 * lane 3 of the quad in the register of instruction (own lane)): fetched with ds_bpermute from a register picked by lane number */
__global__ __launch_bounds__(256) void k_quad(const uint4* __restrict__ tab, uint32_t* out, int visits, uint32_t mask) {
    const __amdgpu_buffer_rsrc_t b = rsrc(tab);
    const uint32_t lane = threadIdx.x & 63u, j = lane & 3u, qbase = lane & ~3u;
    uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask, acc = 0;
    for (int v = 0; v < visits; ++v) {
        /* the four current records of the quad */
        const uint32_t c0 = __builtin_amdgcn_mov_dpp(cur, 0x00, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp(cur, 0x55, 0xf, 0xf, true);
        const uint32_t c2 = __builtin_amdgcn_mov_dpp(cur, 0xaa, 0xf, 0xf, true), c3 = __builtin_amdgcn_mov_dpp(cur, 0xff, 0xf, 0xf, true);
        const u32x4 p0 = __builtin_amdgcn_raw_buffer_load_b128(b, (int)((c0 << 6) + j * 16u), 0, 0);
        const u32x4 p1 = __builtin_amdgcn_raw_buffer_load_b128(b, (int)((c1 << 6) + j * 16u), 0, 0);
        const u32x4 p2 = __builtin_amdgcn_raw_buffer_load_b128(b, (int)((c2 << 6) + j * 16u), 0, 0);
        const u32x4 p3 = __builtin_amdgcn_raw_buffer_load_b128(b, (int)((c3 << 6) + j * 16u), 0, 0);
        acc += p0.y ^ p1.z ^ p2.w ^ p3.y;        /* (a traversal would run lane j's child-j slab tests for the four rays here) */
        /* word 12 of record i is p_i.x in lane 3 of the quad: lane 3 offers the one its reader wants through the LDS crossbar */
        const uint32_t a = __builtin_amdgcn_ds_bpermute((int)((qbase + 3u) * 4u), (int)p0.x), bq = __builtin_amdgcn_ds_bpermute((int)((qbase + 3u) * 4u), (int)p1.x);
        const uint32_t c = __builtin_amdgcn_ds_bpermute((int)((qbase + 3u) * 4u), (int)p2.x), d = __builtin_amdgcn_ds_bpermute((int)((qbase + 3u) * 4u), (int)p3.x);
        cur = (j == 0 ? a : (j == 1 ? bq : (j == 2 ? c : d))) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + cur;
}

/* 32-B records: own (2 x 16 B per lane) */
__global__ __launch_bounds__(256) void k_own32(const uint4* __restrict__ tab, uint32_t* out, int visits, uint32_t mask) {
    const __amdgpu_buffer_rsrc_t b = rsrc(tab);
    uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask, acc = 0;
    for (int v = 0; v < visits; ++v) {
        const int off = (int)(cur << 5);
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(b, off, 0, 0), q1 = __builtin_amdgcn_raw_buffer_load_b128(b, off + 16, 0, 0);
        acc += q0.y ^ q1.z;
        cur = q1.x & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + cur;
}

/* ... one 16-B load per lane and visit (a 16-B record): the floor of this access pattern */
__global__ __launch_bounds__(256) void k_own16(const uint4* __restrict__ tab, uint32_t* out, int visits, uint32_t mask) {
    const __amdgpu_buffer_rsrc_t b = rsrc(tab);
    uint32_t cur = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask, acc = 0;
    for (int v = 0; v < visits; ++v) {
        const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(b, (int)(cur << 6), 0, 0);
        acc += q0.y;
        cur = q0.x & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc + cur;
}

template <class K> static double run(const char* name, K k, const uint4* tab, uint32_t* out, int wavesPerSimd, int cus, uint32_t mask, double recBytes) {
    const int visits = 2000, blocks = cus * wavesPerSimd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, out, 50, mask);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, out, visits, mask);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double nsPerVisit = ms * 1e6 / visits;                               /* every wave makes `visits` visits side by side */
    const double waveVisitsPerUsPerCu = (double)wavesPerSimd * 4 * visits / (ms * 1e3);
    printf("  %-44s %d waves/SIMD: %8.1f ns per visit of a wave, %6.2f wave-visits per us per CU, %7.1f GB/s of records\n", name, wavesPerSimd, nsPerVisit, waveVisitsPerUsPerCu,
           (double)blocks * 256 * visits * recBytes / (ms * 1e6));
    return nsPerVisit;
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int mbLog : {18, 20}) {                                              /* 2^18 records x 64 B = 16.8 MB; 2^20 = 67 MB */
        const uint32_t n = 1u << mbLog, mask = n - 1u;
        std::vector<uint32_t> perm(n); std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(7); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint32_t> tab((size_t)n * 16);
        for (uint32_t i = 0; i < n; ++i) { for (int w = 0; w < 16; ++w) tab[(size_t)i * 16 + w] = i * 31u + w; tab[(size_t)i * 16 + 12] = perm[i]; tab[(size_t)i * 16 + 0] = perm[i]; tab[(size_t)i * 16 + 4] = perm[i]; }
        uint4* d; (void)hipMalloc(&d, tab.size() * 4); (void)hipMemcpy(d, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
        uint32_t* out; (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
        printf("table of 2^%d records of 64 B = %.1f MB, random walk, %d CUs\n", mbLog, n * 64.0 / 1e6, cus);
        for (int w : {8, 4}) {
            run("own: 4 x 16 B per lane (today)", k_own, d, out, w, cus, mask, 64);
            run("quad: 4 lanes x 16 B per record and instruction", k_quad, d, out, w, cus, mask, 64);
            run("own, 32-B records: 2 x 16 B per lane", k_own32, d, out, w, cus, mask, 32);
            run("own, one 16-B load per visit", k_own16, d, out, w, cus, mask, 16);
        }
        (void)hipFree(d); (void)hipFree(out);
    }
    return 0;
}
