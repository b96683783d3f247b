// Does a HIP launch on this stack pass more than 4 KB of kernel arguments by value?  (RTR_MAX_BATCH is sized by the answer.)
//   hipcc --offload-arch=gfx950 kernarg_size.hip -o /tmp/kernarg_size && /tmp/kernarg_size
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> struct Big { unsigned v[N]; };
template <int N> __global__ void k(Big<N> b, unsigned* out) { unsigned s = 0; for (int i = threadIdx.x; i < N; i += 64) s += b.v[i]; atomicAdd(out, s); }
template <int N> static void run(unsigned* d) {
    Big<N> b; unsigned want = 0; for (int i = 0; i < N; ++i) { b.v[i] = (unsigned)(i * 2654435761u) >> 8; want += b.v[i]; }
    hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, b, d);
    hipError_t e = hipGetLastError(); hipError_t e2 = hipDeviceSynchronize();
    unsigned got = 0; hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
    printf("%6zu bytes of arguments: launch %s, sync %s, sum %s\n", sizeof(b) + sizeof(d), hipGetErrorString(e), hipGetErrorString(e2), got == want ? "right" : "WRONG");
}
int main() { unsigned* d; hipMalloc(&d, 4); run<512>(d); run<1000>(d); run<1020>(d); run<1024>(d); run<2048>(d); run<4096>(d); run<16000>(d); return 0; }
