// How many waves per SIMD the fast VALU class (v_bitop3_b32, v_xor_b32 ...: 2.5 cycles per wave-instruction at 8 waves per SIMD)
// needs to reach its rate, and what a dependent chain costs: one workgroup per CU of 4 * W waves, C independent chains per lane.
//   hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void k(unsigned* out, int iters) {
    unsigned a[C];
#pragma unroll
    for (int i = 0; i < C; ++i) a[i] = threadIdx.x * 2654435761u + i;
    unsigned b = out[threadIdx.x & 3], c = out[(threadIdx.x & 3) + 4];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64 / C; ++r)
#pragma unroll
            for (int i = 0; i < C; ++i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(b), "v"(c));
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C> void run(unsigned* d, int waves_per_simd) {
    const int iters = 4000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<C>, dim3(256), dim3(threads), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<C>, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)waves_per_simd * iters * 64;
    printf("chains %d  waves/SIMD %d: %.2f cycles per wave-instr per SIMD, %.2f per wave (2.4 GHz)\n", C, waves_per_simd,
           ms * 1e-3 * 2.4e9 / per_simd, ms * 1e-3 * 2.4e9 / (iters * 64.0));
}
int main() {
    unsigned* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    for (int w : {1, 2, 3, 4}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); }
    return 0;
}
