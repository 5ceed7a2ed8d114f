// What a scattered dword wave-load costs the CU's vector memory path by the number of DISTINCT 128-byte lines its 64 lanes touch
// (the table reads of mask_from_codes_kernel: one line per lane that needs the table, 27 on average): a 4-MB table (L2-resident),
// every wave-load with k distinct random lines, the lanes spread evenly over them.
//   hipcc -O3 --offload-arch=gfx950 scatter_rate.hip -o scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const float* table, int table_lines, int distinct, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    unsigned h = wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h = h * 1664525u + 1013904223u;                       // wave-uniform stream of line groups
            const unsigned line = (h >> 8) + (unsigned)(lane % distinct) * 7919u;
            acc += table[(size_t)(line % (unsigned)table_lines) * 32 + (lane & 31)];
        }
    }
    out[wave * 64 + lane] = acc;
}
int main() {
    const int lines = (4 << 20) / 128;
    float* table; hipMalloc(&table, (size_t)lines * 128); hipMemset(table, 0, (size_t)lines * 128);
    float* out; hipMalloc(&out, 256 * 8 * 4 * 64 * 4);
    for (int distinct : {1, 2, 4, 8, 16, 27, 32, 64}) {
        const int iters = 200, blocks = 256 * 4;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, lines, distinct, 4, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, table, lines, distinct, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double loads_per_cu = (double)blocks * 4 * iters * 8 / 256;
        printf("%2d distinct lines per wave-load: %7.3f ms  %6.1f cycles per wave-load and CU (2.4 GHz)\n", distinct, ms, ms * 1e-3 * 2.4e9 / loads_per_cu);
    }
    return 0;
}
