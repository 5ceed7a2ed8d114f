// In-kernel clock under load: s_memtime (shader cycles) vs s_memrealtime (100 MHz) around an MFMA loop
// and around a VALU min/max loop, every CU busy. Prints the sustained shader clock for each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* out, float* sink, int iters) {
    floatx16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f + 0.37f, y = 1.0001f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = r0; out[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void valu_loop(unsigned long long* out, float* sink, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = sink[threadIdx.x & 3];
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile("v_min_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_max_i32 %3, %3, %4\n"
                     "v_min_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_min_i32 %2, %2, %4\n v_max_i32 %3, %3, %4\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = r0; out[2 * blockIdx.x + 1] = r1 - r0; }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <class K> void probe(const char* name, K k, int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    unsigned long long* d; float* sink;
    hipMalloc(&d, blocks * 16); hipMalloc(&sink, blocks * 256 * 4); hipMemset(sink, 0, blocks * 256 * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float kernel_ms; hipEventElapsedTime(&kernel_ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> dur, start;
    unsigned long long first = ~0ull;
    for (int b = 0; b < blocks; ++b) first = std::min(first, h[2 * b]);
    for (int b = 0; b < blocks; ++b) { dur.push_back(h[2 * b + 1] / 1e5); start.push_back((h[2 * b] - first) / 1e5); }
    std::sort(dur.begin(), dur.end()); std::sort(start.begin(), start.end());
    printf("%-10s blocks/CU=%d  loop ms min %.2f median %.2f max %.2f | start offset ms: median %.2f, p75 %.2f, max %.2f\n", name, blocks_per_cu,
           dur.front(), dur[dur.size() / 2], dur.back(), start[start.size() / 2], start[start.size() * 3 / 4], start.back());
    printf("           whole kernel %.2f ms for %d blocks\n", kernel_ms, blocks);
}
int main() {
    probe("mfma_f32", mfma_loop, 1, 200000);
    probe("mfma_f32", mfma_loop, 2, 200000);
    probe("mfma_f32", mfma_loop, 4, 200000);
    probe("valu_minmax", valu_loop, 4, 400000);
    return 0;
}
