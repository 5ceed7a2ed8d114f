// How fast does s_memtime tick? One wave spins for a fixed number of ticks, hipEvents time it; once idle, once with
// every CU busy with MFMA work beside it (clocks drop under load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t - t0;
}
__global__ void burn(float* sink, int iters) {
    floatx16 acc = {};
    halfx8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    for (int i = 0; i < iters; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.f) sink[0] = acc[1];
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipStream_t s2; hipStreamCreate(&s2);
    for (int load = 0; load < 2; ++load) {
        for (int rep = 0; rep < 3; ++rep) {
            if (load) hipLaunchKernelGGL(burn, dim3(256 * 8), dim3(256), 0, s2, sink, 400000);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, 20000000ull, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: 2e7 ticks in %.3f ms -> %.1f MHz\n", load ? "beside MFMA load" : "idle GPU", ms, 2e7 / ms / 1e3);
            hipDeviceSynchronize();
        }
    }
    return 0;
}
