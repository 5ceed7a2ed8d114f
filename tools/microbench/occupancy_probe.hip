// How many waves of a kernel with V VGPRs and L bytes of LDS per 64-thread workgroup does a CU really hold?
// Every workgroup records the s_memrealtime span it was resident for while spinning ~50 us; the host counts how many overlap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int V>
__global__ __launch_bounds__(64) void probe(unsigned long long* span, float* sink) {
    extern __shared__ float lds[];
    float r[V];
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    while (t - t0 < 5000) {                       // 50 us at 100 MHz
#pragma unroll
        for (int i = 0; i < V; ++i) r[i] = r[i] * 1.0001f + 0.5f;
        t = __builtin_amdgcn_s_memrealtime();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < V; ++i) s += r[i];
    lds[threadIdx.x] = s;
    if (threadIdx.x == 0) { span[2 * blockIdx.x] = t0; span[2 * blockIdx.x + 1] = t; }
    if (s == 1.2345f) sink[0] = lds[(threadIdx.x + 1) & 63];
}
template <int V>
void run(int lds_bytes) {
    const int n = 256 * 24;
    unsigned long long* span; float* sink;
    hipMalloc(&span, n * 16); hipMalloc(&sink, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(probe<V>, dim3(n), dim3(64), lds_bytes, 0, span, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2 * n);
    hipMemcpy(h.data(), span, n * 16, hipMemcpyDeviceToHost);
    unsigned long long t0 = h[0];
    for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
    int first_round = 0;
    for (int i = 0; i < n; ++i) if (h[2 * i] - t0 < 2500) ++first_round;      // started in the first 25 us
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, probe<V>, 64, lds_bytes);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&probe<V>));
    printf("regs kept %3d (numRegs %3d)  LDS %6d B : %5d workgroups resident at once = %.2f per CU   (occupancy API: %d per CU)\n",
           V, fa.numRegs, lds_bytes, first_round, first_round / 256.0, api);
    hipFree(span); hipFree(sink);
}
int main() {
    run<84>(1024); run<88>(1024); run<92>(1024); run<96>(1024); run<100>(1024); run<104>(1024); run<108>(1024); run<112>(1024);
    run<116>(1024); run<120>(1024); run<124>(1024);
    for (int lds : {32768, 32000, 40384, 36864}) run<40>(lds);
    return 0;
}
