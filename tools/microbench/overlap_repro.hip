// Minimal search for the cross-kernel interaction described in DESIGN.md ("Contexts and concurrency"): an aggressor
// that only issues v_mfma_f32_32x32x16_f16 on one stream, a victim on another stream whose output is compared with
// its own solo run. Victim flavours: 0 = LDS ping-pong with complex (packed-fp32) math, 1 = the same math in registers
// only (no LDS), 2 = LDS ping-pong moving data only (no packed math).
//   hipcc -O3 --offload-arch=gfx950 overlap_repro.hip -o overlap_repro && ./overlap_repro [rounds]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void aggressor(float* sink, int iters) {
    halfx8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (k + 1)); }
    floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += acc0[k] + acc1[k] + acc2[k] + acc3[k];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <int FLAVOUR>
__global__ __launch_bounds__(256) void victim(const float2* __restrict__ in, float2* __restrict__ out, int rounds) {
    __shared__ float2 buf0[1024];
    __shared__ float2 buf1[1024];
    const int tid = threadIdx.x;
    const float2* src = in + (size_t)blockIdx.x * 1024;
    float2 r[4];
    for (int q = 0; q < 4; ++q) r[q] = src[tid + 256 * q];
    if (FLAVOUR != 1) {
        for (int q = 0; q < 4; ++q) buf0[tid + 256 * q] = r[q];
        __syncthreads();
    }
    float2* a = buf0;
    float2* b = buf1;
    const float2 w1 = make_float2(0.99999f, 0.004363f), w2 = make_float2(0.70710678f, -0.70710678f);
    for (int it = 0; it < rounds; ++it) {
        if (FLAVOUR == 1) {
            const float2 t0 = make_float2(r[0].x + r[2].x, r[0].y + r[2].y), t1 = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
            const float2 u1 = cmul(r[1], w1), u3 = cmul(r[3], w2);
            r[0] = make_float2(0.5f * (t0.x + u1.x), 0.5f * (t0.y + u1.y));
            r[1] = make_float2(0.5f * (t1.x + u3.y), 0.5f * (t1.y - u3.x));
            r[2] = make_float2(0.5f * (t0.x - u1.x), 0.5f * (t0.y - u1.y));
            r[3] = make_float2(0.5f * (t1.x - u3.y), 0.5f * (t1.y + u3.x));
        } else {
            const int i = tid, k = i & 63, j = ((i - k) << 2) + k;
            float2 u0 = a[i], u1 = a[i + 256], u2 = a[i + 512], u3 = a[i + 768];
            if (FLAVOUR == 0) {
                u1 = cmul(u1, w1); u3 = cmul(u3, w2);
                const float2 t0 = make_float2(u0.x + u2.x, u0.y + u2.y), t1 = make_float2(u0.x - u2.x, u0.y - u2.y);
                u0 = make_float2(0.5f * (t0.x + u1.x), 0.5f * (t0.y + u1.y));
                u2 = make_float2(0.5f * (t0.x - u1.x), 0.5f * (t0.y - u1.y));
                u1 = make_float2(0.5f * (t1.x + u3.y), 0.5f * (t1.y - u3.x));
                u3 = make_float2(0.5f * (t1.x - u3.y), 0.5f * (t1.y + u3.x));
            }
            b[(j) & 1023] = u0; b[(j + 64) & 1023] = u1; b[(j + 128) & 1023] = u2; b[(j + 192) & 1023] = u3;
            __syncthreads();
            float2* t = a; a = b; b = t;
        }
    }
    float2* dst = out + (size_t)blockIdx.x * 1024;
    if (FLAVOUR == 1) for (int q = 0; q < 4; ++q) dst[tid + 256 * q] = r[q];
    else for (int q = 0; q < 4; ++q) dst[tid + 256 * q] = a[tid + 256 * q];
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FLAVOUR>
int run(int rounds, bool with_aggressor, const float2* d_in, float2* d_out, float* d_sink, std::vector<float2>& ref, bool make_ref) {
    const int blocks = 2048;
    hipStream_t sv, sa;
    CHECK(hipStreamCreate(&sv));
    CHECK(hipStreamCreate(&sa));
    std::vector<float2> host((size_t)blocks * 1024);
    int bad_runs = 0;
    for (int r = 0; r < rounds; ++r) {
        if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, d_sink, 2000);
        hipLaunchKernelGGL(victim<FLAVOUR>, dim3(blocks), dim3(256), 0, sv, d_in, d_out, 40);
        if (with_aggressor) hipLaunchKernelGGL(aggressor, dim3(1024), dim3(256), 0, sa, d_sink, 2000);
        CHECK(hipStreamSynchronize(sv));
        CHECK(hipMemcpy(host.data(), d_out, host.size() * sizeof(float2), hipMemcpyDeviceToHost));
        if (make_ref) { ref = host; make_ref = false; continue; }
        if (memcmp(host.data(), ref.data(), host.size() * sizeof(float2)) != 0) ++bad_runs;
    }
    CHECK(hipDeviceSynchronize());
    (void)hipStreamDestroy(sv);
    (void)hipStreamDestroy(sa);
    return bad_runs;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
    const int blocks = 2048;
    std::vector<float2> h_in((size_t)blocks * 1024);
    for (size_t i = 0; i < h_in.size(); ++i) h_in[i] = make_float2(0.001f * (float)(i % 977), -0.002f * (float)(i % 613));
    float2 *d_in, *d_out;
    float* d_sink;
    CHECK(hipMalloc(&d_in, h_in.size() * sizeof(float2)));
    CHECK(hipMalloc(&d_out, h_in.size() * sizeof(float2)));
    CHECK(hipMalloc(&d_sink, 4096));
    CHECK(hipMemcpy(d_in, h_in.data(), h_in.size() * sizeof(float2), hipMemcpyHostToDevice));
    std::vector<float2> ref;
    printf("rounds %d\n", rounds);
    { int b0 = run<0>(3, false, d_in, d_out, d_sink, ref, true); int b = run<0>(rounds, false, d_in, d_out, d_sink, ref, false); int c = run<0>(rounds, true, d_in, d_out, d_sink, ref, false);
      printf("victim 0 (LDS + complex math): solo mismatches %d, beside f16 MFMA %d (warm %d)\n", b, c, b0); }
    { int b0 = run<1>(3, false, d_in, d_out, d_sink, ref, true); int b = run<1>(rounds, false, d_in, d_out, d_sink, ref, false); int c = run<1>(rounds, true, d_in, d_out, d_sink, ref, false);
      printf("victim 1 (registers only):     solo mismatches %d, beside f16 MFMA %d (warm %d)\n", b, c, b0); }
    { int b0 = run<2>(3, false, d_in, d_out, d_sink, ref, true); int b = run<2>(rounds, false, d_in, d_out, d_sink, ref, false); int c = run<2>(rounds, true, d_in, d_out, d_sink, ref, false);
      printf("victim 2 (LDS moves only):     solo mismatches %d, beside f16 MFMA %d (warm %d)\n", b, c, b0); }
    return 0;
}
