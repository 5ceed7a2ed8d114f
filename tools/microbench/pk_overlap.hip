// Standalone reproducer (no Python, no engine) of the interaction in DESIGN.md "Contexts and concurrency": the forward
// STFT kernel on one stream, the f16-split similarity kernels on another; every STFT result is compared with the first.
// `make -C repet-python_amd/csrc repro` links it against the library's own kernel objects twice: build/pk_overlap_pk has
// the FFT kernels built WITH packed-fp32 VALU ops, build/pk_overlap_nopk the shipped ones (without).
// usage: ./pk_overlap_pk [iterations] [aggressor] [victim: 0 = the library's STFT kernel (default), 1 = a loop of inline-asm
//        v_pk_fma/mul/add_f32 on registers, 2 = the same arithmetic in scalar fp32,
//        3 = inline-asm packed butterflies exchanging through LDS]      aggressor:
//        [ 1 = split + f16 Gram (default), 2 = split only, 3 = f16 Gram only, 4 = fp32 Gram,
//        5..8 = nothing but v_mfma_f32_32x32x16_f16 / 32x32x8_f16 / 16x16x32_f16 / 32x32x16_bf16 on registers (132 registers per wave),
//        9 / 10 / 14 = 32x32x16_f16 with AGPR accumulators and 132 / 200 / 164 registers per wave, 11 = 200 registers rewritten by
//        v_mov only, 12 = 32x32x16_f16 with VGPR accumulators, 13 = 32x32x8_f16 with AGPR accumulators and 200 registers, 0 = none]
#include "common.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace repet {
hipError_t ensure_dynamic_lds(const void* fn, int bytes) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace repet
using namespace repet;

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Synthetic aggressors: nothing but one kind of MFMA on register operands (no LDS, no global memory in the loop).
template <int KIND>
__global__ __launch_bounds__(256) void mfma_only(float* sink, int iters) {
    halfx8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (k + 1)); }
    floatx16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    floatx4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {            // gfx950: v_mfma_f32_32x32x16_f16
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0);
        } else if (KIND == 1) {     // gfx90a: v_mfma_f32_32x32x8_f16
            const halfx4 a4 = __builtin_shufflevector(a, a, 0, 1, 2, 3), b4 = __builtin_shufflevector(b, b, 0, 1, 2, 3);
            c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(b4, a4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x8f16(b4, a4, c3, 0, 0, 0);
        } else if (KIND == 2) {     // gfx950: v_mfma_f32_16x16x32_f16
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d3, 0, 0, 0);
        } else if (KIND == 4 || KIND == 5) {   // 32x32x16_f16 with the accumulators in AGPRs (KIND 5: and 136 arch VGPRs allocated)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %5, %4, %2\n v_mfma_f32_32x32x16_f16 %3, %5, %4, %3\n"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
            if (KIND == 5) asm volatile("v_mov_b32 v135, 0" ::: "v135");
        } else if (KIND == 6) {     // no MFMA at all: 200 VGPRs allocated, the upper ones rewritten in a loop
            asm volatile("v_mov_b32 v199, 0x7fc00000\n v_mov_b32 v180, 0x7fc00000\n v_mov_b32 v160, 0x7fc00000\n v_mov_b32 v140, 0x7fc00000\n"
                         "v_mov_b32 v120, 0x7fc00000\n v_mov_b32 v100, 0x7fc00000\n v_mov_b32 v90, 0x7fc00000\n s_nop 7\n"
                         ::: "v199", "v180", "v160", "v140", "v120", "v100", "v90");
        } else if (KIND == 7) {     // 32x32x16_f16, accumulators in VGPRs, 200 VGPRs allocated, no AGPRs
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0);
            asm volatile("v_mov_b32 v199, 0" ::: "v199");
        } else if (KIND == 8) {     // old 32x32x8_f16, AGPR accumulators, 136 arch VGPRs
            asm volatile("v_mfma_f32_32x32x8_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x8_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x8_f16 %2, %5, %4, %2\n v_mfma_f32_32x32x8_f16 %3, %5, %4, %3\n"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(__builtin_shufflevector(a, a, 0, 1, 2, 3)), "v"(__builtin_shufflevector(b, b, 0, 1, 2, 3)));
            asm volatile("v_mov_b32 v135, 0" ::: "v135");
        } else if (KIND == 9) {     // 32x32x16_f16, AGPR accumulators, 100 arch VGPRs (164 in all)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"
                         "v_mfma_f32_32x32x16_f16 %2, %5, %4, %2\n v_mfma_f32_32x32x16_f16 %3, %5, %4, %3\n"
                         : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3) : "v"(a), "v"(b));
            asm volatile("v_mov_b32 v99, 0" ::: "v99");
        } else if (KIND == 3) {     // gfx950: v_mfma_f32_32x32x16_bf16
            bf16x8 x, y;
            for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(float)a[k]; y[k] = (__bf16)(float)b[k]; }
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c3, 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += c0[k] + c1[k] + c2[k] + c3[k];
    for (int k = 0; k < 4; ++k) t += d0[k] + d1[k] + d2[k] + d3[k];
    if (t == 12345.678f) sink[threadIdx.x] = t;
}

// Synthetic victim: nothing but packed-fp32 arithmetic on registers (KIND 0: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// written as inline asm so the build flags cannot change them; KIND 1: the same arithmetic as scalar v_fma_f32).
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void pk_victim(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    floatx2 x = {in[2 * i], in[2 * i + 1]}, y = {0.25f, -0.5f}, z = {1.0f, 0.75f};
    const floatx2 a = {0.999f, 1.001f}, b = {0.001f, -0.002f};
    for (int k = 0; k < iters; ++k) {
        if (KIND == 0) {
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n v_pk_mul_f32 %1, %1, %3\n v_pk_add_f32 %2, %2, %4\n"
                         "v_pk_fma_f32 %1, %0, %4, %1\n v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n"
                         : "+v"(x), "+v"(y), "+v"(z) : "v"(a), "v"(b));
        } else {
            x = x * a + b; y = y * a; z = z + b; y = x * b + y; x = x - z;
            asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
        }
    }
    out[2 * i] = x[0] + y[0] + z[0];
    out[2 * i + 1] = x[1] + y[1] + z[1];
}

// Synthetic victim with LDS in the loop: radix-2-like exchanges through LDS, the arithmetic as inline-asm packed fp32.
__global__ __launch_bounds__(256) void pk_lds_victim(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ floatx2 buf[2][1024];
    const int tid = threadIdx.x;
    const float* src = in + (size_t)(blockIdx.x & 63) * 2048;
    for (int q = 0; q < 4; ++q) buf[0][tid + 256 * q] = floatx2{src[2 * (tid + 256 * q)], src[2 * (tid + 256 * q) + 1]};
    __syncthreads();
    const floatx2 w = {0.70710678f, 0.70710678f}, h = {0.5f, 0.5f};
    int cur = 0;
    for (int k = 0; k < iters; ++k) {
        const int p = 1 << (k % 9);
        for (int q = 0; q < 2; ++q) {
            const int i = tid + 256 * q, lo = i & (p - 1), j = ((i - lo) << 1) + lo;
            floatx2 u0 = buf[cur][i], u1 = buf[cur][i + 512], s, d;
            asm volatile("v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %0, %1\n v_pk_add_f32 %3, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_mul_f32 %2, %2, %5\n v_pk_mul_f32 %3, %3, %5\n"
                         : "+v"(u0), "+v"(u1), "=&v"(s), "=&v"(d) : "v"(w), "v"(h));
            buf[cur ^ 1][j] = s;
            buf[cur ^ 1][j + p] = d;
        }
        __syncthreads();
        cur ^= 1;
    }
    float* dst = out + (size_t)blockIdx.x * 2048;
    for (int q = 0; q < 4; ++q) { dst[2 * (tid + 256 * q)] = buf[cur][tid + 256 * q][0]; dst[2 * (tid + 256 * q) + 1] = buf[cur][tid + 256 * q][1]; }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int aggressor = argc > 2 ? atoi(argv[2]) : 1;
    const int victim = argc > 3 ? atoi(argv[3]) : 0;      // 0 = the library's STFT kernel, 1 = synthetic packed-fp32 loop, 2 = the same loop in scalar fp32
    // victim: STFT of a 9-s mono clip at 16 kHz, W = 1024, H = 512, centred
    const int W = 1024, H = 512, F = W / 2 + 1, FS = 544;
    const int64_t n = 9 * 16000;
    const int64_t T = (n + H - 1) / H + 1;
    std::vector<float> x(n), win(W);
    std::vector<float2> tw(W);
    unsigned seed = 12345u;
    for (auto& v : x) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    for (int i = 0; i < W; ++i) {
        win[i] = (float)(0.54 - 0.46 * std::cos(6.283185307179586 * i / W));
        tw[i] = make_float2((float)std::cos(6.283185307179586 * i / W), (float)(-std::sin(6.283185307179586 * i / W)));
    }
    float *d_x, *d_win, *d_V;
    float2 *d_tw, *d_X;
    const int64_t chan_stride = T * FS;
    CHECK(hipMalloc(&d_x, n * sizeof(float)));
    CHECK(hipMalloc(&d_win, W * sizeof(float)));
    CHECK(hipMalloc(&d_tw, W * sizeof(float2)));
    CHECK(hipMalloc(&d_X, chan_stride * sizeof(float2)));
    CHECK(hipMalloc(&d_V, chan_stride * sizeof(float)));
    CHECK(hipMemcpy(d_x, x.data(), n * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_win, win.data(), W * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tw, tw.data(), W * sizeof(float2), hipMemcpyHostToDevice));
    StftArgs a{};
    a.audio = d_x; a.n_samples = n; a.n_channels = 1; a.sample_offset = 0; a.window = d_win; a.twiddle = d_tw;
    a.W = W; a.H = H; a.T = T; a.FS = FS; a.centred = 1; a.X = d_X; a.V = d_V; a.chan_stride = chan_stride;

    // aggressor: similarity matrix of 2048 unit rows of 513 (+ pad) bins
    const int64_t TA = 2048, TS = 2048;
    std::vector<float> rows(TA * FS, 0.f);
    for (int64_t t = 0; t < TA; ++t) {
        double ss = 0;
        for (int f = 0; f < F; ++f) { seed = seed * 1664525u + 1013904223u; rows[t * FS + f] = ((seed >> 8) & 0xFFFF) / 65536.0f; ss += (double)rows[t * FS + f] * rows[t * FS + f]; }
        for (int f = 0; f < F; ++f) rows[t * FS + f] = (float)(rows[t * FS + f] / std::sqrt(ss));
    }
    float *d_rows, *d_S;
    void* d_planes;
    int2* d_tiles;
    std::vector<int2> tiles;
    const int n_tiles = gram_tile_list((int)(TA / 128), (int)(TA / 128), &tiles);
    CHECK(hipMalloc(&d_rows, rows.size() * sizeof(float)));
    CHECK(hipMalloc(&d_S, TA * TS * sizeof(float)));
    CHECK(hipMalloc(&d_planes, rows.size() * 4));
    CHECK(hipMalloc(&d_tiles, tiles.size() * sizeof(int2)));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tiles, tiles.data(), tiles.size() * sizeof(int2), hipMemcpyHostToDevice));

    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CHECK(launch_split_f16(d_rows, d_planes, (int64_t)rows.size(), sa));
    CHECK(hipStreamSynchronize(sa));

    auto launch_victim = [&]() -> hipError_t {
        if (victim == 0) return launch_stft(a, sv);
        // 2048 workgroups x 256 threads, one float2 each: reads d_x (n >= 2 * 2048 * 256 is not needed: indices wrap below)
        if (victim == 1) hipLaunchKernelGGL(pk_victim<0>, dim3(256), dim3(256), 0, sv, d_x, reinterpret_cast<float*>(d_X), 3000);
        else if (victim == 2) hipLaunchKernelGGL(pk_victim<1>, dim3(256), dim3(256), 0, sv, d_x, reinterpret_cast<float*>(d_X), 3000);
        else hipLaunchKernelGGL(pk_lds_victim, dim3(128), dim3(256), 0, sv, d_x, reinterpret_cast<float*>(d_X), 200);
        return hipGetLastError();
    };
    std::vector<float2> want(chan_stride), got(chan_stride);
    CHECK(launch_victim());
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipMemcpy(want.data(), d_X, chan_stride * sizeof(float2), hipMemcpyDeviceToHost));
    int bad = 0;
    int64_t first_bad_elem = -1, bad_elems = 0;
    for (int it = 0; it < iters; ++it) {
        for (int rep = 0; rep < 2; ++rep) {
            if (aggressor == 1 || aggressor == 2) CHECK(launch_split_f16(d_rows, d_planes, (int64_t)rows.size(), sa));
            if (aggressor == 1 || aggressor == 3) CHECK(launch_gram_full_f16(d_planes, TA, FS, d_S, TS, d_tiles, n_tiles, sa));
            if (aggressor == 4) CHECK(launch_gram_full(d_rows, TA, FS, d_S, TS, d_tiles, n_tiles, sa));
            if (aggressor == 5) hipLaunchKernelGGL(mfma_only<0>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (aggressor == 6) hipLaunchKernelGGL(mfma_only<1>, dim3(1024), dim3(256), 0, sa, d_S, 800);
            if (aggressor == 7) hipLaunchKernelGGL(mfma_only<2>, dim3(1024), dim3(256), 0, sa, d_S, 800);
            if (aggressor == 8) hipLaunchKernelGGL(mfma_only<3>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (aggressor == 9) hipLaunchKernelGGL(mfma_only<4>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (aggressor == 10) hipLaunchKernelGGL(mfma_only<5>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (aggressor == 11) hipLaunchKernelGGL(mfma_only<6>, dim3(1024), dim3(256), 0, sa, d_S, 20000);
            if (aggressor == 12) hipLaunchKernelGGL(mfma_only<7>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (aggressor == 13) hipLaunchKernelGGL(mfma_only<8>, dim3(1024), dim3(256), 0, sa, d_S, 800);
            if (aggressor == 14) hipLaunchKernelGGL(mfma_only<9>, dim3(1024), dim3(256), 0, sa, d_S, 400);
            if (rep == 0) CHECK(launch_victim());
        }
        CHECK(hipStreamSynchronize(sv));
        CHECK(hipMemcpy(got.data(), d_X, chan_stride * sizeof(float2), hipMemcpyDeviceToHost));
        if (memcmp(got.data(), want.data(), chan_stride * sizeof(float2)) != 0) {
            ++bad;
            if (first_bad_elem < 0)
                for (int64_t i = 0; i < chan_stride; ++i)
                    if (memcmp(&got[i], &want[i], sizeof(float2)) != 0) { if (first_bad_elem < 0) first_bad_elem = i; ++bad_elems; }
        }
        if ((it & 63) == 63) CHECK(hipStreamSynchronize(sa));
    }
    CHECK(hipDeviceSynchronize());
    printf("aggressor %d: %d of %d STFT results differ", aggressor, bad, iters);
    if (first_bad_elem >= 0) printf(" (first damaged result: %lld bins, first at frame %lld bin %lld)", (long long)bad_elems,
                                    (long long)(first_bad_elem / FS), (long long)(first_bad_elem % FS));
    printf("\n");
    return 0;
}
