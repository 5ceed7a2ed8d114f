// K1 / K9: Hamming-windowed STFT and overlap-add inverse for gfx950, written around an in-LDS
// Stockham FFT (radix 4 with a closing radix-2 stage). A real frame of W samples is transformed
// as ONE complex FFT of W/2 points (even samples -> re, odd -> im) plus a split/merge pass, so a
// frame never leaves LDS between windowing and the magnitude / channel-mean epilogue.
//
// Replaces repet.py:1001-1060 (_stft), :1063-1105 (_istft), the magnitude + channel mean of
// :158,:162,:667 and the column normalisation of :1220.
#include "common.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace repet {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// In-LDS Stockham autosort FFT of N complex points by all threads of the block.
// in: `a` filled and synchronised. Returns the buffer holding the result (synchronised).
// tw[m] = exp(-2 pi i m / TW) for m < TW, TW a multiple of N. INVERSE conjugates (no 1/N scale).
template <int N, bool INVERSE>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const float2* __restrict__ tw, int TW) {
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int p = 1; p < N;) {
        if (N / p >= 4) {
            const int tstep = TW / (p * 4);
            for (int i = tid; i < N / 4; i += nth) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                float2 u0 = a[i], u1 = a[i + N / 4], u2 = a[i + N / 2], u3 = a[i + 3 * N / 4];
                if (p > 1) {
                    float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                    if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
                    u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
                }
                const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3);
                const float2 d = csub(u1, u3);
                const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                b[j] = cadd(t0, t2);
                b[j + p] = cadd(t1, t3);
                b[j + 2 * p] = csub(t0, t2);
                b[j + 3 * p] = csub(t1, t3);
            }
            p *= 4;
        } else {
            const int tstep = TW / (p * 2);
            for (int i = tid; i < N / 2; i += nth) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                float2 u0 = a[i], u1 = a[i + N / 2];
                float2 w1 = tw[k * tstep];
                if (INVERSE) w1 = cconj(w1);
                u1 = cmul(u1, w1);
                b[j] = cadd(u0, u1);
                b[j + p] = csub(u0, u1);
            }
            p *= 2;
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

constexpr int kFftThreads = 256;

// s_memtime stamps of one workgroup in 61 (diagnostic build only: make stamps, tools/peak_stamps.py)
#ifdef REPET_FFT_STAMPS
__device__ unsigned long long g_fft_stamps[2 * 8 * 8];      // [kernel][sampled workgroup][phase] summed cycles
#define FSTAMP_DECL unsigned long long fst_prev = __builtin_amdgcn_s_memtime(); const bool fst_on = threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x % 61) == 7 && blockIdx.x / 61 < 8;
#define FSTAMP(kern, k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (fst_on) g_fft_stamps[(kern) * 64 + (blockIdx.x / 61) * 8 + (k)] += now_ - fst_prev; fst_prev = now_; }
#else
#define FSTAMP_DECL
#define FSTAMP(kern, k)
#endif

// Same Stockham FFT with the per-thread stage twiddles held in registers. Thread `tid` always owns the
// butterflies i = tid + 256*q (q < BPT), whose twiddle exponent k = i & (p-1) depends only on the stage, so
// the 3 twiddles per butterfly and stage are loaded ONCE per workgroup (fft_twiddles) and reused for every
// transform the workgroup performs -- the global twiddle fetch leaves the per-stage critical path.
template <int N> struct FftPlan {
    static constexpr int kRadix4Stages = []() { int s = 0; for (int p = 1; N / p >= 4; p *= 4) ++s; return s; }();
    static constexpr bool kFinalRadix2 = (N >> (2 * kRadix4Stages)) == 2;
    static constexpr int kBpt4 = (N / 4 + kFftThreads - 1) / kFftThreads;     // radix-4 butterflies per thread
    static constexpr int kBpt2 = (N / 2 + kFftThreads - 1) / kFftThreads;
};

template <int N>
struct FftTwiddles {
    float2 w4[FftPlan<N>::kRadix4Stages > 0 ? FftPlan<N>::kRadix4Stages : 1][FftPlan<N>::kBpt4][3];
    float2 w2[FftPlan<N>::kBpt2];
};

template <int N, bool INVERSE>
__device__ __forceinline__ void fft_twiddles(FftTwiddles<N>& t, const float2* __restrict__ tw, int TW) {
    using P = FftPlan<N>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < P::kRadix4Stages; ++s) {
        const int p = 1 << (2 * s);
        const int tstep = TW / (p * 4);
#pragma unroll
        for (int q = 0; q < P::kBpt4; ++q) {
            const int i = tid + kFftThreads * q;
            const int k = i & (p - 1);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // the load is unconditional (index 0 for an idle thread) and the choice made afterwards: a load inside a
                // branch is waited for at the join, so the sixteen table loads of the prologue came back one after the
                // other (12 000 cycles per workgroup of the inverse kernel)
                const float2 l = tw[(i < N / 4) ? (r + 1) * k * tstep : 0];
                const float2 w = (i < N / 4) ? l : make_float2(1.f, 0.f);
                t.w4[s][q][r] = INVERSE ? cconj(w) : w;
            }
        }
    }
    if (P::kFinalRadix2) {
        const int p = N / 2;
        const int tstep = TW / (p * 2);
#pragma unroll
        for (int q = 0; q < P::kBpt2; ++q) {
            const int i = tid + kFftThreads * q;
            const float2 l = tw[(i < N / 2) ? (i & (p - 1)) * tstep : 0];
            const float2 w = (i < N / 2) ? l : make_float2(1.f, 0.f);
            t.w2[q] = INVERSE ? cconj(w) : w;
        }
    }
}

template <int N, bool INVERSE>
__device__ __forceinline__ float2* fft_lds_regs(float2* a, float2* b, const FftTwiddles<N>& t) {
    using P = FftPlan<N>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < P::kRadix4Stages; ++s) {
        const int p = 1 << (2 * s);
#pragma unroll
        for (int q = 0; q < P::kBpt4; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 4) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                float2 u0 = a[i], u1 = a[i + N / 4], u2 = a[i + N / 2], u3 = a[i + 3 * N / 4];
                if (s > 0) { u1 = cmul(u1, t.w4[s][q][0]); u2 = cmul(u2, t.w4[s][q][1]); u3 = cmul(u3, t.w4[s][q][2]); }
                const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3);
                const float2 d = csub(u1, u3);
                const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                b[j] = cadd(t0, t2);
                b[j + p] = cadd(t1, t3);
                b[j + 2 * p] = csub(t0, t2);
                b[j + 3 * p] = csub(t1, t3);
            }
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    if (P::kFinalRadix2) {
        const int p = N / 2;
#pragma unroll
        for (int q = 0; q < P::kBpt2; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 2) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                const float2 u0 = a[i];
                const float2 u1 = cmul(a[i + N / 2], t.w2[q]);
                b[j] = cadd(u0, u1);
                b[j + p] = csub(u0, u1);
            }
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

// Same Stockham FFT with the stage twiddles read from an LDS copy of the root table tw[m] = exp(-2 pi i m / TW),
// m <= 3 TW / 4 (TW a multiple of 4N): no twiddle registers, no per-workgroup gather of 15 scattered loads. A
// stage reads 3 more float2 per butterfly from LDS; early stages hit a handful of addresses (broadcast).
template <int N, bool INVERSE>
__device__ __forceinline__ float2* fft_lds_tab(float2* a, float2* b, const float2* tw, int TW) {
    using P = FftPlan<N>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < P::kRadix4Stages; ++s) {
        const int p = 1 << (2 * s);
        const int tstep = TW / (p * 4);
#pragma unroll
        for (int q = 0; q < P::kBpt4; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 4) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                float2 u0 = a[i], u1 = a[i + N / 4], u2 = a[i + N / 2], u3 = a[i + 3 * N / 4];
                if (s > 0) {
                    float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                    if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
                    u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
                }
                const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3);
                const float2 d = csub(u1, u3);
                const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                b[j] = cadd(t0, t2);
                b[j + p] = cadd(t1, t3);
                b[j + 2 * p] = csub(t0, t2);
                b[j + 3 * p] = csub(t1, t3);
            }
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    if (P::kFinalRadix2) {
        const int p = N / 2;
        const int tstep = TW / (p * 2);
#pragma unroll
        for (int q = 0; q < P::kBpt2; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 2) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                float2 w1 = tw[k * tstep];
                if (INVERSE) w1 = cconj(w1);
                const float2 u0 = a[i];
                const float2 u1 = cmul(a[i + N / 2], w1);
                b[j] = cadd(u0, u1);
                b[j + p] = csub(u0, u1);
            }
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

constexpr int kStftFrameRun = 4;     // least frames per workgroup: the tables are loaded once per run (launch_stft picks the run)

// Two transforms at once by the same 256 threads (the two channels of a stereo frame): every stage does the
// butterflies of both before the ONE barrier they share and reads its twiddles once, so each thread has two
// independent LDS -> math -> LDS chains in flight. a0/b0 and a1/b1 are the ping-pong pairs; tw as in fft_lds_tab.
template <int N, bool INVERSE>
__device__ __forceinline__ void fft_lds_tab2(float2*& a0, float2*& b0, float2*& a1, float2*& b1, const float2* tw, int TW) {
    using P = FftPlan<N>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < P::kRadix4Stages; ++s) {
        const int p = 1 << (2 * s);
        const int tstep = TW / (p * 4);
#pragma unroll
        for (int q = 0; q < P::kBpt4; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 4) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                float2 u0 = a0[i], u1 = a0[i + N / 4], u2 = a0[i + N / 2], u3 = a0[i + 3 * N / 4];
                float2 v0 = a1[i], v1 = a1[i + N / 4], v2 = a1[i + N / 2], v3 = a1[i + 3 * N / 4];
                if (s > 0) {
                    float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                    if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
                    u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
                    v1 = cmul(v1, w1); v2 = cmul(v2, w2); v3 = cmul(v3, w3);
                }
                {
                    const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3), d = csub(u1, u3);
                    const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                    b0[j] = cadd(t0, t2); b0[j + p] = cadd(t1, t3); b0[j + 2 * p] = csub(t0, t2); b0[j + 3 * p] = csub(t1, t3);
                }
                {
                    const float2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), d = csub(v1, v3);
                    const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                    b1[j] = cadd(t0, t2); b1[j + p] = cadd(t1, t3); b1[j + 2 * p] = csub(t0, t2); b1[j + 3 * p] = csub(t1, t3);
                }
            }
        }
        __syncthreads();
        float2* tmp = a0; a0 = b0; b0 = tmp;
        tmp = a1; a1 = b1; b1 = tmp;
    }
    if (P::kFinalRadix2) {
        const int p = N / 2;
        const int tstep = TW / (p * 2);
#pragma unroll
        for (int q = 0; q < P::kBpt2; ++q) {
            const int i = tid + kFftThreads * q;
            if (i < N / 2) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                float2 w1 = tw[k * tstep];
                if (INVERSE) w1 = cconj(w1);
                const float2 u0 = a0[i], u1 = cmul(a0[i + N / 2], w1);
                const float2 v0 = a1[i], v1 = cmul(a1[i + N / 2], w1);
                b0[j] = cadd(u0, u1); b0[j + p] = csub(u0, u1);
                b1[j] = cadd(v0, v1); b1[j + p] = csub(v0, v1);
            }
        }
        __syncthreads();
        float2* tmp = a0; a0 = b0; b0 = tmp;
        tmp = a1; a1 = b1; b1 = tmp;
    }
}

// STFT of channel PAIRS (n_channels even, W <= 2048): same outputs as stft_kernel. The two channels of a frame are
// transformed together (fft_lds_tab2), their interleaved samples arrive as one float2 per sample and are fetched
// one frame ahead; window and twiddles live in LDS (see stft_kernel for why).
template <int W>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(3, 8))) void stft_pair_kernel(StftArgs a, int run) {
    constexpr int N = W / 2;
    constexpr int SLOTS = N / kFftThreads + 1;
    constexpr int LOADS = (N + kFftThreads - 1) / kFftThreads;
    constexpr int kTwLoads = (3 * W / 4 + 1 + kFftThreads - 1) / kFftThreads;
    __shared__ float2 bufs[4][N];
    __shared__ float2 win_lds[N];
    __shared__ float2 tw_lds[3 * W / 4 + 1];
    __shared__ float red[kFftThreads / kWave];

    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int C = a.n_channels;
    FSTAMP_DECL
    a.sample_offset += b * a.batch_sample_stride;
    a.X += b * a.batch_spec_stride;
    a.V += b * a.batch_spec_stride;
    if (a.Vm) a.Vm += b * a.batch_mean_stride;
    if (a.Vn) a.Vn += b * a.batch_mean_stride;
    if (a.Vh) a.Vh = static_cast<_Float16*>(a.Vh) + 2 * b * a.batch_mean_stride;
    if (a.P) a.P += b * a.batch_mean_stride;
    {
        float2 wreg[LOADS], treg[kTwLoads];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int n = tid + kFftThreads * i;
            wreg[i] = *reinterpret_cast<const float2*>(a.window + 2 * (n < N ? n : 0));
        }
#pragma unroll
        for (int i = 0; i < kTwLoads; ++i) {
            const int m = tid + kFftThreads * i;
            treg[i] = a.twiddle[m <= 3 * W / 4 ? m : 0];
        }
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int n = tid + kFftThreads * i;
            if (n < N) win_lds[n] = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < kTwLoads; ++i) {
            const int m = tid + kFftThreads * i;
            if (m <= 3 * W / 4) tw_lds[m] = treg[i];
        }
    }
    const int64_t t_begin = (int64_t)blockIdx.x * run;
    const int64_t t_end = (t_begin + run < a.T) ? t_begin + run : a.T;

    float2 raw0[LOADS], raw1[LOADS];               // (channel c, channel c+1) of samples 2n and 2n+1
    auto fetch = [&](int64_t t, int c) {
        const int64_t start = t * a.H - (a.centred ? W / 2 : 0);
        const float* base = a.audio + ((a.sample_offset + start) * C + c);
        if (start >= 0 && start + W <= a.n_samples) {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                const int m = n < N ? n : 0;
                raw0[i] = *reinterpret_cast<const float2*>(base + (2 * m) * C);
                raw1[i] = *reinterpret_cast<const float2*>(base + (2 * m + 1) * C);
            }
        } else {
            // out-of-range samples read sample 0 of the clip and are zeroed by value: selecting between the global
            // address and the address of a local zero would put that zero in scratch and turn the loads into flat ones
            const float* clip = a.audio + (a.sample_offset * C + c);
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                const int64_t s0 = start + 2 * n, s1 = s0 + 1;
                const bool in0 = n < N && s0 >= 0 && s0 < a.n_samples, in1 = n < N && s1 >= 0 && s1 < a.n_samples;
                const float2 v0 = *reinterpret_cast<const float2*>(clip + (in0 ? s0 : 0) * C);
                const float2 v1 = *reinterpret_cast<const float2*>(clip + (in1 ? s1 : 0) * C);
                raw0[i] = make_float2(in0 ? v0.x : 0.f, in0 ? v0.y : 0.f);
                raw1[i] = make_float2(in1 ? v1.x : 0.f, in1 ? v1.y : 0.f);
            }
        }
    };
    if (t_begin < t_end) fetch(t_begin, 0);
    __syncthreads();
    FSTAMP(0, 0)                                   // prologue: tables into LDS, first fetch issued

    for (int64_t t = t_begin; t < t_end; ++t) {
        const int64_t row = t * a.FS;
        float acc[SLOTS];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) acc[i] = 0.f;

        for (int c = 0; c < C; c += 2) {
            float2 *a0 = bufs[0], *b0 = bufs[1], *a1 = bufs[2], *b1 = bufs[3];
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                if (n < N) {
                    const float2 w = win_lds[n];
                    a0[n] = make_float2(raw0[i].x * w.x, raw1[i].x * w.y);
                    a1[n] = make_float2(raw0[i].y * w.x, raw1[i].y * w.y);
                }
            }
            FSTAMP(0, 1)                           // wait for the fetched samples, window, LDS
            {
                const bool same_frame = c + 2 < C;
                const int64_t tn = same_frame ? t : t + 1;
                if (tn < t_end) fetch(tn, same_frame ? c + 2 : 0);
            }
            __syncthreads();
            FSTAMP(0, 2)                           // next fetch issued, barrier
            fft_lds_tab2<N, false>(a0, b0, a1, b1, tw_lds, W);
            FSTAMP(0, 3)                           // the FFT stages
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const int k = tid + kFftThreads * i;
                if (k <= N) {
                    const float2 tw = tw_lds[k];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const float2* Z = half == 0 ? a0 : a1;
                        const float2 zk = Z[k & (N - 1)];
                        const float2 zc = cconj(Z[(N - k) & (N - 1)]);
                        const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
                        const float2 d = csub(zk, zc);
                        const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);
                        const float2 x = cadd(e, cmul(tw, o));
                        const float mag = magnitude(x);
                        a.X[(c + half) * a.chan_stride + row + k] = x;
                        a.V[(c + half) * a.chan_stride + row + k] = mag;
                        acc[i] += mag;
                    }
                }
            }
            if (tid < a.FS - (N + 1)) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    a.X[(c + half) * a.chan_stride + row + N + 1 + tid] = make_float2(0.f, 0.f);
                    a.V[(c + half) * a.chan_stride + row + N + 1 + tid] = 0.f;
                }
            }
            __syncthreads();
            FSTAMP(0, 4)                           // split, magnitudes, X / V stores, barrier
        }

        if (a.Vm == nullptr && a.Vn == nullptr && a.P == nullptr) continue;
        const float inv_c = 1.0f / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int k = tid + kFftThreads * i;
            acc[i] = acc[i] * inv_c;
            if (k <= N) ss += acc[i] * acc[i];
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) ss += __shfl_down(ss, off);
        if ((tid & (kWave - 1)) == 0) red[tid / kWave] = ss;
        __syncthreads();
        float total = 0.f;
#pragma unroll
        for (int w = 0; w < kFftThreads / kWave; ++w) total += red[w];
        const float norm = sqrtf(total);
        FSTAMP(0, 5)                               // channel mean, norm reduction
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int k = tid + kFftThreads * i;
            if (k <= N) {
                if (a.Vm) a.Vm[row + k] = acc[i];
                if (a.Vn) a.Vn[row + k] = acc[i] / norm;
                if (a.Vh) store_split_f16(a.Vh, row + k, acc[i] / norm);
                if (a.P) a.P[row + k] = acc[i] * acc[i];
            }
        }
        if (tid < a.FS - (N + 1)) {
            if (a.Vm) a.Vm[row + N + 1 + tid] = 0.f;
            if (a.Vn) a.Vn[row + N + 1 + tid] = 0.f;
            if (a.Vh) store_split_f16(a.Vh, row + N + 1 + tid, 0.f);
            if (a.P) a.P[row + N + 1 + tid] = 0.f;
        }
        __syncthreads();
        FSTAMP(0, 6)                               // mean / unit / squared rows stored, barrier
    }
}

#ifndef REPET_STFT_MIN_WAVES
#define REPET_STFT_MIN_WAVES 4
#endif
#ifndef REPET_ISTFT_MIN_WAVES
#define REPET_ISTFT_MIN_WAVES 1
#endif
template <int W>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(W <= 2048 ? REPET_STFT_MIN_WAVES : 1, 8))) void stft_kernel(StftArgs a, int run) {
    constexpr int N = W / 2;                       // complex FFT length; also the Nyquist bin index
    constexpr int SLOTS = N / kFftThreads + 1;     // bins k = tid + 256*i, k <= N
    constexpr int LOADS = (N + kFftThreads - 1) / kFftThreads;
    // The kernel is bound by global-memory LATENCY, not by the FFT (removing all five stages saves 10 %): a
    // workgroup used to wait for its samples, then for the split twiddles, once per transform. Now the window and
    // the split twiddles sit in LDS for the whole run and the samples of the NEXT transform are fetched into
    // registers before the current one is transformed.
    constexpr bool kTables = W <= 2048;            // 2 x W/2 float2 of LDS beside the two FFT buffers (64 KB static limit)
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    __shared__ float2 win_lds[kTables ? N : 1];
    __shared__ float2 tw_lds[kTables ? 3 * W / 4 + 1 : 1];      // exp(-2 pi i m / W): stage and split twiddles
    __shared__ float red[kFftThreads / kWave];

    const int tid = threadIdx.x;
    const int64_t b = blockIdx.y;
    const int C = a.n_channels;
    a.sample_offset += b * a.batch_sample_stride;
    a.X += b * a.batch_spec_stride;
    a.V += b * a.batch_spec_stride;
    if (a.Vm) a.Vm += b * a.batch_mean_stride;
    if (a.Vn) a.Vn += b * a.batch_mean_stride;
    if (a.Vh) a.Vh = static_cast<_Float16*>(a.Vh) + 2 * b * a.batch_mean_stride;
    if (a.P) a.P += b * a.batch_mean_stride;

    FftTwiddles<kTables ? 4 : N> ft;                // register twiddles only where the tables do not fit in LDS
    if (kTables) {
        // unrolled so that the loads of a thread are all in flight together (a runtime loop waits for each)
        constexpr int kTwLoads = (3 * W / 4 + 1 + kFftThreads - 1) / kFftThreads;
        float2 wreg[LOADS], treg[kTwLoads];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int n = tid + kFftThreads * i;
            wreg[i] = *reinterpret_cast<const float2*>(a.window + 2 * (n < N ? n : 0));
        }
#pragma unroll
        for (int i = 0; i < kTwLoads; ++i) {
            const int m = tid + kFftThreads * i;
            treg[i] = a.twiddle[m <= 3 * W / 4 ? m : 0];
        }
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int n = tid + kFftThreads * i;
            if (n < N) win_lds[n] = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < kTwLoads; ++i) {
            const int m = tid + kFftThreads * i;
            if (m <= 3 * W / 4) tw_lds[m] = treg[i];
        }
    } else {
        fft_twiddles<kTables ? 4 : N, false>(ft, a.twiddle, W);
    }
    const int64_t t_begin = (int64_t)blockIdx.x * run;
    const int64_t t_end = (t_begin + run < a.T) ? t_begin + run : a.T;

    float2 raw[LOADS];                             // samples (2n, 2n+1) of the transform about to be windowed
    auto fetch = [&](int64_t t, int c) {
        const int64_t start = t * a.H - (a.centred ? W / 2 : 0);
        const float* base = a.audio + ((a.sample_offset + start) * C + c);   // only dereferenced inside the clip
        if (start >= 0 && start + W <= a.n_samples) {          // block-uniform: the frame lies inside the clip
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                const int m = n < N ? n : 0;
                raw[i] = make_float2(base[(2 * m) * C], base[(2 * m + 1) * C]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                const int64_t s0 = start + 2 * n, s1 = s0 + 1;
                const float x0 = (n < N && s0 >= 0 && s0 < a.n_samples) ? base[(2 * n) * C] : 0.f;
                const float x1 = (n < N && s1 >= 0 && s1 < a.n_samples) ? base[(2 * n + 1) * C] : 0.f;
                raw[i] = make_float2(x0, x1);
            }
        }
    };
    constexpr bool kPrefetch = W <= 2048;          // larger windows keep 16+ samples per thread: no room to hold two sets
    if (kPrefetch && t_begin < t_end) fetch(t_begin, 0);
    __syncthreads();                               // tables visible

    for (int64_t t = t_begin; t < t_end; ++t) {
        const int64_t row = t * a.FS;
        float acc[SLOTS];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) acc[i] = 0.f;

        for (int c = 0; c < C; ++c) {
            if (!kPrefetch) fetch(t, c);
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int n = tid + kFftThreads * i;
                if (n < N) {
                    const float2 w = kTables ? win_lds[n] : *reinterpret_cast<const float2*>(a.window + 2 * n);
                    buf0[n] = make_float2(raw[i].x * w.x, raw[i].y * w.y);
                }
            }
            // next transform of this workgroup: same frame, next channel, or the next frame's first channel
            if (kPrefetch) {                 // one call site: a second one costs a vmcnt(0) at the join
                const bool same_frame = c + 1 < C;
                const int64_t tn = same_frame ? t : t + 1;
                if (tn < t_end) fetch(tn, same_frame ? c + 1 : 0);
            }
            __syncthreads();
            const float2* Z;
            if constexpr (kTables) Z = fft_lds_tab<N, false>(buf0, buf1, tw_lds, W);
            else Z = fft_lds_regs<N, false>(buf0, buf1, ft);
            float2* Xrow = a.X + c * a.chan_stride + row;
            float* Vrow = a.V + c * a.chan_stride + row;
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const int k = tid + kFftThreads * i;
                if (k <= N) {
                    const float2 zk = Z[k & (N - 1)];
                    const float2 zc = cconj(Z[(N - k) & (N - 1)]);
                    const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
                    const float2 d = csub(zk, zc);
                    const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);   // (zk - zc) / (2i)
                    const float2 x = cadd(e, cmul(kTables ? tw_lds[k] : a.twiddle[k], o));
                    const float mag = magnitude(x);
                    Xrow[k] = x;
                    Vrow[k] = mag;
                    acc[i] += mag;
                }
            }
            if (tid < a.FS - (N + 1)) {      // zero the pad bins [F, FS)
                Xrow[N + 1 + tid] = make_float2(0.f, 0.f);
                Vrow[N + 1 + tid] = 0.f;
            }
            __syncthreads();                 // Z (buf0/buf1) is re-filled by the next channel / frame
        }

        if (a.Vm == nullptr && a.Vn == nullptr && a.P == nullptr) continue;

        // channel mean (repet.py:162,:667 np.mean(axis=2)) and its squared L2 norm over frequency
        const float inv_c = 1.0f / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int k = tid + kFftThreads * i;
            acc[i] = (C == 1) ? acc[i] : acc[i] * inv_c;
            if (k <= N) ss += acc[i] * acc[i];
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) ss += __shfl_down(ss, off);
        if ((tid & (kWave - 1)) == 0) red[tid / kWave] = ss;
        __syncthreads();
        float total = 0.f;
#pragma unroll
        for (int w = 0; w < kFftThreads / kWave; ++w) total += red[w];
        const float norm = sqrtf(total);     // 0 for a silent frame: 0/0 = NaN like repet.py:1220
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int k = tid + kFftThreads * i;
            if (k <= N) {
                if (a.Vm) a.Vm[row + k] = acc[i];
                if (a.Vn) a.Vn[row + k] = acc[i] / norm;
                if (a.Vh) store_split_f16(a.Vh, row + k, acc[i] / norm);
                if (a.P) a.P[row + k] = acc[i] * acc[i];
            }
        }
        if (tid < a.FS - (N + 1)) {
            if (a.Vm) a.Vm[row + N + 1 + tid] = 0.f;
            if (a.Vn) a.Vn[row + N + 1 + tid] = 0.f;
            if (a.Vh) store_split_f16(a.Vh, row + N + 1 + tid, 0.f);
            if (a.P) a.P[row + N + 1 + tid] = 0.f;
        }
        __syncthreads();                     // red[] is reused by the next frame
    }
}

template <int W>
__global__ __launch_bounds__(kFftThreads) void istft_frames_kernel(IstftArgs a) {
    constexpr int N = W / 2;
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int c = blockIdx.y;
    const float2* Y = a.Y + c * a.chan_stride + t * a.FS;
    for (int k = tid; k < N; k += kFftThreads) {
        const float2 xk = Y[k];
        const float2 xc = cconj(Y[N - k]);
        const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
        const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
        const float2 o = cmul(d, cconj(a.twiddle[k]));       // * exp(+2 pi i k / W)
        buf0[k] = make_float2(e.x - o.y, e.y + o.x);         // E + i O
    }
    __syncthreads();
    const float2* z = fft_lds<N, true>(buf0, buf1, a.twiddle, W);
    const float scale = 1.0f / (float)N;
    float2* out = reinterpret_cast<float2*>(a.frames + (c * a.T + t) * (int64_t)W);
    for (int n = tid; n < N; n += kFftThreads) out[n] = make_float2(z[n].x * scale, z[n].y * scale);
}

__global__ __launch_bounds__(256) void overlap_add_kernel(OlaArgs a) {
    const int C = a.n_channels;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_out * C) return;
    const int64_t n = e / C;
    const int c = (int)(e - n * C);
    const int64_t m = n + a.trim;
    int64_t j_lo = (m - a.W + a.H) / a.H;        // ceil((m - W + 1) / H) for m - W + 1 > 0
    if (m - a.W + 1 <= 0) j_lo = 0;
    int64_t j_hi = m / a.H;
    if (j_hi > a.T - 1) j_hi = a.T - 1;
    float sum = 0.f;
    for (int64_t j = j_lo; j <= j_hi; ++j) sum += a.frames[(c * a.T + j) * (int64_t)a.W + (m - j * a.H)];
    sum *= a.scale;
    float* dst = a.out + (a.out_offset + n) * C + c;
    if (a.accumulate_weighted) {
        // stage export: a single segment, so only its own rise and (if asked for) one fall over its last samples
        float w = segment_weight(n, a.fade_in, 0, 0, 0);
        if (a.fade_out > 0 && n >= a.n_out - a.fade_out) w *= (float)(2 * (a.n_out - 1 - n) + 1) / (float)(2 * a.fade_out);
        *dst += w * sum;
    } else {
        *dst = sum;
    }
}

// K9 fused: masked spectrum -> inverse FFT -> overlap-add -> interleaved output, hop H = W/2.
// One workgroup produces RUN consecutive hops of every channel: hop h = second half of frame h-1 + first
// half of frame h, so it inverts frames h0-1 .. h0+RUN-1 and carries each channel's tail in LDS. The
// time-domain frames never touch HBM and every output sample is written exactly once, coalesced over
// the interleaved channels.
constexpr int kOlaRun = 8;          // hops per workgroup, about: launch_istft_ola fits the run to the launch (frames_per_workgroup)
template <int W>
struct InverseTables {
    FftTwiddles<W / 2> ft;
    float2 wk[(W / 2 + kFftThreads - 1) / kFftThreads];      // conj(exp(-2 pi i k / W)) for k = tid + 256*i
};

template <int W>
__device__ __forceinline__ void inverse_tables(InverseTables<W>& t, const float2* __restrict__ tw) {
    constexpr int N = W / 2;
    fft_twiddles<N, true>(t.ft, tw, W);
#pragma unroll
    for (int i = 0; i < (N + kFftThreads - 1) / kFftThreads; ++i) {
        const int k = threadIdx.x + kFftThreads * i;
        const float2 l = tw[k < N ? k : 0];
        t.wk[i] = (k < N) ? cconj(l) : make_float2(0.f, 0.f);
    }
}

template <int W>
struct SpectrumRegs { float2 xk[(W / 2 + kFftThreads - 1) / kFftThreads], xc[(W / 2 + kFftThreads - 1) / kFftThreads]; };

template <int W, bool MASKED>
__device__ __forceinline__ void fetch_spectrum(SpectrumRegs<W>& r, const float2* __restrict__ Y, const float* __restrict__ M) {
    constexpr int N = W / 2;
#pragma unroll
    for (int i = 0; i < (N + kFftThreads - 1) / kFftThreads; ++i) {
        const int k = threadIdx.x + kFftThreads * i;
        if (k < N) {
            float2 xk = Y[k], xc = Y[N - k];
            if constexpr (MASKED) {          // the products the mask kernels would have stored: same bits
                const float mk = M[k], mc = M[N - k];
                // rounded products, never contracted into the sums of the repack (then they would differ from the stored
                // ones in the last bit, and the stream from the offline result)
                xk = make_float2(mul_rounded(xk.x, mk), mul_rounded(xk.y, mk));
                xc = make_float2(mul_rounded(xc.x, mc), mul_rounded(xc.y, mc));
            }
            r.xk[i] = xk; r.xc[i] = xc;
        }
    }
}

// Hermitian repack of a fetched (masked) spectrum into the packed W/2-point transform (LDS); the registers are free
// for the next fetch afterwards.
template <int W>
__device__ __forceinline__ void repack_spectrum(const SpectrumRegs<W>& r, const InverseTables<W>& t, float2* buf0) {
    constexpr int N = W / 2;
#pragma unroll
    for (int i = 0; i < (N + kFftThreads - 1) / kFftThreads; ++i) {
        const int k = threadIdx.x + kFftThreads * i;
        if (k < N) {
            const float2 xk = r.xk[i];
            const float2 xc = cconj(r.xc[i]);
            const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
            const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
            // the complex product spelled out, one fused multiply-add per component: left to the compiler, the masked and
            // the plain instantiation of the kernel contract these sums differently and their outputs differ in the last bit
            const float2 w = t.wk[i];
            const float ox = fmaf(d.x, w.x, -mul_rounded(d.y, w.y));
            const float oy = fmaf(d.x, w.y, mul_rounded(d.y, w.x));
            buf0[k] = make_float2(e.x - oy, e.y + ox);
        }
    }
}

struct __attribute__((packed, aligned(4))) Float4U { float x, y, z, w; };    // a float4 at any dword address (segment offsets)

// accumulate_weighted: 0 = store, 1 = out += w y, 2 = out = w y (a class of segments that tiles its span and is the first
// to write there: no read of the cleared buffer)
template <int W, bool MASKED>
__global__ __launch_bounds__(kFftThreads) __attribute__((amdgpu_waves_per_eu(W <= 2048 ? REPET_ISTFT_MIN_WAVES : 1, 8))) void istft_ola_kernel(IstftOlaArgs a, int run) {
    constexpr int N = W / 2;          // complex points per frame = samples per hop (H = W/2)
    constexpr int HP = N / 2;         // float2 per half frame
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int C = a.n_channels;
    float2* tails = reinterpret_cast<float2*>(dyn);            // [C][HP] second half of the previous frame
    float* stage = dyn + (size_t)C * N;                        // [N samples][C] one hop, interleaved
    float* wts = stage + (size_t)C * N;                        // [N] cross-fade weights of the hop's samples (modes 1, 2)
    const int tid = threadIdx.x;
    const int64_t h0 = a.first_hop + (int64_t)blockIdx.x * run;
    const int64_t h_last = (h0 + run - 1 < a.last_hop) ? h0 + run - 1 : a.last_hop;
    const float inv_n = 1.0f / (float)N;
    FSTAMP_DECL
    InverseTables<W> tables;
    inverse_tables<W>(tables, a.twiddle);
    if (a.n_batch > 0) {
        const int j = a.batch_first + (int)blockIdx.y * a.batch_step;
        a.Y += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        if (MASKED) a.M += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        a.out_offset += (int64_t)j * a.batch_out_stride;
        a.fade_in = j > 0 ? a.overlap : 0;
        a.fade_out = a.overlap;
        a.seg_step = a.batch_out_stride;
        a.later = a.batch_total - 1 - j;
    }
    const int mode = a.accumulate_weighted;
    const bool narrow = a.n_out < (1 << 30) && a.seg_step < (1 << 30) && a.fade_in < (1 << 30) && a.fade_out < (1 << 30);
    const int fade_in = (int)a.fade_in, overlap = (int)a.fade_out, step = (int)a.seg_step;
    const float den_in = 1.0f / (float)(2 * a.fade_in), den_ov = 1.0f / (float)(2 * a.fade_out);       // (reciprocals: segment_weight32)
    // whole float4 groups of the interleaved hop (positions inside a segment in 32 bits; otherwise element by element)
    const int CT = a.out_channels > 0 ? a.out_channels : C;   // channels the output is interleaved over (a channel group: > C)
    const int cshift = ((mode != 0 && !narrow) || CT != C) ? -1 : C == 1 ? 0 : C == 2 ? 1 : C == 4 ? 2 : -1;
    const int n_groups = N * C / 4;

    // Measured and dropped (cfg 2 / 3 / 5): fetching the next spectrum while the current one is inverted, and fetching
    // the old output values of an accumulating hop ahead -- the 16 + 8 registers cost a resident workgroup per CU and
    // more than the latency they hide (0.076 / 0.512 / 0.679 ms against 0.071 / 0.488 / 0.600 without).
    SpectrumRegs<W> spec;
    FSTAMP(1, 0)                                               // prologue: twiddle registers
    for (int64_t t = h0 - 1; t <= h_last; ++t) {
        const bool have = t >= 0 && t < a.T;                   // hop T holds only the last frame's tail
        const bool emit = t >= h0;
        const int64_t n_base = t * N - a.trim;                 // output sample of stage[0]; hop t = [t N, (t+1) N) padded
        float* const dst0 = a.out + (a.out_offset + n_base) * CT + a.out_chan0;
        if (emit && mode != 0 && cshift >= 0) {                // read after the barriers of the channel loop
            for (int i = tid; i < N; i += kFftThreads) {
                const int64_t n = n_base + i;
                wts[i] = (n >= 0 && n < a.n_out) ? segment_weight32((int)n, fade_in, overlap, step, a.later, den_in, den_ov) : 0.f;
            }
        }
        for (int c = 0; c < C; ++c) {
            const float2* z = buf0;
            if (have) {
                fetch_spectrum<W, MASKED>(spec, a.Y + c * a.chan_stride + t * a.FS, MASKED ? a.M + c * a.chan_stride + t * a.FS : nullptr);
                repack_spectrum<W>(spec, tables, buf0);
                __syncthreads();
                z = fft_lds_regs<N, true>(buf0, buf1, tables.ft);
            }
            FSTAMP(1, 2)                                       // fetch + repack + inverse FFT
            for (int m = tid; m < HP; m += kFftThreads) {
                const float2 head = have ? z[m] : make_float2(0.f, 0.f);
                if (emit) {
                    const float2 tail = tails[c * HP + m];
                    stage[(2 * m) * C + c] = (head.x + tail.x) * inv_n;
                    stage[(2 * m + 1) * C + c] = (head.y + tail.y) * inv_n;
                }
                tails[c * HP + m] = have ? z[HP + m] : make_float2(0.f, 0.f);
            }
            __syncthreads();
            FSTAMP(1, 3)                                       // heads + tails into the hop image, barrier
        }
        if (!emit) continue;
        if (cshift >= 0) {
            for (int g = tid; g < n_groups; g += kFftThreads) {
                const int64_t n0 = n_base + ((4 * g) >> cshift), n1 = n0 + (4 >> cshift) - 1;
                const float4 raw = *reinterpret_cast<const float4*>(stage + 4 * g);
                float v[4] = {raw.x * a.scale, raw.y * a.scale, raw.z * a.scale, raw.w * a.scale};
                if (n0 >= 0 && n1 < a.n_out) {
                    Float4U* dst = reinterpret_cast<Float4U*>(dst0 + 4 * g);
                    if (mode != 0) {
                        const float* wg = wts + ((4 * g) >> cshift);
                        const float w[4] = {wg[0], wg[1 >> cshift], wg[2 >> cshift], wg[3 >> cshift]};
                        if (mode == 1) {
                            const Float4U o = *dst;
                            v[0] = o.x + w[0] * v[0]; v[1] = o.y + w[1] * v[1]; v[2] = o.z + w[2] * v[2]; v[3] = o.w + w[3] * v[3];
                        } else {
                            v[0] = w[0] * v[0]; v[1] = w[1] * v[1]; v[2] = w[2] * v[2]; v[3] = w[3] * v[3];
                        }
                    }
                    *dst = Float4U{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {              // the first and the last hop of a clip or segment
                        const int64_t n = n_base + ((4 * g + e) >> cshift);
                        if (n < 0 || n >= a.n_out) continue;
                        float* dst = dst0 + 4 * g + e;
                        const float w = mode != 0 ? wts[(4 * g + e) >> cshift] : 1.f;
                        if (mode == 1) *dst += w * v[e];
                        else *dst = w * v[e];
                    }
                }
            }
        } else {
            for (int i = tid; i < N * C; i += kFftThreads) {
                const int64_t n = n_base + i / C;
                if (n < 0 || n >= a.n_out) continue;
                const float v = stage[i] * a.scale;
                float* dst = dst0 + (int64_t)(i / C) * CT + (i % C);
                const float w = mode != 0 ? segment_weight(n, a.fade_in, a.fade_out, a.seg_step, a.later) : 1.f;
                if (mode == 1) *dst += w * v;
                else *dst = mode == 2 ? w * v : v;
            }
        }
        __syncthreads();
        FSTAMP(1, 4)                                           // the hop written (or accumulated), barrier
    }
}

// =====================================================================================================
// Wave-synchronous variants (default for W <= 4096): ONE 64-lane wavefront owns one transform in LDS, so
// the radix-4 stages need no workgroup barrier -- LDS instructions of a wave execute in order, a stage
// reads all its operands into registers before it writes, and a compiler-level fence between stages keeps
// hipcc from caching LDS values across the cross-lane exchange. The twiddle table exp(-2 pi i m / W) and the
// window sit in LDS, four transforms (frames x channels) are in flight per 256-thread workgroup.
// =====================================================================================================
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// In-place Stockham FFT of N points in `buf` (LDS) by one wave. tw[m] = exp(-2 pi i m / (2N)), m < 2N (LDS).
template <int N, bool INVERSE>
__device__ __forceinline__ void wave_fft(float2* buf, const float2* tw, int lane) {
    constexpr int Q4 = (N / 4 + 63) / 64;       // radix-4 butterflies per lane and stage
    constexpr int Q2 = (N / 2 + 63) / 64;
    for (int p = 1; p < N;) {
        if (N / p >= 4) {
            const int tstep = 2 * (N / (p * 4));        // table has 2N entries
            float2 u[Q4][4];
#pragma unroll
            for (int b = 0; b < Q4; ++b) {
                const int i = lane + 64 * b;
                if (i < N / 4) { u[b][0] = buf[i]; u[b][1] = buf[i + N / 4]; u[b][2] = buf[i + N / 2]; u[b][3] = buf[i + 3 * N / 4]; }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < Q4; ++b) {
                const int i = lane + 64 * b;
                if (i < N / 4) {
                    const int k = i & (p - 1);
                    const int j = ((i - k) << 2) + k;
                    float2 u0 = u[b][0], u1 = u[b][1], u2 = u[b][2], u3 = u[b][3];
                    if (p > 1) {
                        float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                        if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
                        u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
                    }
                    const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3);
                    const float2 d = csub(u1, u3);
                    const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                    buf[j] = cadd(t0, t2);
                    buf[j + p] = cadd(t1, t3);
                    buf[j + 2 * p] = csub(t0, t2);
                    buf[j + 3 * p] = csub(t1, t3);
                }
            }
            p *= 4;
        } else {
            const int tstep = 2 * (N / (p * 2));
            float2 u[Q2][2];
#pragma unroll
            for (int b = 0; b < Q2; ++b) {
                const int i = lane + 64 * b;
                if (i < N / 2) { u[b][0] = buf[i]; u[b][1] = buf[i + N / 2]; }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < Q2; ++b) {
                const int i = lane + 64 * b;
                if (i < N / 2) {
                    const int k = i & (p - 1);
                    const int j = ((i - k) << 1) + k;
                    float2 w1 = tw[k * tstep];
                    if (INVERSE) w1 = cconj(w1);
                    const float2 u1 = cmul(u[b][1], w1);
                    buf[j] = cadd(u[b][0], u1);
                    buf[j + p] = csub(u[b][0], u1);
                }
            }
            p *= 2;
        }
        wave_sync();
    }
}

// LDS layout of the wave kernels (float2 units): tw[W] | win[N] (w[2n], w[2n+1]) | buf[4][N] | extra
template <int W>
__global__ __launch_bounds__(256) void stft_wave_kernel(StftArgs a, int frames_per_wg) {
    constexpr int N = W / 2;
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* tw = lds2;
    float2* win = lds2 + W;
    float2* bufs = win + N;
    float* vst = reinterpret_cast<float*>(bufs + 4 * N);     // [jobs per round][FS] magnitudes
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.n_channels;
    const int64_t b = blockIdx.y;
    a.sample_offset += b * a.batch_sample_stride;
    a.X += b * a.batch_spec_stride;
    a.V += b * a.batch_spec_stride;
    if (a.Vm) a.Vm += b * a.batch_mean_stride;
    if (a.Vn) a.Vn += b * a.batch_mean_stride;
    if (a.Vh) a.Vh = static_cast<_Float16*>(a.Vh) + 2 * b * a.batch_mean_stride;
    if (a.P) a.P += b * a.batch_mean_stride;

    for (int i = tid; i < W; i += 256) tw[i] = a.twiddle[i];
    for (int i = tid; i < N; i += 256) win[i] = make_float2(a.window[2 * i], a.window[2 * i + 1]);
    __syncthreads();

    const int FI = C >= 4 ? 1 : 4 / C;          // frames per round (C = 3: one frame, three waves busy)
    const int JR = FI * C;                      // transforms per round (<= 8)
    const int64_t t_begin = (int64_t)blockIdx.x * frames_per_wg;
    const int64_t t_end = (t_begin + frames_per_wg < a.T) ? t_begin + frames_per_wg : a.T;
    float2* buf = bufs + wave * N;
    const bool want_mean = a.Vm || a.Vn || a.P;

    for (int64_t t0 = t_begin; t0 < t_end; t0 += FI) {
        for (int job = wave; job < JR; job += 4) {
            const int df = job / C, c = job - df * C;
            const int64_t t = t0 + df;
            if (t >= t_end) continue;
            const int64_t start = t * a.H - (a.centred ? W / 2 : 0);
            for (int n = lane; n < N; n += 64) {
                const int64_t s0 = start + 2 * n, s1 = s0 + 1;
                float x0 = 0.f, x1 = 0.f;
                if (s0 >= 0 && s0 < a.n_samples) x0 = a.audio[(a.sample_offset + s0) * C + c];
                if (s1 >= 0 && s1 < a.n_samples) x1 = a.audio[(a.sample_offset + s1) * C + c];
                const float2 w = win[n];
                buf[n] = make_float2(x0 * w.x, x1 * w.y);
            }
            wave_sync();
            wave_fft<N, false>(buf, tw, lane);
            const int64_t row = t * a.FS;
            float2* Xrow = a.X + c * a.chan_stride + row;
            float* Vrow = a.V + c * a.chan_stride + row;
            float* vrow = vst + job * a.FS;
            for (int k = lane; k <= N; k += 64) {
                const float2 zk = buf[k & (N - 1)];
                const float2 zc = cconj(buf[(N - k) & (N - 1)]);
                const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
                const float2 d = csub(zk, zc);
                const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);   // (zk - zc) / (2i)
                const float2 x = cadd(e, cmul(tw[k], o));
                const float mag = magnitude(x);
                Xrow[k] = x;
                Vrow[k] = mag;
                vrow[k] = mag;
            }
            if (lane < a.FS - (N + 1)) {      // zero the pad bins [F, FS)
                Xrow[N + 1 + lane] = make_float2(0.f, 0.f);
                Vrow[N + 1 + lane] = 0.f;
            }
            wave_sync();                      // buf is refilled by this wave's next job
        }
        if (!want_mean) continue;
        __syncthreads();
        for (int df = 0; df < FI; ++df) {
            const int64_t t = t0 + df;
            if (t >= t_end) break;
            const int64_t row = t * a.FS;
            // channel mean (repet.py:162,:667) and its L2 norm over frequency (repet.py:1220)
            float ss = 0.f;
            for (int k = tid; k <= N; k += 256) {
                float m = 0.f;
                for (int c = 0; c < C; ++c) m += vst[(df * C + c) * a.FS + k];
                if (C > 1) m *= 1.0f / (float)C;
                vst[(df * C) * a.FS + k] = m;                 // channel 0's slot now holds the mean
                ss += m * m;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));   // 0 for a silent frame: NaN row
            for (int k = tid; k < a.FS; k += 256) {
                const float m = (k <= N) ? vst[(df * C) * a.FS + k] : 0.f;
                if (a.Vm) a.Vm[row + k] = m;
                if (a.Vn) a.Vn[row + k] = (k <= N) ? m / norm : 0.f;
                if (a.Vh) store_split_f16(a.Vh, row + k, (k <= N) ? m / norm : 0.f);
                if (a.P) a.P[row + k] = m * m;
            }
            __syncthreads();                  // red[] and vst are reused
        }
    }
}

// Fused inverse: RUN hops per workgroup, FI frames x C channels inverted per round by the four waves,
// then all threads add heads and tails and write whole hops, interleaved over the channels.
template <int W>
__global__ __launch_bounds__(256) void istft_ola_wave_kernel(IstftOlaArgs a, int run) {
    constexpr int N = W / 2;          // samples per hop
    extern __shared__ __attribute__((aligned(16))) float2 lds2[];
    float2* tw = lds2;
    float2* bufs = lds2 + W;                                        // [4][N] complex = [4][W] samples
    float* tails = reinterpret_cast<float*>(bufs + 4 * N);          // [C][N] second half of the previous frame
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.n_channels;
    if (a.n_batch > 0) {
        const int j = a.batch_first + (int)blockIdx.y * a.batch_step;
        a.Y += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        if (a.M) a.M += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        a.out_offset += (int64_t)j * a.batch_out_stride;
        a.fade_in = j > 0 ? a.overlap : 0;
        a.fade_out = a.overlap;
        a.seg_step = a.batch_out_stride;
        a.later = a.batch_total - 1 - j;
    }
    for (int i = tid; i < W; i += 256) tw[i] = a.twiddle[i];
    for (int i = tid; i < C * N; i += 256) tails[i] = 0.f;
    __syncthreads();

    const int FI = 4 / C;                       // frames per round (C <= 4)
    const int64_t h0 = a.first_hop + (int64_t)blockIdx.x * run;
    int64_t h1 = h0 + run - 1;
    if (h1 > a.last_hop) h1 = a.last_hop;
    float2* buf = bufs + wave * N;
    const float inv_n = a.scale / (float)N;
    // frames h0-1 .. h1 ; frame t feeds hop t (first half) and hop t+1 (second half)
    for (int64_t t0 = h0 - 1; t0 <= h1; t0 += FI) {
        if (wave < FI * C) {
            const int df = wave / C, c = wave - df * C;
            const int64_t t = t0 + df;
            if (t >= 0 && t < a.T && t <= h1) {
                const float2* Y = a.Y + c * a.chan_stride + t * a.FS;
                const float* Mr = a.M ? a.M + c * a.chan_stride + t * a.FS : nullptr;
                for (int k = lane; k < N; k += 64) {
                    float2 xk = Y[k];
                    float2 xc = cconj(Y[N - k]);
                    if (Mr) {
                        const float mk = Mr[k], mc = Mr[N - k];
                        xk = make_float2(mul_rounded(xk.x, mk), mul_rounded(xk.y, mk));
                        xc = make_float2(mul_rounded(xc.x, mc), mul_rounded(xc.y, mc));
                    }
                    const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
                    const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
                    const float2 o = cmul(d, cconj(tw[k]));
                    buf[k] = make_float2(e.x - o.y, e.y + o.x);
                }
                wave_sync();
                wave_fft<N, true>(buf, tw, lane);
            } else {
                for (int k = lane; k < N; k += 64) buf[k] = make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        const float* samples = reinterpret_cast<const float*>(bufs);   // [4][W]
        for (int i = tid; i < N * C; i += 256) {
            const int sidx = i / C, c = i - sidx * C;
            float prev = tails[c * N + sidx];
            for (int df = 0; df < FI; ++df) {
                const int64_t h = t0 + df;                     // hop fed by the first half of frame t0+df
                const float* fr = samples + (df * C + c) * W;
                const float v = (prev + fr[sidx]) * inv_n;
                prev = fr[N + sidx];
                if (h < h0 || h > h1) continue;
                const int64_t n = h * N - a.trim + sidx;
                if (n < 0 || n >= a.n_out) continue;
                float* dst = a.out + (a.out_offset + n) * C + c;
                if (a.accumulate_weighted) {
                    const float w = segment_weight(n, a.fade_in, a.fade_out, a.seg_step, a.later);
                    *dst += w * v;
                } else {
                    *dst = v;
                }
            }
            tails[c * N + sidx] = prev;
        }
        __syncthreads();
    }
}

template <typename Fn>
static hipError_t dispatch_window(int W, Fn&& fn) {
    switch (W) {
        case 64: fn(std::integral_constant<int, 64>{}); break;
        case 128: fn(std::integral_constant<int, 128>{}); break;
        case 256: fn(std::integral_constant<int, 256>{}); break;
        case 512: fn(std::integral_constant<int, 512>{}); break;
        case 1024: fn(std::integral_constant<int, 1024>{}); break;
        case 2048: fn(std::integral_constant<int, 2048>{}); break;
        case 4096: fn(std::integral_constant<int, 4096>{}); break;
        case 8192: fn(std::integral_constant<int, 8192>{}); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// Frames (or hops) per workgroup. A launch runs in rounds of the device's resident workgroup slots and a workgroup's time
// is its run plus a fixed part (tables, the frame before an overlap-add run), so the run is chosen to minimise
// rounds x (run + fixed): cfg 2's forward STFT was 1 939 workgroups of 4 frames on 768 slots -- three rounds, the last
// half empty; 705 workgroups of 11 frames are one.
static int frames_per_workgroup(const void* kernel, size_t dynamic_lds, int64_t units, int64_t batches, int least, double fixed) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, int> slots_of;
    int slots;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = slots_of.find({kernel, dynamic_lds});
        if (it == slots_of.end()) {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop{};
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kFftThreads, dynamic_lds) != hipSuccess || per_cu < 1) per_cu = 2;
            // the occupancy query divides 160 KB by the bytes asked for; the hardware hands LDS out in 1 280-byte granules
            // (32 768 bytes take 26 of the 128: four workgroups per CU, not five -- measured on the inverse kernel)
            hipFuncAttributes fa{};
            if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) {
                const size_t lds = round_up((int64_t)(fa.sharedSizeBytes + dynamic_lds), 1280);
                if (lds > 0 && (int)(163840 / lds) < per_cu) per_cu = (int)(163840 / lds) > 0 ? (int)(163840 / lds) : 1;
            }
            if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 1) prop.multiProcessorCount = 256;
            it = slots_of.emplace(std::make_pair(kernel, dynamic_lds), per_cu * prop.multiProcessorCount).first;
        }
        slots = it->second;
    }
    int best = least;
    double best_cost = 0;
    for (int run = least; run <= 8 * least; ++run) {
        const int64_t wgs = ceil_div(units, run) * batches;
        const double cost = (double)ceil_div(wgs, slots) * (run + fixed);
        if (run == least || cost < best_cost * 0.98) { best = run; best_cost = cost; }     // longer runs only for a real gain
    }
    return best;
}

static bool use_wave_kernels() {
    static const bool on = [] { const char* e = getenv("REPET_FFT_PATH"); return e && e[0] == 'w'; }();   // "wave" selects the wave-synchronous kernels (slower today: see DESIGN.md)
    return on;
}
constexpr int kStftFramesPerWg = 8;
constexpr int kOlaWaveRun = 15;      // + the frame before = 16 frames per workgroup

// Strict reference mode (REPET_FLAG_STRICT_REFERENCE): a frame that holds an INFINITE sample. What repet.py makes of it is
// decided inside pocketfft: every bin has a component at +-inf by the transform's definition, np.abs of (inf, NaN) is inf, and
// an inf magnitude is an ordinary, largest member of a median -- but where the butterflies meet inf - inf BOTH components turn
// NaN, and that depends on the sample's position in the frame (measured, 24-s clip, `original`: -inf at one position gives NaN
// in its own three frames only, +inf at another gives NaN at that position of all 23 periods, exactly like a NaN sample). The
// butterflies here meet inf - inf elsewhere. Instead of one arbitrary mix for another, an infinite sample is TREATED AS NaN:
// this pass, run behind the forward kernel only when the uploaded host array held samples that are not finite, makes the
// spectrum and the magnitudes of such a frame NaN in every bin. For `sim` / `simonline` that is the reference's result (NaN on
// the frame's samples, nothing else changes); for the period family it is the reference's result where pocketfft produced NaN
// bins and a superset of its NaN samples where it did not. One wavefront per frame and channel.
__global__ __launch_bounds__(64) void infinite_frames_kernel(StftArgs a) {
    const int64_t t = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x, C = a.n_channels, F = a.W / 2 + 1;
    const int64_t start = t * a.H - (a.centred ? a.W / 2 : 0);
    const float* clip = a.audio + (a.sample_offset + b * a.batch_sample_stride) * C;
    for (int c = 0; c < C; ++c) {
        bool has_nan = false, has_inf = false;
        for (int i = lane; i < a.W; i += 64) {
            const int64_t sidx = start + i;
            if (sidx < 0 || sidx >= a.n_samples) continue;
            const float x = clip[sidx * C + c];
            has_nan |= x != x;
            has_inf |= fabsf(x) == INFINITY;
        }
        const bool any_nan = __ballot(has_nan) != 0, any_inf = __ballot(has_inf) != 0;
        if (any_nan || !any_inf) continue;                   // (a NaN sample has made every bin NaN already)
        float2* Xrow = a.X + b * a.batch_spec_stride + c * a.chan_stride + t * a.FS;
        float* Vrow = a.V + b * a.batch_spec_stride + c * a.chan_stride + t * a.FS;
        const float nanf_ = __uint_as_float(0x7fc00000u);
        for (int f = lane; f < a.FS; f += 64) {
            Vrow[f] = f < F ? nanf_ : 0.f;
            Xrow[f] = f < F ? make_float2(nanf_, nanf_) : make_float2(0.f, 0.f);
        }
    }
}
hipError_t launch_infinite_frames_fix(const StftArgs& a, hipStream_t s) {
    if (a.T <= 0) return hipSuccess;
    hipLaunchKernelGGL(infinite_frames_kernel, dim3((unsigned)a.T, (unsigned)(a.n_batch > 0 ? a.n_batch : 1)), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_stft(const StftArgs& a, hipStream_t s) {
    if (a.T <= 0) return hipSuccess;
    // (also for the handful of frames of a streaming push: the stream's output must equal the offline result bit for bit)
    if (reg_fft_supported(a.W, a.n_channels, false)) return launch_stft_reg(a, s);
    if (use_wave_kernels() && a.W <= 4096 && a.n_channels <= 8) {
        const int C = a.n_channels, N = a.W / 2;
        const int FI = C >= 4 ? 1 : 4 / C;
        const size_t dyn = (size_t)(a.W + N + 4 * N) * sizeof(float2) + (size_t)FI * C * a.FS * sizeof(float);
        const int fpw = (int)round_up(kStftFramesPerWg, FI);
        return dispatch_window(a.W, [&](auto w) {
            constexpr int Wc = decltype(w)::value;
            if constexpr (Wc <= 4096) {
                (void)ensure_dynamic_lds(reinterpret_cast<const void*>(&stft_wave_kernel<Wc>), (int)dyn);
                hipLaunchKernelGGL(stft_wave_kernel<Wc>, dim3((unsigned)ceil_div(a.T, fpw), (unsigned)(a.n_batch > 0 ? a.n_batch : 1)),
                                   dim3(256), dyn, s, a, fpw);
            }
        });
    }
    const int64_t batches = a.n_batch > 0 ? a.n_batch : 1;
    if ((a.n_channels % 2) == 0 && a.W <= 2048) {
        return dispatch_window(a.W, [&](auto w) {
            constexpr int Wc = decltype(w)::value;
            if constexpr (Wc <= 2048) {
                const int run = frames_per_workgroup(reinterpret_cast<const void*>(&stft_pair_kernel<Wc>), 0, a.T, batches, kStftFrameRun, 0.35);
                hipLaunchKernelGGL(stft_pair_kernel<Wc>, dim3((unsigned)ceil_div(a.T, run), (unsigned)batches), dim3(kFftThreads), 0, s, a, run);
            }
        });
    }
    return dispatch_window(a.W, [&](auto w) {
        constexpr int Wc = decltype(w)::value;
        const int run = frames_per_workgroup(reinterpret_cast<const void*>(&stft_kernel<Wc>), 0, a.T, batches, kStftFrameRun, 0.35);
        hipLaunchKernelGGL(stft_kernel<Wc>, dim3((unsigned)ceil_div(a.T, run), (unsigned)batches), dim3(kFftThreads), 0, s, a, run);
    });
}

hipError_t launch_istft_frames(const IstftArgs& a, hipStream_t s) {
    if (a.T <= 0) return hipSuccess;
    return dispatch_window(a.W, [&](auto w) {
        hipLaunchKernelGGL(istft_frames_kernel<decltype(w)::value>, dim3((unsigned)a.T, (unsigned)a.n_channels),
                           dim3(kFftThreads), 0, s, a);
    });
}

hipError_t launch_istft_ola(const IstftOlaArgs& a0, hipStream_t s) {
    IstftOlaArgs a = a0;
    if (a.T <= 0 || a.n_out <= 0) return hipSuccess;
    // hops that intersect [trim, trim + n_out): first = floor(trim / N), last = floor((trim + n_out - 1) / N)
    const int N = a.W / 2;
    a.first_hop = a.trim / N;
    a.last_hop = (a.trim + a.n_out - 1) / N;
    if (a.last_hop > a.T) a.last_hop = a.T;                  // hop T holds the last frame's tail, later hops are empty
    const int64_t hops = a.last_hop - a.first_hop + 1;
    if (hops <= 0) return hipSuccess;
    if (a.out_channels == 0 && reg_fft_supported(a.W, a.n_channels, true)) {
        const hipError_t e = launch_istft_ola_reg(a, hops, s);
        if (e != hipErrorNotSupported) return e;
    }
    // Only the register kernel applies a repeating-segment MODEL itself (IstftOlaArgs::model with M == nullptr); the kernels
    // below would emit the unmasked mixture without a word. The engine asks istft_reg_takes() before it chooses the model
    // form; should the two ever disagree, this is an error, not a silent wrong answer.
    if (a.model) return hipErrorInvalidValue;
    if (a.out_channels == 0 && use_wave_kernels() && a.W <= 4096 && a.n_channels <= 4 && a.n_channels != 3) {
        const int C = a.n_channels;
        const int FI = 4 / C;
        const int run = (int)round_up(kOlaWaveRun + 1, FI) - 1;
        const size_t dynw = (size_t)(a.W + 4 * N) * sizeof(float2) + (size_t)C * N * sizeof(float);
        return dispatch_window(a.W, [&](auto w) {
            constexpr int Wc = decltype(w)::value;
            if constexpr (Wc <= 4096) {
                (void)ensure_dynamic_lds(reinterpret_cast<const void*>(&istft_ola_wave_kernel<Wc>), (int)dynw);
                hipLaunchKernelGGL(istft_ola_wave_kernel<Wc>, dim3((unsigned)ceil_div(hops, run), (unsigned)(a.n_batch > 0 ? a.n_batch : 1)),
                                   dim3(256), dynw, s, a, run);
            }
        });
    }
    // tails [C][N/2] float2 + stage [N][C] (+ weights [N] for the cross-faded segments of `extended`)
    const size_t dyn = (size_t)a.n_channels * N * sizeof(float) * 2 + (a.accumulate_weighted ? (size_t)N * sizeof(float) : 0);
    if (dyn > 96 * 1024) {
        // more channels than one workgroup's LDS holds a frame tail and a hop image for (repet.py:152,179 loop over any
        // number): groups of channels, one launch each, into their places of the interleaved output
        const int per = (int)((96 * 1024 - (a.accumulate_weighted ? (size_t)N * sizeof(float) : 0)) / ((size_t)N * sizeof(float) * 2));
        if (per < 1 || a.out_channels > 0) return hipErrorInvalidValue;
        for (int c0 = 0; c0 < a.n_channels; c0 += per) {
            IstftOlaArgs g = a0;
            g.n_channels = std::min(per, a0.n_channels - c0);
            g.Y = a0.Y + (int64_t)c0 * a0.chan_stride;
            if (a0.M) g.M = a0.M + (int64_t)c0 * a0.chan_stride;
            g.out_channels = a0.n_channels; g.out_chan0 = c0;
            const hipError_t e = launch_istft_ola(g, s);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    return dispatch_window(a.W, [&](auto w) {
        constexpr int Wc = decltype(w)::value;
        const int64_t batches = a.n_batch > 0 ? a.n_batch : 1;
        auto go = [&](auto kernel) {
            (void)ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)dyn);
            // a run of r hops costs r + 1 inversions (the frame before it) and the twiddle prologue
            const int run = frames_per_workgroup(reinterpret_cast<const void*>(kernel), dyn, hops, batches, kOlaRun - 2, 1.5);
            hipLaunchKernelGGL(kernel, dim3((unsigned)ceil_div(hops, run), (unsigned)batches), dim3(kFftThreads), dyn, s, a, run);
        };
        if constexpr (Wc <= 4096) {
            if (a.M) go(&istft_ola_kernel<Wc, true>); else go(&istft_ola_kernel<Wc, false>);
        } else {
            go(&istft_ola_kernel<Wc, false>);       // the engine keeps the mask in X for the longest window (registers)
        }
    });
}

hipError_t launch_overlap_add(const OlaArgs& a, hipStream_t s) {
    const int64_t total = a.n_out * a.n_channels;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- elementwise helpers --------------------------------------------------------------------------
template <typename T>
__global__ void convert_in_kernel(const T* src, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}
__global__ void convert_out_kernel(const float* src, double* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}
__global__ void square_kernel(const float* src, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i] * src[i];
}
// dst[t][0..FS) = src[t][0..F) / ||src[t]||, pad bins zero (repet.py:1220 for a caller-supplied matrix)
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* src, float* dst, int F, int FS) {
    __shared__ float red[4];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    float ss = 0.f;
    for (int k = tid; k < F; k += 256) { const float v = src[t * F + k]; ss += v * v; }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int k = tid; k < FS; k += 256) dst[t * FS + k] = (k < F) ? src[t * F + k] / norm : 0.f;
}

// foreground = audio - background (README.md:69), as float64, and the channel mean of a signal (README.md:79)
__global__ void foreground_kernel(const float* audio, const float* background, double* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)audio[i] - (double)background[i];
}
// which: 0 mixture, 1 background, 2 foreground
__global__ void channel_mean_kernel(const float* audio, const float* background, int which, int C, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = audio[i * C + c], b = background[i * C + c];
            acc += which == 0 ? a : (which == 1 ? b : a - b);
        }
        dst[i] = acc / (float)C;
    }
}

static unsigned stream_grid(int64_t n) {
    int64_t g = ceil_div(n, 256);
    return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

// float64 -> the fp32 sample and its fp32 remainder (what hostio.hip's split_part does on the host)
__global__ void convert_in_split_kernel(const double* src, float* dst, float* dst_lo, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double x = src[i];
        const float h = (float)x;
        dst[i] = h;
        dst_lo[i] = (float)(x - (double)h);
    }
}
hipError_t launch_convert_in(const void* src, int dtype, float* dst, int64_t n, hipStream_t s, float* dst_lo) {
    if (n <= 0) return hipSuccess;
    if (dst_lo) {
        if (dtype == 1) {
            hipLaunchKernelGGL(convert_in_split_kernel, dim3(stream_grid(n)), dim3(256), 0, s, (const double*)src, dst, dst_lo, n);
            return hipGetLastError();
        }
        const hipError_t e = hipMemsetAsync(dst_lo, 0, (size_t)n * sizeof(float), s);       // exact in fp32: no remainder
        if (e != hipSuccess) return e;
    }
    switch (dtype) {
        case 0: hipLaunchKernelGGL(convert_in_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, s, (const float*)src, dst, n); break;
        case 1: hipLaunchKernelGGL(convert_in_kernel<double>, dim3(stream_grid(n)), dim3(256), 0, s, (const double*)src, dst, n); break;
        case 2: hipLaunchKernelGGL(convert_in_kernel<int16_t>, dim3(stream_grid(n)), dim3(256), 0, s, (const int16_t*)src, dst, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_convert_out(const float* src, double* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(convert_out_kernel, dim3(stream_grid(n)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_foreground(const float* audio, const float* background, double* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(foreground_kernel, dim3(stream_grid(n)), dim3(256), 0, s, audio, background, dst, n);
    return hipGetLastError();
}
__global__ void foreground_f32_kernel(const float* audio, const float* background, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)((double)audio[i] - (double)background[i]);
}
hipError_t launch_foreground_f32(const float* audio, const float* background, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(foreground_f32_kernel, dim3(stream_grid(n)), dim3(256), 0, s, audio, background, dst, n);
    return hipGetLastError();
}
hipError_t launch_channel_mean(const float* audio, const float* background, int which, int C, float* dst, int64_t n,
                               hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(channel_mean_kernel, dim3(stream_grid(n)), dim3(256), 0, s, audio, background, which, C, dst, n);
    return hipGetLastError();
}
hipError_t launch_square(const float* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(square_kernel, dim3(stream_grid(n)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_unit_rows(const float* src, float* dst, int64_t T, int32_t F, int32_t FS, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)T), dim3(256), 0, s, src, dst, F, FS);
    return hipGetLastError();
}

}  // namespace repet

#ifdef REPET_FFT_STAMPS
extern "C" int repet_debug_fft_stamps(unsigned long long* out, int clear) {
    if (clear) {
        static unsigned long long zeros[2 * 8 * 8];
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(repet::g_fft_stamps), zeros, sizeof(zeros));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(repet::g_fft_stamps), sizeof(unsigned long long) * 2 * 8 * 8);
}
#endif
