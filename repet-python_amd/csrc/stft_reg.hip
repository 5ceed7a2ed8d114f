// K1r / K9r: register-resident STFT and inverse for the 2048-sample window (44.1 / 48 kHz), gfx950.
//
// One WAVEFRONT owns one transform: the 1024-point complex FFT behind a real 2048-sample frame lives in 16
// float2 registers per lane and is computed as 16 x 16 x 4 (Cooley-Tukey, decimation in time):
//   n = 64 n1 + n2,  k = k1 + 16 (j1 + 16 j2),  n2 = 4 m1 + m2
//   A: lane n2 holds x[64 n1 + n2]            -> DFT16 over n1, times W_1024^(n2 k1)
//   B: lane (k1, m2) holds those for n2 = 4 m1 + m2 -> DFT16 over m1, times W_64^(m2 j1)
//   C: lane (u, k1) holds (k1, j1 = 4 i + u, m2)    -> DFT4 over m2
// and lane l ends with X[l + 64 s], s < 16 -- the distribution it started with. The two transposes go through
// a private 10 KB LDS region per wave with conflict-free pitches (68 and 80/20, see the bank rules in
// MI355X_MICROARCH.md: ds_write_b64 = 16-lane groups mod 32 banks, ds_read_b64 = 32-lane groups mod 64), and
// there is no workgroup barrier anywhere in the forward kernel: the block-wide Stockham version (stft.hip)
// spends 7 barriers per transform. The Hermitian split needs Z[N-k], which sits in lane 64-l, slot 15-s:
// one ds_bpermute per component. Same arithmetic contract as stft.hip (repet.py:1001-1105, :158, :1220).
#include "common.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace repet {

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

__device__ __forceinline__ void wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <bool INV>
__device__ __forceinline__ void dft4(float2& a, float2& b, float2& c, float2& d) {
    const float2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), e = csub(b, d);
    const float2 t3 = INV ? make_float2(-e.y, e.x) : make_float2(e.y, -e.x);   // -i e (forward), +i e (inverse)
    a = cadd(t0, t2); b = cadd(t1, t3); c = csub(t0, t2); d = csub(t1, t3);
}

// 16-point DFT in place: input v[n] natural (n = 4p + q); output X[k] in v[tr16(k)].
__host__ __device__ constexpr int tr16(int k) { return (k >> 2) + 4 * (k & 3); }

template <bool INV>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4<INV>(v[q], v[q + 4], v[q + 8], v[q + 12]);      // v[q + 4r] = y[q][r]
    // y[q][r] *= W_16^(q r)
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    auto mulw = [](float2 x, float wr, float wi) {      // x * (wr - i wi) forward, (wr + i wi) inverse
        const float im = INV ? wi : -wi;
        return make_float2(x.x * wr - x.y * im, x.x * im + x.y * wr);
    };
    v[1 + 4] = mulw(v[1 + 4], c1, s1);        // q r = 1
    v[1 + 8] = mulw(v[1 + 8], h, h);          // 2
    v[1 + 12] = mulw(v[1 + 12], s1, c1);      // 3
    v[2 + 4] = mulw(v[2 + 4], h, h);          // 2
    v[2 + 8] = mulw(v[2 + 8], 0.f, 1.f);      // 4: -i
    v[2 + 12] = mulw(v[2 + 12], -h, h);       // 6
    v[3 + 4] = mulw(v[3 + 4], s1, c1);        // 3
    v[3 + 8] = mulw(v[3 + 8], -h, h);         // 6
    v[3 + 12] = mulw(v[3 + 12], -c1, -s1);    // 9
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4<INV>(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);   // v[s + 4r] = X[r + 4s]
}

#ifdef REPET_FFT_STAMPS
__device__ unsigned long long g_reg_stamps[8 * 8];         // [sampled wave][phase] summed cycles
#define RSTAMP_DECL unsigned long long rst_prev = __builtin_amdgcn_s_memtime(); const bool rst_on = (threadIdx.x & 63) == 0 && blockIdx.y == 0 && (blockIdx.x % 61) == 7 && blockIdx.x / 61 < 8 && threadIdx.x < 64;
#define RSTAMP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (rst_on) g_reg_stamps[(blockIdx.x / 61) * 8 + (k)] += now_ - rst_prev; rst_prev = now_; }
#else
#define RSTAMP_DECL
#define RSTAMP(k)
#endif

constexpr int kRegN = 1024;          // complex FFT length
constexpr int kExPitch = 1280;       // float2 per wave: max(16 * 68, 16 * 80)

// Stage twiddles, shared by the four waves of a workgroup, in LDS: a[k1][lane] = W_1024^(lane k1) (lane-contiguous,
// conflict-free) and b[m2][j1] = W_64^(m2 j1) (four addresses per read, broadcast). 8.2 KB.
constexpr int kTwFloat2 = 16 * 64 + 64;
struct RegTwiddles { const float2* a; const float2* b; };

template <bool INV>
__device__ __forceinline__ RegTwiddles load_reg_twiddles(float2* lds, const float2* __restrict__ tw2048, int tid) {
    for (int i = tid; i < 16 * 64; i += 256) {
        const int k1 = i >> 6, l = i & 63;
        const float2 w = tw2048[2 * l * k1];                  // exp(-2 pi i l k1 / 1024)
        lds[i] = INV ? cconj(w) : w;
    }
    if (tid < 64) {
        const int m2 = tid >> 4, j1 = tid & 15;
        const float2 w = tw2048[32 * m2 * j1];                // exp(-2 pi i m2 j1 / 64)
        lds[16 * 64 + tid] = INV ? cconj(w) : w;
    }
    return RegTwiddles{lds, lds + 16 * 64};
}

// v[n1] = x[64 n1 + lane] in; v[s] = X[lane + 64 s] out (unscaled). `ex` is this wave's private LDS region.
template <bool INV>
__device__ __forceinline__ void wave_fft1024(float2 (&v)[16], float2* ex, const RegTwiddles& t, int lane) {
    dft16<INV>(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const float2 x = v[tr16(k1)];
        ex[68 * k1 + lane] = (k1 == 0) ? x : cmul(x, t.a[64 * k1 + lane]);
    }
    wave_fence();
    {
        const float2* src = ex + 68 * (lane >> 2) + (lane & 3);
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) v[m1] = src[4 * m1];
    }
    wave_fence();
    dft16<INV>(v);
    {
        float2* dst = ex + (lane >> 2) + 20 * (lane & 3);
#pragma unroll
        for (int j1 = 0; j1 < 16; ++j1) {
            const float2 x = v[tr16(j1)];
            dst[80 * j1] = (j1 == 0) ? x : cmul(x, t.b[16 * (lane & 3) + j1]);
        }
    }
    wave_fence();
    {
        const float2* src = ex + 80 * (lane >> 4) + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2 c0 = src[320 * i], c1 = src[320 * i + 20], c2 = src[320 * i + 40], c3 = src[320 * i + 60];
            dft4<INV>(c0, c1, c2, c3);
            v[i] = c0; v[i + 4] = c1; v[i + 8] = c2; v[i + 12] = c3;       // slot s = i + 4 j2
        }
    }
    wave_fence();
}

__device__ __forceinline__ float lane_fetch(float x, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(x)));
}

// ---- forward ---------------------------------------------------------------------------------------
// Twelve wavefronts per workgroup, ONE workgroup per CU (147 KB of LDS: twelve private exchange regions and the tables
// every wave reads -- stage twiddles, window, split twiddles). A wave takes whole frames (every channel of frame u,
// u = wave id + k x waves of the launch, frames of a batch numbered through): no workgroup barrier after the prologue,
// twelve independent load -> FFT -> store chains per CU instead of three barrier-coupled ones.
// Stereo clips arrive as [n][2] floats: samples 2n and 2n+1 of both channels are ONE float4 per lane, fetched once per
// frame (the block kernels and the first version of this one read 4-byte elements 16 bytes apart).
#ifndef REPET_FWD_SPLIT_BATCH
#define REPET_FWD_SPLIT_BATCH 4
#endif
constexpr int kFwdWaves = 12;
constexpr int kFwdLdsFloat2 = kFwdWaves * kExPitch + kTwFloat2 + 2 * kRegN;      // + window [N] + split twiddles [N]

struct __attribute__((packed, aligned(4))) Float4A { float x, y, z, w; };         // any dword address (segment offsets)
struct __attribute__((packed, aligned(4))) Float2A { float x, y; };

template <int CMODE>      // 2: stereo float4 path, 1: mono float2 path, 0: any channel count, element loads
__global__ __launch_bounds__(64 * kFwdWaves) void stft_reg_kernel(StftArgs a, int64_t units, int balance) {
    constexpr int N = kRegN, W = 2 * kRegN;
    extern __shared__ __attribute__((aligned(16))) float2 fwd_lds[];
    float2* ex_all = fwd_lds;
    float2* tw_lds = ex_all + kFwdWaves * kExPitch;
    float2* win_lds = tw_lds + kTwFloat2;               // (window[2n], window[2n+1])
    float2* split_lds = win_lds + N;                    // exp(-2 pi i k / W), k < N
    const int tid = threadIdx.x, lane_id = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: frame numbers, rows and pointers stay in SGPRs
    float2* ex = ex_all + wave * kExPitch;
    const int C = CMODE ? CMODE : a.n_channels;
    RSTAMP_DECL

    for (int i = tid; i < 16 * 64; i += 64 * kFwdWaves) {
        const int k1 = i >> 6, l = i & 63;
        tw_lds[i] = a.twiddle[2 * l * k1];                    // exp(-2 pi i l k1 / 1024)
    }
    if (tid < 64) tw_lds[16 * 64 + tid] = a.twiddle[32 * (tid >> 4) * (tid & 15)];    // exp(-2 pi i m2 j1 / 64)
    for (int i = tid; i < N; i += 64 * kFwdWaves) {
        win_lds[i] = *reinterpret_cast<const float2*>(a.window + 2 * i);
        const float2 w = a.twiddle[i];
        split_lds[i] = make_float2(0.5f * w.x, 0.5f * w.y);       // W^k / 2 (exact)
    }
    const RegTwiddles tw{tw_lds, tw_lds + 16 * 64};
    __syncthreads();
    RSTAMP(0)
    const bool want_mean = a.Vm || a.Vn || a.P || a.Ph;
    // round k of the launch: workgroup b's twelve waves take the twelve CONSECUTIVE frames (k gridDim + b) 12 + wave, so
    // the half frame two neighbours share is asked for by the same CU at about the same time
    const int64_t stride = (int64_t)gridDim.x * kFwdWaves;
    // the frames left after the whole rounds are dealt over ALL workgroups (`last` each, in the first waves) rather than
    // twelve each to the first few
    const int64_t whole = balance ? units / stride : (units + stride - 1) / stride;
    const int64_t rest = units - whole * stride;
    const int last = balance ? (int)((rest + gridDim.x - 1) / gridDim.x) : 0;

    for (int64_t k = 0; k <= whole; ++k) {
        int64_t u = k * stride + (int64_t)blockIdx.x * kFwdWaves + wave;
        if (k == whole) {
            if (wave >= last) break;
            u = whole * stride + (int64_t)blockIdx.x * last + wave;
        }
        if (u >= units) break;
        const int64_t b = a.n_batch > 1 ? (int64_t)((uint32_t)u / (uint32_t)a.T) : 0;     // launch_stft_reg: units < 2^31
        const int64_t t = u - b * a.T;
        const int64_t start = t * a.H - (a.centred ? W / 2 : 0);
        const int64_t row = t * a.FS;
        float2* Xb = a.X + b * a.batch_spec_stride;
        float* Vb = a.V + b * a.batch_spec_stride;
        float acc[17];
#pragma unroll
        for (int s = 0; s < 17; ++s) acc[s] = 0.f;
        // the lane number, made opaque once per frame: otherwise the compiler hoists the per-slot addresses out of the
        // loops and spills them
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        lane &= 63;                                    // (the range again: addresses as scalar base + 32-bit lane offset + immediate)
        const int partner = (64 - lane) & 63;
        const bool inside = start >= 0 && start + W <= a.n_samples;       // wave-uniform
        // a frame over an edge of the clip: positions [lo, hi) of it exist (32-bit, relative to the frame's first sample;
        // `frame` may point before the clip then and is only dereferenced at existing positions -- `safe` is one)
        const float* frame = a.audio + (a.sample_offset + b * a.batch_sample_stride + start) * C;
        const int lo = start < 0 ? (int)std::min<int64_t>(-start, W) : 0;
        const int hi = (int)std::max<int64_t>(std::min<int64_t>(a.n_samples - start, W), 0);
        const int safe = lo;
        const bool none = lo >= hi;                                        // nothing of the frame exists (an empty clip): all zeros, no load

        // Measured and dropped: fetching the next transform's samples (two dwords per point and channel, 32 registers)
        // while the current one is transformed -- cfg 2 / 3 / 5: 0.094 / 0.647 / 0.838 ms against 0.084 / 0.537 / 0.755
        // with the plain float4 fetch below; the strided dword loads cost the memory pipeline more than the wait they hide.
        float2 v[16];
        float2 v1[CMODE == 2 ? 16 : 1];                // stereo: channel 1, windowed, parked while channel 0 is transformed
        if constexpr (CMODE == 2) {
            float4 raw4[16];
            if (inside) {
                const Float4A* src = reinterpret_cast<const Float4A*>(frame) + lane;
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) { const Float4A q = src[64 * n1]; raw4[n1] = make_float4(q.x, q.y, q.z, q.w); }
            } else if (none) {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) raw4[n1] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const int p0 = 2 * (64 * n1 + lane), p1 = p0 + 1;
                    const bool in0 = p0 >= lo && p0 < hi, in1 = p1 >= lo && p1 < hi;
                    const Float2A q0 = *reinterpret_cast<const Float2A*>(frame + (in0 ? p0 : safe) * 2);
                    const Float2A q1 = *reinterpret_cast<const Float2A*>(frame + (in1 ? p1 : safe) * 2);
                    raw4[n1] = make_float4(in0 ? q0.x : 0.f, in0 ? q0.y : 0.f, in1 ? q1.x : 0.f, in1 ? q1.y : 0.f);
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float2 w = win_lds[64 * n1 + lane];
                v[n1] = make_float2(raw4[n1].x * w.x, raw4[n1].z * w.y);
                v1[n1] = make_float2(raw4[n1].y * w.x, raw4[n1].w * w.y);
            }
        }

#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            if constexpr (CMODE == 2) {
                if (c == 1) {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) v[n1] = v1[n1];
                }
            } else if constexpr (CMODE == 1) {
                if (inside) {
                    const Float2A* src = reinterpret_cast<const Float2A*>(frame) + lane;
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const Float2A q = src[64 * n1];
                        const float2 w = win_lds[64 * n1 + lane];
                        v[n1] = make_float2(q.x * w.x, q.y * w.y);
                    }
                } else if (none) {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) v[n1] = make_float2(0.f, 0.f);
                } else {
#pragma unroll
                    for (int n1 = 0; n1 < 16; ++n1) {
                        const int p0 = 2 * (64 * n1 + lane), p1 = p0 + 1;
                        const bool in0 = p0 >= lo && p0 < hi, in1 = p1 >= lo && p1 < hi;
                        const float q0 = frame[in0 ? p0 : safe], q1 = frame[in1 ? p1 : safe];
                        const float2 w = win_lds[64 * n1 + lane];
                        v[n1] = make_float2(in0 ? q0 * w.x : 0.f, in1 ? q1 * w.y : 0.f);
                    }
                }
            } else {
                int lane = lane_id;                    // opaque per channel: see above
                asm volatile("" : "+v"(lane));
        lane &= 63;                                    // (the range again: addresses as scalar base + 32-bit lane offset + immediate)
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const int p0 = 2 * (64 * n1 + lane), p1 = p0 + 1;
                    const bool in0 = !none && p0 >= lo && p0 < hi, in1 = !none && p1 >= lo && p1 < hi;
                    const float q0 = none ? 0.f : frame[(in0 ? p0 : safe) * C + c], q1 = none ? 0.f : frame[(in1 ? p1 : safe) * C + c];
                    const float2 w = win_lds[64 * n1 + lane];
                    v[n1] = make_float2(in0 ? q0 * w.x : 0.f, in1 ? q1 * w.y : 0.f);
                }
            }
            RSTAMP(1)                                  // samples fetched, windowed
            __builtin_amdgcn_sched_barrier(0);
            wave_fft1024<false>(v, ex, tw, lane);
            __builtin_amdgcn_sched_barrier(0);
            RSTAMP(2)                                  // FFT

            float2* Xrow = Xb + c * a.chan_stride + row;
            float* Vrow = Vb + c * a.chan_stride + row;
            // Hermitian split: X[k] = E + W_2048^k O with E = (Z[k] + conj Z[N-k]) / 2, O = (Z[k] - conj Z[N-k]) / 2i.
            // Z[N-k] for k = lane + 64 s is slot 15 - s of lane 64 - lane (lane 0: its own slot (16 - s) & 15).
            // Round 6: bins k and N - k come out of the SAME two values -- E[N-k] = conj E[k], O[N-k] = conj O[k], W^(N-k) = -conj W^k,
            // so X[N-k] = conj(E - W^k O) -- and a lane computes its slots s < 8 only, in pairs: X[k] for its own bin and X[N-k]
            // for the bin that lane 64 - lane holds in slot 15 - s, which it hands over (one cross-lane fetch per component,
            // as many as the split took before; twelve instructions per PAIR where every bin took twenty). Every lane still
            // stores its own sixteen bins, in ascending lane order: storing the mirrored bins where they are computed -- lanes on
            // descending addresses -- made the kernel 20 % SLOWER at cfg 3 / 5 (0.383 -> 0.459, 0.778 -> 0.899 ms; equal at cfg 2).
            // The table holds W^k / 2. Lane 0's pairs are its own slots s and 16 - s; its pair 0 is (0, N): the Nyquist bin;
            // bin N/2 (its slot 8) pairs with itself: X[512] = conj Z[512].
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto put = [&](int s_, float2 x) {                       // bin lane + 64 s_ of this lane
                const int k = lane + 64 * s_;
                const float mag = magnitude(x);
                // non-temporal: the spectrum is read again only after the similarity / period stages -- stored the ordinary way it
                // pushes the magnitudes and unit rows those stages read next out of the caches (1-2 % of every variant's step)
                { f2 y; y.x = x.x; y.y = x.y; __builtin_nontemporal_store(y, reinterpret_cast<f2*>(Xrow + k)); }
                Vrow[k] = mag;
                acc[s_] += mag;
            };
            // (pairs 4 .. 7 first, then 0 .. 3, each group's mirrored bins handed over before the next: four of them live instead
            // of eight -- with all eight the stereo kernel spilled two registers. Lane 0's slot 15 - j takes ITS pair j + 1: the
            // first group needs x512 for slot 8, the second keeps pair 4's mirrored bin for slot 12.)
            float2 xs[8];
            const float2 x512 = make_float2(v[8].x, -v[8].y);
#pragma unroll
            for (int g = 1; g >= 0; --g) {
#pragma unroll
            for (int s = 4 * g; s < 4 * g + 4; ++s) {
                const float2 zk = v[s];
                const float2 other = make_float2(lane_fetch(v[15 - s].x, partner), lane_fetch(v[15 - s].y, partner));
                const float2 mine = v[(16 - s) & 15];
                const float2 zn = (lane == 0) ? mine : other;
                const float sx = zk.x + zn.x, sy = zk.y - zn.y;           // Z[k] + conj Z[N-k]
                const float dx = zk.x - zn.x, dy = zk.y + zn.y;           // Z[k] - conj Z[N-k]
                const float2 wh = split_lds[lane + 64 * s];                // W^k / 2
                const float tx = fmaf(wh.x, dy, wh.y * dx);               // W^k O, O = (dy, -dx) / 2
                const float ty = fmaf(wh.y, dy, -(wh.x * dx));
                put(s, make_float2(fmaf(0.5f, sx, tx), fmaf(0.5f, sy, ty)));
                xs[s] = make_float2(fmaf(0.5f, sx, -tx), fmaf(-0.5f, sy, ty));
                if (s & 1) __builtin_amdgcn_sched_barrier(0);             // a few bins in flight, not sixteen
            }
#pragma unroll
            for (int j = 4 * g; j < 4 * g + 4; ++j) {                     // slots 15 - j: what the mirror lane computed for them
                const float2 got = make_float2(lane_fetch(xs[j].x, partner), lane_fetch(xs[j].y, partner));
                const float2 own0 = j < 7 ? xs[j + 1] : x512;
                put(15 - j, (lane == 0) ? own0 : got);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
            }
            {   // k = N (Nyquist): lane 0's pair 0
                const float2 x = make_float2(xs[0].x, 0.f);
                const float mag = magnitude(x);           // (not fabsf: the inverse recomputes every bin's magnitude the same way)
                if (lane == 0) { Xrow[N] = x; Vrow[N] = mag; }
                acc[16] += mag;
            }
            if (lane < a.FS - (N + 1)) {      // zero the pad bins [F, FS)
                Xrow[N + 1 + lane] = make_float2(0.f, 0.f);
                Vrow[N + 1 + lane] = 0.f;
            }
            RSTAMP(3)                                  // split + X / V stores issued
        }
        if (!want_mean) continue;

        // channel mean (repet.py:162,:667) and its L2 norm over frequency (repet.py:1220), all inside the wave
        const float inv_c = 1.0f / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int s = 0; s < 17; ++s) {
            acc[s] = (C == 1) ? acc[s] : acc[s] * inv_c;
            if (s < 16 || lane == 0) ss += acc[s] * acc[s];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        // A frame whose squares leave fp32's range at the bottom (every mean magnitude below 2^-40: ss underflows, or its
        // reciprocal root overflows) is normalised from the row scaled by 2^90 -- wave-uniform and almost never taken; the
        // float64 reference gives such a frame finite unit values, and so does magnitude() (common.h) for its bins.
        float pre = 1.0f;
        if (ss < 0x1p-80f) {
            pre = 0x1p90f;
            ss = 0.f;
#pragma unroll
            for (int s = 0; s < 17; ++s) if (s < 16 || lane == 0) ss += (acc[s] * pre) * (acc[s] * pre);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
        }
        const float norm = sqrtf(ss);        // 0 for a silent frame: 0/0 = NaN like repet.py:1220
        // (the row times the correctly rounded reciprocal: seventeen IEEE divisions were 180 instructions per frame in a
        // kernel bound by instruction issue; a silent frame gives 0 * inf = NaN, as 0 / 0 did)
        const float inv_norm = 1.0f / norm;
        const int64_t mrow = b * a.batch_mean_stride + row;
        float* Vm = a.Vm ? a.Vm + mrow : nullptr;
        float* Vn = a.Vn ? a.Vn + mrow : nullptr;
        float* P = a.P ? a.P + mrow : nullptr;
        void* Vh = a.Vh ? static_cast<void*>(static_cast<_Float16*>(a.Vh) + 2 * b * a.batch_mean_stride) : nullptr;
        // component k = lane + 64 s of a row that starts on a 32-component boundary sits at 2 row + 64 (k >> 5) + (k & 31)
        // of the interleaved f16 planes: a per-lane base and 256 bytes per s (as immediates), instead of 64-bit shifts and
        // masks per store (300 integer instructions per frame)
        const int64_t plane_lane = 2 * row + ((lane >> 5) << 6) + (lane & 31);
        _Float16* vh_lane = Vh ? static_cast<_Float16*>(Vh) + plane_lane : nullptr;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = lane + 64 * s;
            if (Vm) Vm[k] = acc[s];
            const float unit = (acc[s] * pre) * inv_norm;
            if (Vn) Vn[k] = unit;
            if (Vh) store_split_f16_at(vh_lane + 128 * s, unit);
            if (P) P[k] = acc[s] * acc[s];
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (lane == 0) {
            if (Vm) Vm[N] = acc[16];
            if (Vn) Vn[N] = (acc[16] * pre) * inv_norm;
            if (Vh) store_split_f16(Vh, row + N, (acc[16] * pre) * inv_norm);
            if (P) P[N] = acc[16] * acc[16];
        }
        if (lane < a.FS - (N + 1)) {
            if (Vm) Vm[N + 1 + lane] = 0.f;
            if (Vn) Vn[N + 1 + lane] = 0.f;
            if (Vh) store_split_f16(Vh, row + N + 1 + lane, 0.f);
            if (P) P[N + 1 + lane] = 0.f;
        }
        if (a.Ph) {
            // the squared row straight into the row-scaled f16 planes (split_f16_rows_kernel's arithmetic on the same values)
            float mx = 0.f;
#pragma unroll
            for (int s = 0; s < 17; ++s) if (s < 16 || lane == 0) mx = fmaxf(mx, acc[s] * acc[s]);      // fmaxf drops NaN
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float sc = f16_row_scale(mx);
            void* Ph = static_cast<void*>(static_cast<_Float16*>(a.Ph) + 2 * b * a.batch_mean_stride);
            if (lane == 0) a.Ph_inv[b * a.batch_inv_stride + t] = 1.0f / sc;
            _Float16* ph_lane = static_cast<_Float16*>(Ph) + plane_lane;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                store_split_f16_scaled_at(ph_lane + 128 * s, acc[s] * acc[s], sc);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (lane == 0) store_split_f16_scaled(Ph, row + N, acc[16] * acc[16], sc);
            if (lane < a.FS - (N + 1)) store_split_f16_scaled(Ph, row + N + 1 + lane, 0.f, sc);
        }
        RSTAMP(4)                                      // mean rows
    }
}

// ---- inverse + overlap-add ---------------------------------------------------------------------------
// Twelve wavefronts per workgroup, one workgroup per CU, a wave inverts a whole frame (every channel) per round and the
// twelve frames of a round are CONSECUTIVE: hop t = first half of frame t + second half of frame t - 1, so a wave
// leaves its frame's second half in its (now idle) exchange region, one barrier, and reads its left neighbour's; the
// last wave's goes through a carry region to the first wave of the next round. The workgroup's first wave inverts the
// frame before the run once more (its tail is all that is used). Two barriers per round of 12 x C transforms, none
// inside a transform; output as one 16-byte store per lane and sample pair of a stereo clip.
constexpr int kInvWaves = 12;
constexpr int kInvLdsFloat2 = kInvWaves * kExPitch + kTwFloat2 + kRegN + 2 * 8 * 64;    // + split twiddles [N] + carry [2][8][64]

template <int C, int MASKED>      // C = 1, 2; MASKED: 0 no mask, 1 a mask plane M, 2 magnitudes + repeating-segment model
__global__ __launch_bounds__(64 * kInvWaves) void istft_ola_reg_kernel(IstftOlaArgs a, int rounds) {
    constexpr int N = kRegN;          // samples per hop = complex FFT length
    extern __shared__ __attribute__((aligned(16))) float2 inv_lds[];
    float2* ex_all = inv_lds;
    float2* tw_lds = ex_all + kInvWaves * kExPitch;
    float2* split_lds = tw_lds + kTwFloat2;                     // conj(exp(-2 pi i k / W)), k < N
    float2* carry = split_lds + N;                              // [C][8][64]: second half of the frame before the round
    const int tid = threadIdx.x, lane_id = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* ex = ex_all + wave * kExPitch;
    if (a.n_batch > 0) {
        const int j = a.batch_first + (int)blockIdx.y * a.batch_step;
        a.Y += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        if (MASKED == 1) a.M += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        if (MASKED == 2) {
            const int local = a.batch_local0 + (int)blockIdx.y * a.batch_step;
            a.model += (int64_t)local * a.model_batch_stride;
            a.periods += local;
        }
        a.out_offset += (int64_t)j * a.batch_out_stride;
        a.fade_in = j > 0 ? a.overlap : 0;
        a.fade_out = a.overlap;
        a.seg_step = a.batch_out_stride;
        a.later = a.batch_total - 1 - j;
    }
    for (int i = tid; i < 16 * 64; i += 64 * kInvWaves) {
        const int k1 = i >> 6, l = i & 63;
        tw_lds[i] = cconj(a.twiddle[2 * l * k1]);                 // exp(+2 pi i l k1 / 1024)
    }
    if (tid < 64) tw_lds[16 * 64 + tid] = cconj(a.twiddle[32 * (tid >> 4) * (tid & 15)]);
    for (int i = tid; i < N; i += 64 * kInvWaves) { const float2 w = a.twiddle[i]; split_lds[i] = make_float2(0.5f * w.x, -0.5f * w.y); }   // conj(W^k) / 2 (exact)
    for (int i = tid; i < C * 8 * 64; i += 64 * kInvWaves) carry[i] = make_float2(0.f, 0.f);
    const RegTwiddles tw{tw_lds, tw_lds + 16 * 64};
    __syncthreads();

    const int period = MASKED == 2 ? __builtin_amdgcn_readfirstlane(a.periods[0]) : 1;
    const int hops_per_wg = kInvWaves * rounds - 1;
    const int64_t h0 = a.first_hop + (int64_t)blockIdx.x * hops_per_wg;
    const int64_t h_last = (h0 + hops_per_wg - 1 < a.last_hop) ? h0 + hops_per_wg - 1 : a.last_hop;
    const float inv_n = 1.0f / (float)N;
    const int mode = a.accumulate_weighted;
    // (positions inside a segment fit 32 bits here: launch_istft_ola_reg leaves longer segments to the block kernel)
    const int fade_in = (int)a.fade_in, overlap = (int)a.fade_out, step = (int)a.seg_step;
    const float den_in = 1.0f / (float)(2 * a.fade_in), den_ov = 1.0f / (float)(2 * a.fade_out);       // (reciprocals: segment_weight32)
    auto weight = [&](int64_t n) -> float { return segment_weight32((int)n, fade_in, overlap, step, a.later, den_in, den_ov); };
    // With a step of at least half a segment only the NEXT segment's fall can reach a sample (the default: 10-s segments
    // every 5 s): the same two factors without segment_weight32's loop over the later segments, whose trip count differs
    // from lane to lane (sixteen divergent loops per hop and lane). Wave-uniform choice per hop below.
    auto weight_one_later = [&](int n) -> float {
        float w = 1.f;
        if (n < fade_in) w = (float)(2 * n + 1) * den_in;
        const int rr = n - step;
        if (rr >= 0 && rr < overlap) w *= (float)(2 * (overlap - rr) - 1) * den_ov;
        return w;
    };
    const bool later_any = a.later >= 1 && overlap > 0 && step > 0;

    for (int r = 0; r < rounds; ++r) {
        const int64_t t = h0 - 1 + (int64_t)kInvWaves * r + wave;       // this wave's frame; it emits hop t
        const bool have = t >= 0 && t < a.T && t <= h_last;              // wave-uniform
        float2 head[C][8], tail[C][8];
#pragma unroll 1
        for (int c = 0; c < C; ++c) {
            int lane = lane_id;                                          // opaque per transform (see the forward kernel)
            asm volatile("" : "+v"(lane));
            float2 v[16];
            if (have) {
                const float2* Y = a.Y + c * a.chan_stride + t * a.FS;
                const float* Mr = MASKED == 1 ? a.M + c * a.chan_stride + t * a.FS : nullptr;
                // MASKED == 2: the model row of this frame's position inside its period (wave-uniform)
                const float* Wr = MASKED == 2 ? a.model + c * a.model_chan_stride + (t % period) * a.FS : nullptr;
                // merge the half spectrum back into the packed transform: Z[k] = E + i conj(W_2048^k) D, E = (A + B) / 2,
                // D = (A - B) / 2, A = X[k], B = conj X[N-k] (both masked first). Round 6: in PAIRS, as the forward kernel splits --
                // Z[N-k] = conj(E) + i conj(conj(W^k) D) comes out of the same E and D, so a lane takes its slots s < 8 only:
                // it fetches X[k] and X[N-k] once (every bin was fetched twice before, by its own lane and by its mirror's), keeps
                // Z[k] and hands Z[N-k] to lane 64 - lane, whose slot 15 - s it is (one cross-lane fetch per component). The masks
                // of exactly these sixteen bins are what the lane needs: MASKED == 2 computes them where they are used -- the
                // trip of all masks through the wave's exchange region (and the two waits around it) is gone. The table holds
                // conj(W^k) / 2. Lane 0: its pairs are its own slots s and 16 - s; (0, N) gives Z[0] alone; bin N/2 pairs with
                // itself, Z[N/2] = conj(A). All loads of a transform are issued before the first is consumed.
                const int partner = (64 - lane) & 63;
                float2 xk[8], xc[8], x512;
                float mk[MASKED == 1 ? 8 : 1], mc[MASKED == 1 ? 8 : 1], m512 = 1.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) { xk[j] = Y[64 * j + lane]; xc[j] = Y[N - (64 * j + lane)]; }
                x512 = Y[N / 2];                                              // (every lane the same address: one request)
                if constexpr (MASKED == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { mk[j] = Mr[64 * j + lane]; mc[j] = Mr[N - (64 * j + lane)]; }
                    m512 = Mr[N / 2];
                }
                float wk[MASKED == 2 ? 8 : 1], wc[MASKED == 2 ? 8 : 1];          // MASKED == 2: the model at the lane's sixteen bins
                if constexpr (MASKED == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { wk[j] = Wr[64 * j + lane]; wc[j] = Wr[N - (64 * j + lane)]; }
                    m512 = soft_mask(magnitude(x512), Wr[N / 2], N / 2, a.cutoff);
                }
                float2 send[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int k = 64 * j + lane;
                    float2 a_ = xk[j], b_ = xc[j];
                    if constexpr (MASKED == 2) {
                        // the pair's masks where they are used: |X| with the forward kernel's own roundings (magnitude(): V is not
                        // read at all); the model values die here (all sixteen masks first cost a register spill)
                        const float mk2 = soft_mask(magnitude(a_), wk[j], k, a.cutoff), mc2 = soft_mask(magnitude(b_), wc[j], N - k, a.cutoff);
                        a_ = make_float2(mul_rounded(a_.x, mk2), mul_rounded(a_.y, mk2));
                        b_ = make_float2(mul_rounded(b_.x, mc2), mul_rounded(b_.y, mc2));
                        __builtin_amdgcn_sched_barrier(0);                     // two divisions' temporaries at a time, not sixteen
                    } else if constexpr (MASKED == 1) {
                        a_ = make_float2(mul_rounded(a_.x, mk[j]), mul_rounded(a_.y, mk[j]));
                        b_ = make_float2(mul_rounded(b_.x, mc[j]), mul_rounded(b_.y, mc[j]));
                    }
                    const float ex_ = a_.x + b_.x, ey_ = a_.y - b_.y;         // A + B   (B = conj X[N-k])
                    const float dx_ = a_.x - b_.x, dy_ = a_.y + b_.y;         // A - B
                    const float2 w = split_lds[k];                             // conj(W^k) / 2
                    const float ox = fmaf(dx_, w.x, -(dy_ * w.y));            // o = conj(W^k) D
                    const float oy = fmaf(dx_, w.y, dy_ * w.x);
                    v[j] = make_float2(fmaf(0.5f, ex_, -oy), fmaf(0.5f, ey_, ox));         // E + i o
                    send[j] = make_float2(fmaf(0.5f, ex_, oy), fmaf(-0.5f, ey_, ox));      // conj(E) + i conj(o)
                }
                float2 z512 = x512;
                if constexpr (MASKED != 0) z512 = make_float2(mul_rounded(x512.x, m512), mul_rounded(x512.y, m512));
                z512.y = -z512.y;                                              // Z[N/2] = conj(A)
                // slots 15 .. 8: from the mirror lane's pairs 0 .. 7 (lane 0: its own pairs 1 .. 7 are its slots 15 .. 9, slot 8 is bin N/2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 got = make_float2(lane_fetch(send[j].x, partner), lane_fetch(send[j].y, partner));
                    const float2 own0 = j < 7 ? send[j + 1] : z512;
                    v[15 - j] = (lane == 0) ? own0 : got;
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_sched_barrier(0);
                wave_fft1024<true>(v, ex, tw, lane);
                __builtin_amdgcn_sched_barrier(0);
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) v[s] = make_float2(0.f, 0.f);
            }
            // v[s] = samples (2n, 2n + 1), n = lane + 64 s, of the frame (unscaled)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (c == 0) { head[0][s] = v[s]; tail[0][s] = v[8 + s]; }
                else { head[C - 1][s] = v[s]; tail[C - 1][s] = v[8 + s]; }
            }
        }
        // this frame's second halves -> my exchange region (idle between transforms), [c][s][lane]
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int s = 0; s < 8; ++s) ex[(c * 8 + s) * 64 + lane_id] = tail[c][s];
        __syncthreads();
        // hop t = my heads + the second halves of frame t - 1 (my left neighbour's; wave 0: the carry of the last round)
        const float2* prev = wave > 0 ? ex - kExPitch : carry;
        const bool emit = t >= h0 && t <= h_last;                       // wave-uniform
        if (emit) {
            const int64_t n_base = t * N - a.trim;                       // output sample of the hop's first sample
            // (no later segment at all: the fall factor must not apply; two steps beyond the hop's last sample: only q = 1 can)
            const bool one_later = later_any && n_base >= 0 && n_base + 2 * (int64_t)N < 2 * (int64_t)step;
            int lane = lane_id;                                          // opaque: no per-slot addresses kept across rounds
            asm volatile("" : "+v"(lane));
            // an accumulating class of `extended` segments: the eight old values of this lane are fetched together before
            // the first is used (each behind its own "mode == 1" branch they were eight memory round trips per hop)
            using OldT = std::conditional_t<C == 2, Float4A, Float2A>;
            const bool whole = n_base >= 0 && n_base + N <= a.n_out;    // wave-uniform: the hop lies inside the output
#pragma unroll
            for (int s0 = 0; s0 < 8; s0 += 4) {                          // (four at a time: registers)
            OldT old4[4];
            if (mode == 1 && whole) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    old4[s] = *reinterpret_cast<const OldT*>(a.out + (a.out_offset + n_base + 2 * (lane + 64 * (s0 + s))) * C);
            }
#pragma unroll
            for (int s = s0; s < s0 + 4; ++s) {
                float o[C][2];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float2 p = prev[(c * 8 + s) * 64 + lane];
                    o[c][0] = (head[c][s].x + p.x) * inv_n * a.scale;
                    o[c][1] = (head[c][s].y + p.y) * inv_n * a.scale;
                }
                const int64_t n0 = n_base + 2 * (lane + 64 * s);          // samples n0, n0 + 1
                float* dst = a.out + (a.out_offset + n0) * C;
                if (whole || (n0 >= 0 && n0 + 1 < a.n_out)) {
                    float w0 = 1.f, w1 = 1.f;
                    if (mode != 0) {
                        if (one_later) { w0 = weight_one_later((int)n0); w1 = weight_one_later((int)n0 + 1); }
                        else { w0 = weight(n0); w1 = weight(n0 + 1); }
                    }
                    if constexpr (C == 2) {
                        Float4A* q = reinterpret_cast<Float4A*>(dst);
                        Float4A r{o[0][0], o[1][0], o[0][1], o[1][1]};
                        if (mode == 1) {
                            const Float4A was = whole ? old4[s - s0] : *q;
                            r = Float4A{was.x + w0 * r.x, was.y + w0 * r.y, was.z + w1 * r.z, was.w + w1 * r.w};
                        } else if (mode == 2) r = Float4A{w0 * r.x, w0 * r.y, w1 * r.z, w1 * r.w};
                        *q = r;
                    } else {
                        Float2A* q = reinterpret_cast<Float2A*>(dst);
                        Float2A r{o[0][0], o[0][1]};
                        if (mode == 1) {
                            const Float2A was = whole ? old4[s - s0] : *q;
                            r = Float2A{was.x + w0 * r.x, was.y + w1 * r.y};
                        } else if (mode == 2) r = Float2A{w0 * r.x, w1 * r.y};
                        *q = r;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {                       // the first and the last hop of a clip or segment
                        const int64_t n = n0 + e;
                        if (n < 0 || n >= a.n_out) continue;
                        const float w = mode != 0 ? weight(n) : 1.f;
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            float* q = dst + e * C + c;
                            if (mode == 1) *q += w * o[c][e];
                            else *q = w * o[c][e];
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        if (wave == kInvWaves - 1) {                                    // the next round's first frame follows mine
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int s = 0; s < 8; ++s) carry[(c * 8 + s) * 64 + lane_id] = tail[c][s];
        }
    }
}

}  // namespace

bool reg_fft_supported(int W, int n_channels, bool inverse) {
    // Both wave kernels are the default for mono and stereo clips at W = 2048 (cfg 2 / 3 / 4 / 5, block kernels -> these:
    // forward 0.095 / 0.62 / 0.099 / 0.795 ms -> 0.084 / 0.54 / 0.074 / 0.755; inverse 0.071 / 0.603 / 0.059 / 0.604 ->
    // 0.058 / 0.481 / 0.054 / 0.537 -- the inverse only once ALL loads of a transform were issued before the first is
    // consumed: in batches of four it was no faster than the block kernel).
    // REPET_FFT_PATH=block | wave: the LDS Stockham kernels of stft.hip for everything; =reg: both kernels of this file for
    // any channel count they take; =fwd: the forward one only.
    static const int mode = [] {
        const char* e = getenv("REPET_FFT_PATH");
        if (!e || !e[0]) return 4;
        return e[0] == 'r' ? 3 : e[0] == 'f' ? 1 : 0;
    }();
    if (W != 2 * kRegN) return false;
    if (inverse) return (mode & 6) && (n_channels == 1 || n_channels == 2);
    if (mode & 4) return n_channels == 1 || n_channels == 2;
    return (mode & 1) && n_channels >= 1;
}

hipError_t launch_stft_reg(const StftArgs& a, hipStream_t s) {
    const int64_t batches = a.n_batch > 0 ? a.n_batch : 1;
    const int64_t units = a.T * batches;
    if (units >= (int64_t)1 << 31) return hipErrorInvalidValue;
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop{};
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    const unsigned blocks = (unsigned)std::min<int64_t>(ceil_div(units, kFwdWaves), cus);
    const int balance = 1;
    const size_t dyn = (size_t)kFwdLdsFloat2 * sizeof(float2);
    auto go = [&](auto kernel) {
        (void)ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)dyn);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64 * kFwdWaves), dyn, s, a, units, balance);
    };
    if (a.n_channels == 2) go(&stft_reg_kernel<2>);
    else if (a.n_channels == 1) go(&stft_reg_kernel<1>);
    else go(&stft_reg_kernel<0>);
    return hipGetLastError();
}

hipError_t launch_istft_ola_reg(const IstftOlaArgs& a, int64_t hops, hipStream_t s) {
    const int64_t batches = a.n_batch > 0 ? a.n_batch : 1;
    const int64_t lim = (int64_t)1 << 30;
    if (a.accumulate_weighted && (a.n_out >= lim || a.batch_out_stride >= lim || a.overlap >= lim || a.fade_in >= lim || a.fade_out >= lim))
        return a.model ? hipErrorInvalidValue : hipErrorNotSupported;      // cross-fade positions beyond 32 bits: the block kernel (which takes no model: istft_reg_takes)
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop{};
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    // a workgroup of R rounds emits 12 R - 1 hops (its first wave repeats the frame before the run): R is chosen for the
    // fewest rounds of workgroups x rounds inside them
    int rounds = 1;
    double best = 0;
    for (int R = 1; R <= 16; ++R) {
        const int64_t wgs = ceil_div(hops, kInvWaves * R - 1) * batches;
        const double cost = (double)ceil_div(wgs, cus) * (R + 0.3);
        if (R == 1 || cost < best * 0.98) { best = cost; rounds = R; }
    }
    const unsigned blocks = (unsigned)ceil_div(hops, kInvWaves * rounds - 1);
    const size_t dyn = (size_t)kInvLdsFloat2 * sizeof(float2);
    auto go = [&](auto kernel) {
        (void)ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)dyn);
        hipLaunchKernelGGL(kernel, dim3(blocks, (unsigned)batches), dim3(64 * kInvWaves), dyn, s, a, rounds);
    };
    const int mode = a.model ? 2 : a.M ? 1 : 0;
    if (a.n_channels == 2) { if (mode == 2) go(&istft_ola_reg_kernel<2, 2>); else if (mode == 1) go(&istft_ola_reg_kernel<2, 1>); else go(&istft_ola_reg_kernel<2, 0>); }
    else { if (mode == 2) go(&istft_ola_reg_kernel<1, 2>); else if (mode == 1) go(&istft_ola_reg_kernel<1, 1>); else go(&istft_ola_reg_kernel<1, 0>); }
    return hipGetLastError();
}

bool istft_reg_takes(const IstftOlaArgs& a) {
    const int64_t lim = (int64_t)1 << 30;
    if (a.out_channels != 0 || !reg_fft_supported(a.W, a.n_channels, true)) return false;
    return !(a.accumulate_weighted && (a.n_out >= lim || a.batch_out_stride >= lim || a.overlap >= lim || a.fade_in >= lim || a.fade_out >= lim));
}

}  // namespace repet

#ifdef REPET_FFT_STAMPS
extern "C" int repet_debug_reg_stamps(unsigned long long* out, int clear) {
    if (clear) {
        static unsigned long long zeros[64];
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(repet::g_reg_stamps), zeros, sizeof(zeros));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(repet::g_reg_stamps), sizeof(unsigned long long) * 64);
}
#endif
