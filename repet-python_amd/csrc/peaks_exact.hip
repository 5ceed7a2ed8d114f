// K4 / K4b, second level: rows of the peak picking whose decisions the fp32 SPECTRA cannot settle are decided again from
// float64 spectra (repet.py:149, :1220-1223, :1318-1326: the reference decides in float64 on complex128 spectra).
//
// The first pass (peaks.hip / peaks_wave.hip) takes every decision within `delta` of a tie out of the fp32 similarity
// matrix and settles it with float64 dot products of the fp32 unit rows ("level 1"). Those values still carry the
// rounding of the fp32 FFT, magnitudes and unit rows: against the float64 reference they are off by up to 1e-7 (rms
// 1.2e-8, tools/level_error_probe.py), and repeating music puts a few decisions per hundred rows inside that -- and an
// exactly periodic clip puts ALL of them there (every copy of a frame ties with every other: "flat rows", more near-ties
// than the first pass's lists hold). The first pass therefore hands over every row with a level-1 comparison closer than
// `delta2`, and every flat row, and this kernel decides those rows again, without caps:
//   * the row is scanned again: candidates (not a safe "no" against the fp32 window maximum and threshold), their rivals
//     (window elements within delta below them), level-1 values for all of them -- the first pass's arithmetic;
//   * every element in a level-1 comparison closer than delta2 (against a rival, the threshold, or across the
//     top-`number` cut) gets a LEVEL-2 value: the float64 dot product of float64 unit rows, which are computed on demand
//     from the waveform -- float64 Hamming window, float64 real FFT (one workgroup, W/2-point complex FFT in LDS + split
//     pass), float64 magnitudes, channel mean, norm -- and kept in a table for the other rows of the launch. When the
//     caller's array was float64 the waveform is the fp32 sample PLUS its fp32 remainder (hostio.hip uploads the
//     remainder when it is not zero everywhere), i.e. 48 bits of the caller's 53;
//   * the decisions, the ranking and the top-`number` cut are then the first pass's, on those values.
// A workgroup takes rows from the list until it is empty (fixed grid: the host never learns the count).
#include "peaks.h"
#include "fft_wave_f64.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace repet {

namespace {

// make stamps-style diagnostics: -DREPET_EXACT_STAMPS accumulates the 100 MHz ticks of every phase of a row into stats[10..15]
#ifdef REPET_EXACT_STAMPS
#define XSTAMP_DECL unsigned long long xlast_ = wall_clock64();
#define XSTAMP(k) { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&stats[24 + (k)], (unsigned)(now_ - xlast_)); xlast_ = now_; } }
#else
#define XSTAMP_DECL
#define XSTAMP(k)
#endif

#ifdef REPET_EXACT_STAMPS
__device__ unsigned long long g_unit_stamps[8];
#define USTAMP(k) { const unsigned long long now_ = wall_clock64(); if (lane == 0) atomicAdd(&g_unit_stamps[k], now_ - ulast_); ulast_ = now_; }
#define USTAMP_DECL unsigned long long ulast_ = wall_clock64();
#else
#define USTAMP(k)
#define USTAMP_DECL
#endif
constexpr int kExactThreads = 256, kExactWaves = kExactThreads / 64;
constexpr int kScanChunk = 4096;      // row elements tested per pass through the LDS window-maximum buffers
constexpr int kRankLds = 512;         // candidate lists up to this length are ranked out of LDS
constexpr unsigned int kNear = 1, kSure = 2, kNeed1 = 4, kNeed2 = 8, kBand = 128;

struct ExactArgs {
    PeakArgs a;
    double delta2;
    const int* redo_list;           // (row, clip) pairs, a.stats[4] of them
    ExactSource src;
    unsigned gen;
    int logM;                       // log2(W / 2)
    unsigned char* scratch; size_t scratch_per_wg; int n4;
    // LDS: [stage twiddles, M/2 complex doubles (tw_in_lds)] [work: fft_waves transforms side by side | the two scan buffers]
    int tw_count, fft_waves, wave_bytes, scan_in_lds;
    int reg_fft;                    // W = 2048: the transform in the wave's registers (fft_wave_f64.h); LDS: its twiddles, then the work region
};

struct RowScratch {
    float* v; double* val; unsigned long long* best; unsigned int* st; int* list; int* list2; int* near; float* pval; int* pidx; int* prank;
};

__host__ __device__ inline size_t exact_row_scratch_bytes(int n4) {
    return (size_t)n4 * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 4 + 4 + 4) + 256;
}

__device__ __forceinline__ RowScratch carve_row(unsigned char* p, int n4) {
    RowScratch r;
    r.val = reinterpret_cast<double*>(p); p += (size_t)n4 * 8;
    r.best = reinterpret_cast<unsigned long long*>(p); p += (size_t)n4 * 8;
    r.v = reinterpret_cast<float*>(p); p += (size_t)n4 * 4;
    r.list = reinterpret_cast<int*>(p); p += (size_t)n4 * 4;
    r.near = reinterpret_cast<int*>(p); p += (size_t)n4 * 4;
    r.pval = reinterpret_cast<float*>(p); p += (size_t)n4 * 4;
    r.pidx = reinterpret_cast<int*>(p); p += (size_t)n4 * 4;
    r.prank = reinterpret_cast<int*>(p); p += (size_t)n4 * 4;
    r.list2 = reinterpret_cast<int*>(p); p += (size_t)n4 * 4;
    r.st = reinterpret_cast<unsigned int*>(p);
    return r;
}

// LDS traffic of one wave is ordered by the hardware; this only stops the compiler from moving accesses across
__device__ __forceinline__ void exact_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// float64 unit row of frame row `fr` of clip `clip` into the table, by ONE WAVEFRONT and without a workgroup barrier (the
// waves of a workgroup transform different frames side by side): Hamming window, real FFT of W samples as a W/2-point
// complex FFT in the wave's LDS region Zw + a split pass, magnitudes summed over the channels in registers (ACC_REGS:
// W <= 2048, the lane owns bins lane + 64 j) or in accw, channel mean (repet.py:667), norm (repet.py:1220).
template <bool ACC_REGS, bool STOCKHAM>
__device__ __forceinline__ void wave_unit_row_f64(const ExactSource& s, int logM, int clip, int64_t fr, double2* Zw, double* accw, const double2* tws, int lane) {
    const int W = s.W, M = W >> 1, F = s.F, C = s.n_channels;
    const int64_t s0 = s.frame_sample0 + fr * (int64_t)s.H;
    const float* hi = s.hi + (int64_t)clip * s.clip_stride;
    const float* lo = s.lo ? s.lo + (int64_t)clip * s.clip_stride : nullptr;
    const float* lo_or_hi = lo ? lo : hi;                      // (see wave_unit_row_f64_reg)
    const double lo_scale = lo ? 1.0 : 0.0;
    int rel_lo = (int)min(max(-s0, (int64_t)0), (int64_t)W);
    int rel_hi = (int)min(max(s.n_samples - s0, (int64_t)0), (int64_t)W);
    const bool none = rel_hi <= rel_lo;
    const int64_t anchor = none ? 0 : s0;
    if (none) rel_hi = rel_lo + 1;
    const float* frame_hi = hi + anchor * C;
    const float* frame_lo = lo_or_hi + anchor * C;
    constexpr int kAcc = ACC_REGS ? 17 : 1;
    double acc[kAcc];
#pragma unroll
    for (int jj = 0; jj < kAcc; ++jj) acc[jj] = 0.0;
    if (!ACC_REGS) for (int k = lane; k <= M; k += 64) accw[k] = 0.0;
    for (int c = 0; c < C; ++c) {
        exact_wave_sync();
        // z[r] = x[2r] + i x[2r+1] (windowed) at the bit-reversed position. Coalesced loads (the scatter is the LDS's), eight
        // points per lane in flight, no branch around a load: behind their bounds checks the loads went out one by one,
        // each after the previous round trip
        for (int r0 = 0; r0 < M; r0 += 512) {
            float h0[8], h1[8], l0[8], l1[8];
            double w0[8], w1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = min(r0 + 64 * u + lane, M - 1);
                const int q0 = min(max(2 * r, rel_lo), rel_hi - 1), q1 = min(max(2 * r + 1, rel_lo), rel_hi - 1);     // (32-bit, frame-relative)
                h0[u] = frame_hi[q0 * C + c]; h1[u] = frame_hi[q1 * C + c];
                l0[u] = frame_lo[q0 * C + c]; l1[u] = frame_lo[q1 * C + c];
                w0[u] = s.window64[2 * r]; w1[u] = s.window64[2 * r + 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 64 * u + lane;
                const int p0 = 2 * r;
                const bool in0 = !none && p0 >= rel_lo && p0 < rel_hi, in1 = !none && p0 + 1 >= rel_lo && p0 + 1 < rel_hi;
                const double re = in0 ? ((double)h0[u] + lo_scale * (double)l0[u]) * w0[u] : 0.0;
                const double im = in1 ? ((double)h1[u] + lo_scale * (double)l1[u]) * w1[u] : 0.0;
                if (r < M) Zw[STOCKHAM ? r : (int)(__brev((unsigned)r) >> (32 - logM))] = make_double2(re, im);
            }
        }
        exact_wave_sync();
        if constexpr (STOCKHAM) {
            // M <= 1024: in-place Stockham passes of radix 4 (a closing radix 2 when log2 M is odd), the wave-synchronous
            // scheme of stft.hip's wave_fft in float64: every lane reads the operands of its (up to four) butterflies into
            // registers, then writes the results to their sorted places -- five passes instead of ten, natural order in
            // and out, and few registers (the kernel shares a CU with other kernels' waves). tws[m] = exp(-2 pi i m / M), m < M.
            for (int p = 1; p < M;) {
                if (M / p >= 4) {
                    const int tstep = M / (p * 4);
                    double2 u[4][4];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int i = min(lane + 64 * b, (M >> 2) - 1);
                        u[b][0] = Zw[i]; u[b][1] = Zw[i + (M >> 2)]; u[b][2] = Zw[i + (M >> 1)]; u[b][3] = Zw[i + 3 * (M >> 2)];
                    }
                    exact_wave_sync();
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int i = lane + 64 * b;
                        if (i < (M >> 2)) {
                            const int k = i & (p - 1);
                            const int j = ((i - k) << 2) + k;
                            double2 u0 = u[b][0], u1 = u[b][1], u2 = u[b][2], u3 = u[b][3];
                            if (p > 1) {
                                const double2 w1 = tws[k * tstep], w2 = tws[2 * k * tstep], w3 = tws[3 * k * tstep];
                                u1 = make_double2(u1.x * w1.x - u1.y * w1.y, u1.x * w1.y + u1.y * w1.x);
                                u2 = make_double2(u2.x * w2.x - u2.y * w2.y, u2.x * w2.y + u2.y * w2.x);
                                u3 = make_double2(u3.x * w3.x - u3.y * w3.y, u3.x * w3.y + u3.y * w3.x);
                            }
                            const double2 t0 = make_double2(u0.x + u2.x, u0.y + u2.y), t1 = make_double2(u0.x - u2.x, u0.y - u2.y);
                            const double2 t2 = make_double2(u1.x + u3.x, u1.y + u3.y), d = make_double2(u1.x - u3.x, u1.y - u3.y);
                            const double2 t3 = make_double2(d.y, -d.x);                       // -i d
                            Zw[j] = make_double2(t0.x + t2.x, t0.y + t2.y);
                            Zw[j + p] = make_double2(t1.x + t3.x, t1.y + t3.y);
                            Zw[j + 2 * p] = make_double2(t0.x - t2.x, t0.y - t2.y);
                            Zw[j + 3 * p] = make_double2(t1.x - t3.x, t1.y - t3.y);
                        }
                    }
                    p *= 4;
                } else {
                    const int tstep = M / (p * 2);
                    double2 u[8][2];
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const int i = min(lane + 64 * b, (M >> 1) - 1);
                        u[b][0] = Zw[i]; u[b][1] = Zw[i + (M >> 1)];
                    }
                    exact_wave_sync();
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        const int i = lane + 64 * b;
                        if (i < (M >> 1)) {
                            const int k = i & (p - 1);
                            const int j = ((i - k) << 1) + k;
                            const double2 w1 = tws[k * tstep];
                            const double2 u1 = make_double2(u[b][1].x * w1.x - u[b][1].y * w1.y, u[b][1].x * w1.y + u[b][1].y * w1.x);
                            Zw[j] = make_double2(u[b][0].x + u1.x, u[b][0].y + u1.y);
                            Zw[j + p] = make_double2(u[b][0].x - u1.x, u[b][0].y - u1.y);
                        }
                    }
                    p *= 2;
                }
                exact_wave_sync();
            }
        } else {
        for (int h = 1; h < M; h <<= 1) {
                const int ts = M / (2 * h);                          // exp(-2 pi i k / (2h)) = tws[k * M / (2h)] = twiddle64[2 k M / (2h)]
                for (int b = lane; b < (M >> 1); b += 64) {
                    const int k = b & (h - 1);
                    const int i0 = ((b - k) << 1) + k, i1 = i0 + h;
                    const double2 w = tws ? tws[k * ts] : s.twiddle64[2 * k * ts];
                    const double2 u = Zw[i0], t = Zw[i1];
                    const double tr = t.x * w.x - t.y * w.y, ti = t.x * w.y + t.y * w.x;
                    Zw[i0] = make_double2(u.x + tr, u.y + ti);
                    Zw[i1] = make_double2(u.x - tr, u.y - ti);
                }
                exact_wave_sync();
            }
        }
        // split: X[k] = (Z[k] + conj Z[M-k]) / 2 + exp(-2 pi i k / W) (Z[k] - conj Z[M-k]) / (2i), k = 0 .. M
        auto magnitude = [&](int k, double2 w) -> double {
            const double2 a = Zw[k & (M - 1)], b = Zw[(M - k) & (M - 1)];
            const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);          // even part
            const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);          // (Z[k] - conj Z[M-k]) / 2
            const double orr = di, oi = -dr;                                       // ... / i
            const double xr = er + orr * w.x - oi * w.y, xi = ei + orr * w.y + oi * w.x;
            return sqrt(xr * xr + xi * xi);
        };
        if (ACC_REGS) {
            double2 w[kAcc];
#pragma unroll
            for (int jj = 0; jj < kAcc; ++jj) w[jj] = s.twiddle64[min(lane + 64 * jj, M)];
#pragma unroll
            for (int jj = 0; jj < kAcc; ++jj) { const int k = min(lane + 64 * jj, M); const double m = magnitude(k, w[jj]); acc[jj] += (lane + 64 * jj <= M) ? m : 0.0; }
        } else {
            for (int k0 = 0; k0 <= M; k0 += 512) {
                double2 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = s.twiddle64[min(k0 + 64 * u + lane, M)];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int k = k0 + 64 * u + lane; if (k <= M) accw[k] += magnitude(k, w[u]); }
            }
        }
    }
    double part = 0.0;
    if (ACC_REGS) {
#pragma unroll
        for (int jj = 0; jj < kAcc; ++jj) { const int k = lane + 64 * jj; acc[jj] = acc[jj] / (double)C; if (k < F) part += acc[jj] * acc[jj]; }
    } else {
        exact_wave_sync();
        for (int k = lane; k < F; k += 64) { const double m = accw[k] / (double)C; accw[k] = m; part += m * m; }
        exact_wave_sync();
    }
    const double nrm = sqrt(wave_sum_f64(part));
    double* out = s.u64 + (int64_t)clip * s.u64_clip_stride + fr * (int64_t)s.FS;
    if (ACC_REGS) {
#pragma unroll
        for (int jj = 0; jj < kAcc; ++jj) { const int k = lane + 64 * jj; if (k < s.FS) out[k] = (k < F) ? acc[jj] / nrm : 0.0; }
    } else {
        for (int k = lane; k < s.FS; k += 64) out[k] = (k < F) ? accw[k] / nrm : 0.0;
    }
}

// The same for the 2048-sample window, with the 1024-point transform in the wave's registers (fft_wave_f64.h): ex is the wave's
// private exchange region (kExPitch double2), which also takes the spectrum for the split pass.
__device__ __forceinline__ void wave_unit_row_f64_reg(const ExactSource& s, int clip, int64_t fr, double2* ex, const f64fft::Twiddles& tw, int lane) {
    constexpr int M = f64fft::kRegN;
    const int F = s.F, C = s.n_channels;
    const int64_t s0 = s.frame_sample0 + fr * (int64_t)s.H;
    const float* hi = s.hi + (int64_t)clip * s.clip_stride;
    const float* lo = s.lo ? s.lo + (int64_t)clip * s.clip_stride : nullptr;
    // without remainders the same loads read the samples again and count for nothing: a wave-uniform branch around every
    // load kept them from being issued together
    const float* lo_or_hi = lo ? lo : hi;
    const double lo_scale = lo ? 1.0 : 0.0;
    // the part [rel_lo, rel_hi) of the frame's W samples that exists (relative to its first sample s0; a frame over an edge
    // of the clip -- or beyond it -- reads clamped positions and counts them as zero)
    const int W2 = 2 * M;
    int rel_lo = (int)min(max(-s0, (int64_t)0), (int64_t)W2);
    int rel_hi = (int)min(max(s.n_samples - s0, (int64_t)0), (int64_t)W2);
    const bool none = rel_hi <= rel_lo;
    const int64_t anchor = none ? 0 : s0;                      // (nothing of the frame exists: any readable address will do)
    if (none) rel_hi = rel_lo + 1;
    const float* frame_hi = hi + anchor * C;
    const float* frame_lo = lo_or_hi + anchor * C;
    double acc[17];
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) acc[jj] = 0.0;
    USTAMP_DECL
    for (int c = 0; c < C; ++c) {
        double2 v[16];
        // Samples 2r and 2r+1 of a stereo clip are ONE 16-byte load per lane with both channels in it (an 8-byte load for a
        // mono clip): lanes read consecutive addresses. As two 4-byte loads per channel, 16 bytes apart from lane to lane,
        // the texture path took 16 us per channel of a transform's 68; the second channel re-reads the lines from cache.
        const bool pair_ok = !none && rel_lo == 0 && rel_hi == W2;          // (wave-uniform: the whole frame exists)
        if (pair_ok && C == 2) {
            struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 h4[8], l4[8];
                double2 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int r = 64 * (8 * half + u) + lane;
                    const F4 a4 = *reinterpret_cast<const F4*>(frame_hi + 4 * r), b4 = *reinterpret_cast<const F4*>(frame_lo + 4 * r);
                    h4[u] = make_float4(a4.x, a4.y, a4.z, a4.w); l4[u] = make_float4(b4.x, b4.y, b4.z, b4.w);
                    w[u] = *reinterpret_cast<const double2*>(s.window64 + 2 * r);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a0 = c ? h4[u].y : h4[u].x, a1 = c ? h4[u].w : h4[u].z, b0 = c ? l4[u].y : l4[u].x, b1 = c ? l4[u].w : l4[u].z;
                    v[8 * half + u] = make_double2(((double)a0 + lo_scale * (double)b0) * w[u].x, ((double)a1 + lo_scale * (double)b1) * w[u].y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (pair_ok && C == 1) {
            struct __attribute__((packed, aligned(4))) F2 { float x, y; };
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = 64 * u + lane;
                const F2 a2 = *reinterpret_cast<const F2*>(frame_hi + 2 * r), b2 = *reinterpret_cast<const F2*>(frame_lo + 2 * r);
                const double2 w = *reinterpret_cast<const double2*>(s.window64 + 2 * r);
                v[u] = make_double2(((double)a2.x + lo_scale * (double)b2.x) * w.x, ((double)a2.y + lo_scale * (double)b2.y) * w.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                    // eight points per lane in flight, no branch around a load
            float h0[8], h1[8], l0[8], l1[8];
            double2 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                // positions relative to the frame's first sample, in 32 bits (the 64-bit clamps and products of absolute
                // sample numbers were 1 500 quarter-rate instructions per channel: most of the transform's load phase)
                const int r = 64 * (8 * half + u) + lane;
                const int q0 = min(max(2 * r, rel_lo), rel_hi - 1), q1 = min(max(2 * r + 1, rel_lo), rel_hi - 1);
                h0[u] = frame_hi[q0 * C + c]; h1[u] = frame_hi[q1 * C + c];
                l0[u] = frame_lo[q0 * C + c]; l1[u] = frame_lo[q1 * C + c];       // (no branch around a load: see lo_scale)
                w[u] = *reinterpret_cast<const double2*>(s.window64 + 2 * r);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p0 = 2 * (64 * (8 * half + u) + lane);
                const bool in0 = !none && p0 >= rel_lo && p0 < rel_hi, in1 = !none && p0 + 1 >= rel_lo && p0 + 1 < rel_hi;
                v[8 * half + u] = make_double2(in0 ? ((double)h0[u] + lo_scale * (double)l0[u]) * w[u].x : 0.0,
                                               in1 ? ((double)h1[u] + lo_scale * (double)l1[u]) * w[u].y : 0.0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        USTAMP(0)
        f64fft::wave_fft1024(v, ex, tw, lane);
        __builtin_amdgcn_sched_barrier(0);
        USTAMP(1)
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) ex[lane + 64 * sl] = v[sl];
        f64fft::wave_fence();
        // split: X[k] = (Z[k] + conj Z[M-k]) / 2 + exp(-2 pi i k / W) (Z[k] - conj Z[M-k]) / (2i), k = 0 .. M
        // (six bins per lane at a time: with all seventeen unrolled the compiler hoists every load and needs 500 registers)
#pragma unroll
        for (int j0 = 0; j0 < 18; j0 += 6) {
            double2 tw_k[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) tw_k[u] = s.twiddle64[min(lane + 64 * (j0 + u), M)];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int jj = j0 + u;
                if (jj < 17) {
                    const int k = min(lane + 64 * jj, M);
                    const double2 a = ex[k & (M - 1)], b = ex[(M - k) & (M - 1)];
                    const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
                    const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
                    const double orr = di, oi = -dr;
                    const double xr = er + orr * tw_k[u].x - oi * tw_k[u].y, xi = ei + orr * tw_k[u].y + oi * tw_k[u].x;
                    const double m = sqrt(xr * xr + xi * xi);
                    acc[jj] += (lane + 64 * jj <= M) ? m : 0.0;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        f64fft::wave_fence();
        USTAMP(2)
    }
    double part = 0.0;
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) { const int k = lane + 64 * jj; acc[jj] = acc[jj] / (double)C; if (k < F) part += acc[jj] * acc[jj]; }
    const double nrm = sqrt(wave_sum_f64(part));
    double* out = s.u64 + (int64_t)clip * s.u64_clip_stride + fr * (int64_t)s.FS;
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) { const int k = lane + 64 * jj; if (k < s.FS) out[k] = (k < F) ? acc[jj] / nrm : 0.0; }
    USTAMP(3)
}

// Which transform a kernel instance carries (one per instance: their registers must not add up): the register transform of the
// 2048-sample window, the radix-4 Stockham transform in LDS (W < 2048), the radix-2 one
// (W > 2048).
constexpr int kFftReg = 0, kFftLds4 = 1, kFftLdsAcc = 2;
template <int V>
__device__ __forceinline__ void unit_row_variant(const ExactSource& src, int logM, int clip, int64_t fr, double2* Zw, double* accw,
                                                 const double2* tws, const f64fft::Twiddles& rtw, int lane) {
    if constexpr (V == kFftReg) wave_unit_row_f64_reg(src, clip, fr, Zw, rtw, lane);
    else if constexpr (V == kFftLds4) wave_unit_row_f64<false, true>(src, logM, clip, fr, Zw, accw, tws, lane);
    else wave_unit_row_f64<false, false>(src, logM, clip, fr, Zw, accw, tws, lane);
}

// Which transform a launch carries and what it needs of the LDS.
struct FftPlan { int variant, tw_count, wave_bytes, fixed_bytes; };
static FftPlan fft_plan(int W, bool allow_reg = true) {
    // W = 2048: the register transform (62 us per stereo frame alone; the radix-4 LDS transform 80 us with a third of the registers)
    const int Mh = W / 2;
    FftPlan p{};
    if (W == 2048 && allow_reg) {
        p.variant = kFftReg; p.tw_count = 0; p.wave_bytes = f64fft::kExPitch * (int)sizeof(double2);
        p.fixed_bytes = f64fft::kTwCount * (int)sizeof(double2);
    } else if (W <= 2048) {
        p.variant = kFftLds4; p.tw_count = Mh;                          // exp(-2 pi i m / M), m < M: radix-4 reaches 3M/4
        p.wave_bytes = Mh * (int)sizeof(double2) + (Mh + 8) * (int)sizeof(double);
        p.fixed_bytes = p.tw_count * (int)sizeof(double2);
    } else {
        p.variant = kFftLdsAcc; p.tw_count = Mh / 2;
        p.wave_bytes = Mh * (int)sizeof(double2) + (Mh + 8) * (int)sizeof(double);
        p.fixed_bytes = p.tw_count * (int)sizeof(double2);
        if (p.fixed_bytes + p.wave_bytes > 150 * 1024) { p.tw_count = 0; p.fixed_bytes = 0; }      // W = 8192: twiddles from global memory
    }
    return p;
}

// The float64 unit rows of the frames the first pass queued (PeakArgs::frame_list): the lean kernel between the first pass and
// local_maxima_lite_kernel (peaks_wave.hip). Frames a workgroup of the general kernel needs beyond these are transformed there,
// on demand. (The one-wavefront-per-frame forms of this kernel -- REPET_EXACT_FFT -- are in the history: round 3.)
struct UnitRowsArgs {
    ExactSource src; const int* frame_list; unsigned int* stats; unsigned int gen;
    int logM;
};

// The same rows by ONE WORKGROUP per frame (windows up to 8192): the lean kernel's default. A transform is a chain of
// dependent steps; one wavefront per frame (above) walks it alone -- 62 us per stereo frame however many frames there are --
// while 256 threads hold ONE radix-4 butterfly each per pass of a 1024-point transform: loads in one round trip (a stereo
// frame's samples and remainders as one 16-byte load per thread and point pair, both channels at once), five in-place
// Stockham passes in LDS with two barriers each, the split pass and the magnitude sums in registers (a thread owns bins
// tid + 256 q), block-wide norm. LDS at W <= 2048 (round 5): twiddles exp(-2 pi i m / M) [M / 2] and TWO transforms [2][M] --
// the channels of a stereo frame go through the passes together -- 40 KB, 125 VGPRs: four workgroups per CU as before,
// the chain of passes walked once per frame instead of once per channel (peaks + sort at cfg 2: 0.244 -> 0.237 ms, same bits).
// KQ: radix-4 butterflies per thread and pass = ceil(M / 1024): 1 for W <= 2048 (few registers: four workgroups per CU), 4 up to W = 8192
template <int KQ>
__global__ __launch_bounds__(256, KQ == 1 ? 4 : 1) void unit_rows_f64_wg_kernel(UnitRowsArgs x) {
    constexpr int kWgQ = KQ;
    constexpr int kCP = KQ == 1 ? 2 : 1;        // channels per pass (two transforms of up to 1 024 points fit the LDS beside the twiddles)
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    const ExactSource& s = x.src;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = s.W, M = W >> 1, F = s.F, C = s.n_channels;
    if (x.stats[11] >= x.stats[10]) return;
    // twiddles exp(-2 pi i m / M): the first half only where two transforms share the LDS (w^(m + M/2) = -w^m, exact), so that
    // a workgroup takes 8 + 2 x 16 KB = a quarter of the CU's LDS and four of them -- every queued frame of a clip at once --
    // fit (the block sum at the end borrows the transform's space)
    constexpr bool kHalfTw = KQ == 1;
    const int n_tw = kHalfTw ? M >> 1 : M;
    double2* tw_lds = reinterpret_cast<double2*>(wg_smem);
    double2* Z = tw_lds + n_tw;
    double* red = reinterpret_cast<double*>(Z);
    for (int k = tid; k < n_tw; k += 256) tw_lds[k] = s.twiddle64[2 * k];
    auto tw = [&](int m) -> double2 {
        if constexpr (!kHalfTw) return tw_lds[m];
        else {
            const double2 v = tw_lds[m & ((M >> 1) - 1)];
            return (m & (M >> 1)) ? make_double2(-v.x, -v.y) : v;
        }
    };
    const unsigned int n_frames = x.stats[10];
    const int nq = (M + 255) >> 8;            // point pairs per thread (1 .. 16 for M = 4096: loops below run over q < nq)
    // (frames dealt by position: a shared cursor hands out about 88 slots per microsecond, and the 1 024 workgroups' first
    // requests alone took 12 us)
    for (unsigned int slot = blockIdx.x; slot < n_frames; slot += gridDim.x) {
        __syncthreads();
        const int64_t lin = x.frame_list[slot];
        if (s.u64_gen[lin] == x.gen) continue;                 // already there (workgroup-uniform)
        const int clip = (int)(lin / s.gen_clip_stride);
        const int64_t fr = lin - (int64_t)clip * s.gen_clip_stride;
        unsigned int* g = s.u64_gen + lin;
        const int64_t s0 = s.frame_sample0 + fr * (int64_t)s.H;
        const float* hi = s.hi + (int64_t)clip * s.clip_stride;
        const float* lo = s.lo ? s.lo + (int64_t)clip * s.clip_stride : hi;
        const double lo_scale = s.lo ? 1.0 : 0.0;
        int rel_lo = (int)min(max(-s0, (int64_t)0), (int64_t)W);
        int rel_hi = (int)min(max(s.n_samples - s0, (int64_t)0), (int64_t)W);
        const bool none = rel_hi <= rel_lo;
        if (none) { rel_lo = 0; rel_hi = 1; }
        const float* frame_hi = hi + (none ? 0 : s0) * C;
        const float* frame_lo = lo + (none ? 0 : s0) * C;
        const bool whole = !none && rel_lo == 0 && rel_hi == W;
        double acc[kWgQ * 4 + 1];
#pragma unroll
        for (int q = 0; q < kWgQ * 4 + 1; ++q) acc[q] = 0.0;
        // Channels in PAIRS (windows up to 2048: two transforms fit the LDS): a transform is a chain of dependent passes with two
        // barriers each, and a workgroup's life IS that chain (all queued frames run at once, one per workgroup) -- with the two
        // channels of a stereo frame in the same passes the chain is walked once instead of twice (round 5: 33 -> 19 us alone).
        // The butterflies, their order and the order in which the channels' magnitudes are added are unchanged: the same bits.
        for (int c0 = 0; c0 < C; c0 += kCP) {
            const int nc = min(kCP, C - c0);
            // the channels' windowed points, natural order (no branch around a load; positions frame-relative, 32-bit)
            for (int q0 = 0; q0 < nq; q0 += 4) {
                float h0[kCP][4], h1[kCP][4], l0[kCP][4], l1[kCP][4];
                double2 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = min(tid + 256 * (q0 + u), M - 1);
                    const int a0 = min(max(2 * r, rel_lo), rel_hi - 1), a1 = min(max(2 * r + 1, rel_lo), rel_hi - 1);
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc) {
                        const int c = min(c0 + cc, C - 1);
                        h0[cc][u] = frame_hi[a0 * C + c]; h1[cc][u] = frame_hi[a1 * C + c];
                        l0[cc][u] = frame_lo[a0 * C + c]; l1[cc][u] = frame_lo[a1 * C + c];
                    }
                    w[u] = *reinterpret_cast<const double2*>(s.window64 + 2 * r);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int r = tid + 256 * (q0 + u);
                    const int p0 = 2 * r;
                    const bool in0 = !none && (whole || (p0 >= rel_lo && p0 < rel_hi)), in1 = !none && (whole || (p0 + 1 >= rel_lo && p0 + 1 < rel_hi));
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc)
                        if (r < M && cc < nc) Z[cc * M + r] = make_double2(in0 ? ((double)h0[cc][u] + lo_scale * (double)l0[cc][u]) * w[u].x : 0.0,
                                                                          in1 ? ((double)h1[cc][u] + lo_scale * (double)l1[cc][u]) * w[u].y : 0.0);
                }
            }
            __syncthreads();
            for (int p = 1; p < M;) {
                if (M / p >= 4) {
                    const int tstep = M / (p * 4), quarter = M >> 2;
                    double2 u[kCP][kWgQ][4];
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc)
#pragma unroll
                        for (int b = 0; b < kWgQ; ++b) {
                            const int i = min(tid + 256 * b, quarter - 1);
                            const double2* Zc = Z + (cc < nc ? cc : 0) * M;
                            u[cc][b][0] = Zc[i]; u[cc][b][1] = Zc[i + quarter]; u[cc][b][2] = Zc[i + 2 * quarter]; u[cc][b][3] = Zc[i + 3 * quarter];
                        }
                    __syncthreads();
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc)
#pragma unroll
                        for (int b = 0; b < kWgQ; ++b) {
                            const int i = tid + 256 * b;
                            if (i < quarter && cc < nc) {
                                double2* Zc = Z + cc * M;
                                const int k = i & (p - 1);
                                const int j = ((i - k) << 2) + k;
                                double2 u0 = u[cc][b][0], u1 = u[cc][b][1], u2 = u[cc][b][2], u3 = u[cc][b][3];
                                if (p > 1) {
                                    const double2 w1 = tw(k * tstep), w2 = tw(2 * k * tstep), w3 = tw(3 * k * tstep);
                                    u1 = make_double2(u1.x * w1.x - u1.y * w1.y, u1.x * w1.y + u1.y * w1.x);
                                    u2 = make_double2(u2.x * w2.x - u2.y * w2.y, u2.x * w2.y + u2.y * w2.x);
                                    u3 = make_double2(u3.x * w3.x - u3.y * w3.y, u3.x * w3.y + u3.y * w3.x);
                                }
                                const double2 t0 = make_double2(u0.x + u2.x, u0.y + u2.y), t1 = make_double2(u0.x - u2.x, u0.y - u2.y);
                                const double2 t2 = make_double2(u1.x + u3.x, u1.y + u3.y), d = make_double2(u1.x - u3.x, u1.y - u3.y);
                                const double2 t3 = make_double2(d.y, -d.x);                       // -i d
                                Zc[j] = make_double2(t0.x + t2.x, t0.y + t2.y);
                                Zc[j + p] = make_double2(t1.x + t3.x, t1.y + t3.y);
                                Zc[j + 2 * p] = make_double2(t0.x - t2.x, t0.y - t2.y);
                                Zc[j + 3 * p] = make_double2(t1.x - t3.x, t1.y - t3.y);
                            }
                        }
                    p *= 4;
                } else {
                    const int tstep = M / (p * 2), half = M >> 1;
                    double2 u[kCP][2 * kWgQ][2];
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc)
#pragma unroll
                        for (int b = 0; b < 2 * kWgQ; ++b) {
                            const int i = min(tid + 256 * b, half - 1);
                            const double2* Zc = Z + (cc < nc ? cc : 0) * M;
                            u[cc][b][0] = Zc[i]; u[cc][b][1] = Zc[i + half];
                        }
                    __syncthreads();
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc)
#pragma unroll
                        for (int b = 0; b < 2 * kWgQ; ++b) {
                            const int i = tid + 256 * b;
                            if (i < half && cc < nc) {
                                double2* Zc = Z + cc * M;
                                const int k = i & (p - 1);
                                const int j = ((i - k) << 1) + k;
                                const double2 w1 = tw(k * tstep);
                                const double2 u1 = make_double2(u[cc][b][1].x * w1.x - u[cc][b][1].y * w1.y, u[cc][b][1].x * w1.y + u[cc][b][1].y * w1.x);
                                Zc[j] = make_double2(u[cc][b][0].x + u1.x, u[cc][b][0].y + u1.y);
                                Zc[j + p] = make_double2(u[cc][b][0].x - u1.x, u[cc][b][0].y - u1.y);
                            }
                        }
                    p *= 2;
                }
                __syncthreads();
            }
            // split: X[k] = (Z[k] + conj Z[M-k]) / 2 + exp(-2 pi i k / W) (Z[k] - conj Z[M-k]) / (2i), k = tid + 256 q <= M
#pragma unroll
            for (int q = 0; q < kWgQ * 4 + 1; ++q) {
                const int kk = tid + 256 * q;
                if (q <= nq) {                                   // (q == nq: bin M, thread 0)
                    const int k = min(kk, M);
                    const double2 wk = s.twiddle64[k];
#pragma unroll
                    for (int cc = 0; cc < kCP; ++cc) {
                        if (cc >= nc) break;
                        const double2* Zc = Z + cc * M;
                        const double2 a = Zc[k & (M - 1)], b = Zc[(M - k) & (M - 1)];
                        const double er = 0.5 * (a.x + b.x), ei = 0.5 * (a.y - b.y);
                        const double dr = 0.5 * (a.x - b.x), di = 0.5 * (a.y + b.y);
                        const double orr = di, oi = -dr;
                        const double xr = er + orr * wk.x - oi * wk.y, xi = ei + orr * wk.y + oi * wk.x;
                        const double m = sqrt(xr * xr + xi * xi);
                        acc[q] += (kk <= M) ? m : 0.0;
                    }
                }
            }
            __syncthreads();                                     // (the next channels overwrite Z)
        }
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < kWgQ * 4 + 1; ++q) {
            const int k = tid + 256 * q;
            acc[q] = acc[q] / (double)C;
            if (k < F) part += acc[q] * acc[q];
        }
        part = wave_sum_f64(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        const double nrm = sqrt(red[0] + red[1] + red[2] + red[3]);
        double* out = s.u64 + (int64_t)clip * s.u64_clip_stride + fr * (int64_t)s.FS;
#pragma unroll
        for (int q = 0; q < kWgQ * 4 + 1; ++q) {
            const int k = tid + 256 * q;
            if (k < s.FS) out[k] = (k < F) ? acc[q] / nrm : 0.0;
        }
        if (tid == 0) { *g = x.gen; stat_add(x.stats, 9, 1u); }
    }
}

template <int V>
__global__ __launch_bounds__(kExactThreads) void local_maxima_exact_kernel(ExactArgs x) {
    extern __shared__ __attribute__((aligned(16))) unsigned char exact_smem[];
    __shared__ int sh_slot, n_near, n_list, n_list2, n_peak, n_band, n_above;
    __shared__ float cutv[2];
    __shared__ double red[kExactWaves], red2[kExactWaves];
    __shared__ float s_pval[kRankLds];
    __shared__ int s_pidx[kRankLds];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = x.src.W >> 1;
    RowScratch R = carve_row(x.scratch + (size_t)blockIdx.x * x.scratch_per_wg, x.n4);
    unsigned int* stats = x.a.stats;
    if (stats[5] >= stats[4]) return;                          // (the usual case: nothing for the general path)
    // a butterfly stage waits for its twiddle: out of global memory that was ten far round trips per transform
    const double2* tws = nullptr;
    unsigned char* work = exact_smem;
    f64fft::Twiddles rtw{nullptr, nullptr};
    if (x.reg_fft) {
        rtw = f64fft::load_twiddles(reinterpret_cast<double2*>(exact_smem), x.src.twiddle64, tid, kExactThreads);
        work += (size_t)f64fft::kTwCount * sizeof(double2);
    } else if (x.tw_count > 0) {
        double2* t = reinterpret_cast<double2*>(exact_smem);
        for (int k = tid; k < x.tw_count; k += kExactThreads) t[k] = x.src.twiddle64[2 * k];       // exp(-2 pi i k / M)
        tws = t;
        work += (size_t)x.tw_count * sizeof(double2);
    }

    for (;;) {
        __syncthreads();
        if (tid == 0) sh_slot = (int)atomicAdd(&stats[5], 1u);
        __syncthreads();
        const int slot = sh_slot;
        if (slot >= (int)stats[4]) return;
        PeakArgs a = x.a;
        const int64_t r = x.redo_list[2 * slot];
        const int clip = x.redo_list[2 * slot + 1];
        a.M += clip * a.m_stride;
        a.idx += clip * a.idx_stride;
        a.count += clip * a.cnt_stride;
        a.unit += clip * a.unit_stride;
        const int n = a.n, d = a.d;
        const int64_t j = a.row0 + r;
        const float dlt = a.delta;
        const double d2 = x.delta2, thr64 = a.min_value64;
        auto lag_of = [&](int i) -> int { int l = (int)(j - i) % n; return l < 0 ? l + n : l; };
        auto fetch = [&](int i) -> float {
            if (a.mode == 0) return a.M[j * a.pitch + i];
            const int l = lag_of(i);
            return a.M[(a.mode == 2 ? j - a.shift : j - l - a.shift) * a.pitch + l];
        };
        auto frame_of = [&](int i) -> int64_t { return a.mode == 0 ? (int64_t)i : j - lag_of(i) - a.shift; };
        auto out_index = [&](int i) -> int { return a.mode == 0 ? i : (int)(j - lag_of(i) - a.shift); };
        const int64_t self_fr = j - a.shift;
        const float* self32 = a.unit + self_fr * (int64_t)a.unit_pitch;
        const int len4 = a.unit_pitch >> 2;

        XSTAMP_DECL
        // ---- the row and its candidates ---------------------------------------------------------------------------
        if (tid == 0) { n_near = 0; n_list = 0; n_list2 = 0; n_peak = 0; n_band = 0; n_above = 0; }
        __syncthreads();
        for (int i0 = 0; i0 < x.n4; i0 += 8 * kExactThreads) {          // eight loads in flight per thread, no branch around them
            float e[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = fetch(min(i0 + kExactThreads * u + tid, n - 1));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + kExactThreads * u + tid;
                if (i < x.n4) { R.v[i] = i < n ? nan_to_inf(e[u]) : -INFINITY; R.st[i] = 0u; }
            }
        }
        __syncthreads();
        auto candidate = [&](int i, float vi, float mx) {       // the first pass's test on a value and its window maximum
            if (vi < mx - dlt) return;
            const bool sure = (vi >= a.min_value + dlt) && (vi > mx + dlt);
            if (sure) { R.st[i] = kSure; const int p = atomicAdd(&n_peak, 1); R.pidx[p] = i; R.pval[p] = vi; }
            else { R.st[i] = kNear | kNeed1; R.near[atomicAdd(&n_near, 1)] = i; R.list[atomicAdd(&n_list, 1)] = i; }
        };
        if (x.scan_in_lds) {
            // window maxima M_w[p] = max(v[p .. p+w-1]), w = 2^floor(log2 d), by doubling between two LDS buffers, a chunk
            // of the row (with a halo of d elements) at a time; left = max(M_w[i-d], M_w[i-w]), right = max(M_w[i+1],
            // M_w[i+d-w+1]) as in peaks.hip
            int w = 1;
            while (2 * w <= d) w *= 2;
            const int blen = kScanChunk + 2 * d + 8;
            float* bufA = reinterpret_cast<float*>(work);
            float* bufB = bufA + blen;
            for (int c0 = 0; c0 < n; c0 += kScanChunk) {
                const int lo = c0 - d, t_hi = c0 + kScanChunk < n ? c0 + kScanChunk : n, len = t_hi + d - lo;
                __syncthreads();
                for (int e0 = 0; e0 < len; e0 += 8 * kExactThreads) {       // (batched: eight loads in flight per thread)
                    float t8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t8[u] = R.v[min(max(lo + e0 + kExactThreads * u + tid, 0), n - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + kExactThreads * u + tid, i = lo + e;
                        if (e < len) bufA[e] = (i >= 0 && i < n) ? t8[u] : -INFINITY;
                    }
                }
                __syncthreads();
                float* src = bufA;
                float* dst = bufB;
                for (int step = 1; step < w; step *= 2) {
                    for (int e = tid; e < len; e += kExactThreads) dst[e] = fmaxf(src[e], e + step < len ? src[e + step] : -INFINITY);
                    __syncthreads();
                    float* t = src; src = dst; dst = t;
                }
                for (int i = c0 + tid; i < t_hi; i += kExactThreads) {
                    const float vi = R.v[i];
                    if (!(vi < INFINITY) || vi < a.min_value - dlt) continue;
                    float mx = -INFINITY;
                    if (d > 0) {
                        const int e = i - lo;
                        mx = fmaxf(fmaxf(src[e - d], src[e - w]), fmaxf(src[e + 1], src[e + d - w + 1]));
                    }
                    candidate(i, vi, mx);
                }
            }
        } else {
            for (int i = tid; i < n; i += kExactThreads) {      // (windows too wide for the LDS buffers)
                const float vi = R.v[i];
                if (!(vi < INFINITY) || vi < a.min_value - dlt) continue;
                const int k_lo = i - d > 0 ? i - d : 0, k_hi = i + d < n - 1 ? i + d : n - 1;
                float mx = -INFINITY;
                for (int k = k_lo; k <= k_hi; ++k) {
                    if (k != i) mx = fmaxf(mx, R.v[k]);
                    if (mx > vi + dlt) break;                   // a safe "no": nothing more to learn
                }
                candidate(i, vi, mx);
            }
        }
        __syncthreads();
        const int nn = n_near;
        // every (near-tied element q, window position) pair whose value is within delta below the element (or above it): the
        // element's rivals, walked with all threads
        const int win = 2 * d + 1;
        const long long n_pairs = (long long)nn * win;
        auto for_rival_pairs = [&](auto&& fn) {
            for (long long e = tid; e < n_pairs; e += kExactThreads) {
                const int q = (int)(e / win);
                const int i = R.near[q];
                const int k = i - d + (int)(e - (long long)q * win);
                if (k == i || k < 0 || k >= n) continue;
                if (!(R.v[k] >= R.v[i] - dlt)) continue;
                fn(q, i, k);
            }
        };
        for_rival_pairs([&](int, int, int k) { if (!(atomicOr(&R.st[k], kNeed1) & kNeed1)) R.list[atomicAdd(&n_list, 1)] = k; });
        __syncthreads();

        // level-1 values of the elements appended to R.list since the last call (whoever sets kNeed1 first appends)
        int list_done = 0;
        auto level1 = [&]() {
            __syncthreads();
            const int items = n_list;
            for (int it = list_done + 2 * wave; it < items; it += 2 * kExactWaves) {
                const bool two = it + 1 < items;
                const int i0 = R.list[it], i1 = R.list[two ? it + 1 : it];
                double e0, e1;
                exact_similarity2(self32, a.unit + frame_of(i0) * (int64_t)a.unit_pitch, a.unit + frame_of(i1) * (int64_t)a.unit_pitch,
                                  len4, lane, &e0, &e1);
                if (lane == 0) { R.val[i0] = e0; if (two) R.val[i1] = e1; }
            }
            list_done = items;
            __syncthreads();
        };
        // float64 unit rows of the frames behind everything flagged kNeed2 (and of the row itself), level-2 values
        int list2_done = 0;
        auto level2 = [&]() {
            __syncthreads();
            const int first = list2_done, items = n_list2 - list2_done;
            list2_done = n_list2;
            if (items == 0) return;
            unsigned* gens = x.src.u64_gen + (int64_t)clip * x.src.gen_clip_stride;
            if (wave < x.fft_waves) {
                double2* Zw = reinterpret_cast<double2*>(work + (size_t)wave * x.wave_bytes);
                double* accw = reinterpret_cast<double*>(Zw + M);
                for (int it = wave - 1; it < items; it += x.fft_waves) {        // item -1: the row's own frame
                    const int64_t fr = it < 0 ? self_fr : frame_of(R.list2[first + it]);
                    // (a frame another workgroup is transforming right now is transformed twice: same bits, no waiting)
                    if (__hip_atomic_load(&gens[fr], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == x.gen) continue;
                    unit_row_variant<V>(x.src, x.logM, clip, fr, Zw, accw, tws, rtw, lane);
                    __threadfence();
                    if (lane == 0) {
                        __hip_atomic_store(&gens[fr], x.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        stat_add(stats, 9, 1u);
                    }
                }
            }
            __threadfence();
            __syncthreads();
            const double* base = x.src.u64 + (int64_t)clip * x.src.u64_clip_stride;
            const double* self64 = base + self_fr * (int64_t)x.src.FS;
            double worst = 0.0;
            for (int it = wave; it < items; it += kExactWaves) {
                const int i = R.list2[first + it];
                const double e = dot_rows_f64(self64, base + frame_of(i) * (int64_t)x.src.FS, x.src.FS, lane);
                if (lane == 0) {
                    const double diff = fabs(e - R.val[i]);
                    if (diff > worst) worst = diff;
                    R.val[i] = e;
                }
            }
            if (lane == 0 && worst > 0.0) stat_max(stats, 8, (unsigned)fmin(worst * 1e12, 4.0e9));
            if (tid == 0) stat_add(stats, 6, (unsigned)items);
            __syncthreads();
        };
        // best[q] = 1 + the bit pattern of the largest rival value of near-tied element q (0: no rival). Similarities of
        // magnitude spectra are >= 0, so the patterns order like unsigned integers; a NaN rival's is above every number.
        auto best_rivals = [&]() {
            for (int q = tid; q < nn; q += kExactThreads) R.best[q] = 0ull;
            __syncthreads();
            for_rival_pairs([&](int q, int, int k) { atomicMax(&R.best[q], (unsigned long long)__double_as_longlong(R.val[k]) + 1ull); });
            __syncthreads();
        };

        XSTAMP(0)
        level1();
        XSTAMP(1)
        // a verdict closer than delta2 -- the element against its best rival, or against the threshold -- is taken again
        // from float64 spectra: of the element and of every rival that may be the best one
        best_rivals();
        for (int q = tid; q < nn; q += kExactThreads) {
            const int i = R.near[q];
            const double ei = R.val[i];
            const unsigned long long b = R.best[q];
            const bool versus_rival = b != 0ull && fabs(ei - __longlong_as_double((long long)(b - 1ull))) < d2;
            R.prank[q] = versus_rival ? 1 : 0;
            if (versus_rival || fabs(ei - thr64) < d2) { if (!(atomicOr(&R.st[i], kNeed2) & kNeed2)) R.list2[atomicAdd(&n_list2, 1)] = i; }
        }
        __syncthreads();
        for_rival_pairs([&](int q, int, int k) {
            if (R.prank[q] && R.val[k] > __longlong_as_double((long long)(R.best[q] - 1ull)) - d2) { if (!(atomicOr(&R.st[k], kNeed2) & kNeed2)) R.list2[atomicAdd(&n_list2, 1)] = k; }
        });
        XSTAMP(2)
        level2();
        XSTAMP(3)
        // the verdicts: >= the threshold and strictly above every rival
        best_rivals();
        for (int q = tid; q < nn; q += kExactThreads) {
            const int i = R.near[q];
            const double ei = R.val[i];
            const unsigned long long b = R.best[q];
            if (ei >= thr64 && (b == 0ull || ei > __longlong_as_double((long long)(b - 1ull)))) {
                const int p = atomicAdd(&n_peak, 1);
                R.pidx[p] = i;
                R.pval[p] = (float)ei;
            }
        }
        __syncthreads();

        XSTAMP(4)
        // ---- ranking: value descending, higher index first on ties (np.argsort(...)[::-1]) ------------------------
        const int np_ = n_peak;
        const int kept = np_ < a.number ? np_ : a.number;
        int* out = a.idx + r * (int64_t)a.idx_pitch;
        int differs = 0;                                        // (against the first pass's list: rows that change are counted)
        const int old_count = a.count[r];
        const float* pv = R.pval;
        const int* pi = R.pidx;
        if (np_ <= kRankLds) {
            for (int p = tid; p < np_; p += kExactThreads) { s_pval[p] = R.pval[p]; s_pidx[p] = R.pidx[p]; }
            __syncthreads();
            pv = s_pval; pi = s_pidx;
        }
        for (int p = tid; p < np_; p += kExactThreads) {
            const float v = pv[p];
            const int i = pi[p];
            int rank = 0;
            for (int q = 0; q < np_; ++q) {
                const float u = pv[q];
                rank += (u > v) || (u == v && pi[q] > i);
            }
            R.prank[p] = rank;
        }
        __syncthreads();
        if (np_ > a.number) {
            for (int p = tid; p < np_; p += kExactThreads) {
                if (R.prank[p] == a.number - 1) cutv[0] = R.pval[p];
                if (R.prank[p] == a.number) cutv[1] = R.pval[p];
            }
            __syncthreads();
            const float c_in = cutv[0], c_out = cutv[1];
            if (c_in - c_out <= dlt) {
                // every candidate within delta of the cut is re-ranked by float64 similarity (level 1, and level 2 where the
                // cut itself is closer than delta2); the candidates above that band keep their places
                const float lo = c_out - dlt, hi = c_in + dlt;
                for (int p = tid; p < np_; p += kExactThreads) {
                    const float v = R.pval[p];
                    if (v > hi) atomicAdd(&n_above, 1);
                    else if (v >= lo) { R.near[atomicAdd(&n_band, 1)] = p; const int i = R.pidx[p]; if (!(atomicOr(&R.st[i], kNeed1 | kBand) & kNeed1)) R.list[atomicAdd(&n_list, 1)] = i; }
                }
                level1();
                const int nb = n_band;
                const int above = n_above;
                auto rank_band = [&]() {
                    for (int q = tid; q < nb; q += kExactThreads) {
                        const int p = R.near[q];
                        const int i = R.pidx[p];
                        const double e = R.val[i];
                        int crank = 0;
                        for (int t = 0; t < nb; ++t) {
                            const int k = R.pidx[R.near[t]];
                            const double ek = R.val[k];
                            crank += (ek > e) || (ek == e && k > i);
                        }
                        R.prank[p] = above + crank;
                    }
                    __syncthreads();
                };
                rank_band();
                // the cut separates the lowest value kept from the highest one dropped: closer than delta2, everything
                // that may belong on the other side gets float64 spectra and the band is ranked again
                double low_kept = INFINITY, high_dropped = -INFINITY;
                for (int q = tid; q < nb; q += kExactThreads) {
                    const int p = R.near[q];
                    const double e = R.val[R.pidx[p]];
                    if (R.prank[p] < a.number) low_kept = fmin(low_kept, e); else high_dropped = fmax(high_dropped, e);
                }
                for (int sh = 32; sh > 0; sh >>= 1) {
                    low_kept = fmin(low_kept, __shfl_xor(low_kept, sh));
                    high_dropped = fmax(high_dropped, __shfl_xor(high_dropped, sh));
                }
                __syncthreads();
                if (lane == 0) { red[wave] = low_kept; red2[wave] = high_dropped; }
                __syncthreads();
                for (int k = 0; k < kExactWaves; ++k) { low_kept = fmin(low_kept, red[k]); high_dropped = fmax(high_dropped, red2[k]); }
                if (low_kept - high_dropped < d2) {
                    for (int q = tid; q < nb; q += kExactThreads) {
                        const int p = R.near[q];
                        const int i = R.pidx[p];
                        const double e = R.val[i];
                        const bool kept_now = R.prank[p] < a.number;
                        if ((kept_now && e < high_dropped + d2) || (!kept_now && e > low_kept - d2)) { if (!(atomicOr(&R.st[i], kNeed2) & kNeed2)) R.list2[atomicAdd(&n_list2, 1)] = i; }
                    }
                    level2();
                    rank_band();
                }
                __syncthreads();
            }
        }
        for (int p = tid; p < np_; p += kExactThreads) {
            const int rank = R.prank[p];
            if (rank < a.number) {
                const int o = out_index(R.pidx[p]);
                if (rank >= old_count || out[rank] != o) differs = 1;
                out[rank] = o;
            }
        }
        for (int k = kept + tid; k < a.number; k += kExactThreads) out[k] = -1;
        if (kept != old_count) differs = 1;
        const int any = __syncthreads_or(differs);
        if (tid == 0) {
            a.count[r] = kept;
            if (any) stat_add(stats, 7, 1u);
        }
        XSTAMP(5)
    }
}

}  // namespace

#ifdef REPET_EXACT_STAMPS
extern "C" int repet_debug_unit_stamps(unsigned long long* out) {
    const int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_unit_stamps), sizeof(unsigned long long) * 8);
    unsigned long long zero[8] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_stamps), zero, sizeof(zero));
    return rc;
}
#endif

size_t local_maxima_exact_scratch_bytes(int32_t n_cols, int* grid_out) {
    const int n4 = (int)round_up(n_cols, 4) + 4;
    const size_t per = round_up((int64_t)exact_row_scratch_bytes(n4), 256);
    int64_t grid = 512;
    const int64_t budget = (int64_t)2 << 30;
    if ((int64_t)per * grid > budget) grid = std::max<int64_t>(64, budget / (int64_t)per);
    if (grid_out) *grid_out = (int)grid;
    return per * (size_t)grid;
}

hipError_t launch_local_maxima_exact(const float* M, int64_t row0, int32_t n_cols, int64_t pitch, int32_t mode, float min_value,
                                     int32_t d, int32_t number, int32_t* idx, int32_t idx_pitch, int32_t* count, hipStream_t s,
                                     int64_t shift, const PeakRefine* refine, const PeakBatch* batch, const ExactSource& src,
                                     void* scratch) {
    if (!refine || !refine->unit_rows || !(refine->delta > 0.0f) || !refine->redo_list || !refine->stats) return hipSuccess;
    if (d > n_cols) d = n_cols;
    ExactArgs x{};
    PeakArgs& a = x.a;
    a.M = M; a.row0 = row0; a.n = n_cols; a.pitch = pitch; a.mode = mode; a.min_value = min_value; a.d = d;
    a.number = number; a.idx = idx; a.idx_pitch = idx_pitch; a.count = count; a.shift = shift;
    a.unit = refine->unit_rows; a.unit_pitch = refine->pitch; a.delta = refine->delta;
    a.min_value64 = refine->min_value; a.stats = refine->stats;
    if (batch && batch->n_batch > 0) {
        a.m_stride = batch->m_stride; a.idx_stride = batch->idx_stride; a.cnt_stride = batch->cnt_stride;
        a.unit_stride = batch->unit_stride;
    }
    x.delta2 = refine->delta2; x.redo_list = refine->redo_list; x.src = src; x.gen = refine->gen;
    int logM = 0;
    while ((2 << logM) < src.W) ++logM;
    x.logM = logM;
    int grid = 0;
    const size_t total = local_maxima_exact_scratch_bytes(n_cols, &grid);
    x.n4 = (int)round_up(n_cols, 4) + 4;
    x.scratch = static_cast<unsigned char*>(scratch);
    x.scratch_per_wg = total / grid;
    // LDS: the stage twiddles, then one region that holds up to four transforms side by side (one per wavefront) and, before
    // them, the two window-maximum buffers of the scan. 76 KB leaves room for two workgroups per CU.
    // (never the register transform here: beside the row's own state it needs 31 registers more than a wave can have, and
    // the transforms of this kernel are the rare ones)
    const FftPlan plan = fft_plan(src.W, false);
    x.reg_fft = 0; x.tw_count = plan.tw_count; x.wave_bytes = plan.wave_bytes;
    const int small = 76 * 1024, large = 150 * 1024;
    const int fixed = plan.fixed_bytes;
    int waves = 4;
    while (waves > 1 && fixed + waves * x.wave_bytes > small) --waves;
    if (waves < 2) { waves = 4; while (waves > 1 && fixed + waves * x.wave_bytes > large) --waves; }
    x.fft_waves = waves;
    int work = waves * x.wave_bytes;
    const int scan_bytes = 2 * (kScanChunk + 2 * d + 8) * (int)sizeof(float);
    if (fixed + scan_bytes <= std::max(fixed + work, small)) { x.scan_in_lds = 1; work = std::max(work, scan_bytes); }
    const int lds = fixed + work;
    auto go = [&](auto tag) -> hipError_t {
        constexpr int V = decltype(tag)::value;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&local_maxima_exact_kernel<V>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(local_maxima_exact_kernel<V>, dim3((unsigned)grid), dim3(kExactThreads), lds, s, x);
        return hipGetLastError();
    };
    if (plan.variant == kFftLds4) return go(std::integral_constant<int, kFftLds4>{});
    return go(std::integral_constant<int, kFftLdsAcc>{});
}

hipError_t launch_unit_rows_f64(const ExactSource& src, const PeakRefine* refine, hipStream_t s) {
    if (!refine || !refine->frame_list || !refine->stats) return hipSuccess;
    UnitRowsArgs x{};
    x.src = src; x.frame_list = refine->frame_list; x.stats = refine->stats; x.gen = refine->gen;
    int logM = 0;
    while ((2 << logM) < src.W) ++logM;
    x.logM = logM;
    const int Mh = src.W / 2;
    // windows up to 2048: half a twiddle table + two transforms (channel pairs) = 40 KB, four workgroups per CU; else twiddles + one
    const int lds = Mh <= 1024 ? (Mh / 2 + 2 * Mh) * (int)sizeof(double2) : 2 * Mh * (int)sizeof(double2);
    auto go_wg = [&](auto tag) -> hipError_t {
        constexpr int KQ = decltype(tag)::value;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&unit_rows_f64_wg_kernel<KQ>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(unit_rows_f64_wg_kernel<KQ>, dim3(1024), dim3(256), lds, s, x);
        return hipGetLastError();
    };
    return Mh <= 1024 ? go_wg(std::integral_constant<int, 1>{}) : go_wg(std::integral_constant<int, 4>{});
}

}  // namespace repet
