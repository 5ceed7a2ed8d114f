// Shared declarations for the gfx950 REPET engine (kernels + host orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/repet_hip.h"
#include <vector>

namespace repet {

constexpr int kWave = 64;          // CDNA4 wavefront
constexpr int kFreqAlign = 32;     // spectrogram rows are padded to a multiple of 32 bins (GEMM K-step)
constexpr int kTile = 128;         // Gram tile edge; frame-major matrices are padded to whole tiles
constexpr float kMaskEps = 2.220446049250313e-16f;  // np.finfo(float).eps, repet.py:1446

__host__ __device__ inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) for `kernel` on the current device, grow-only and under a lock: the
// attribute is process state, and setting it per launch from several host threads races with their launches.
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// ---- kernel launchers (implemented in the .hip files; all asynchronous on `s`) -------------------

// K1: frame + window + real FFT + magnitude (+ channel mean, unit-norm rows).
//   audio[n][C] fp32 interleaved -> X[c][t][FS] (re,im), V[c][t][FS], Vm[t][FS] (mean magnitude),
//   Vn[t][FS] (Vm / ||Vm||, the cosine-similarity operand), P[t][FS] (Vm^2, the beat-spectrum operand).
//   Any of Vm/Vn/P may be null. Pad bins [F,FS) are written as zero. centred: repet.py:1031 vs :781.
struct StftArgs {
    const float* audio; int64_t n_samples; int32_t n_channels;
    int64_t sample_offset;   // first sample of the clip inside `audio` (segments of `extended`)
    const float* window; const float2* twiddle;   // twiddle[m] = exp(-2 pi i m / W), m < W
    int32_t W, H; int64_t T; int32_t FS; int32_t centred;
    float2* X; float* V; int64_t chan_stride;     // elements between channels in X and V
    float* Vm; float* Vn; float* P;
    // (nullable) the f16 hi / lo planes of Vn for the f16-split Gram kernels (gram_f16.hip: [row][FS/32][hi 32 | lo 32],
    // scale 2^7), written beside Vn so that no separate split pass reads Vn again; clips batch_mean_stride * 2 halves apart
    void* Vh;
    // (nullable, stft_reg.hip only) the f16 planes of P = Vm^2 scaled ROW BY ROW for the banded Gram of the beat spectrum
    // (what split_f16_rows_kernel makes of P) and the inverse scale of every row: the wave that owns a frame has the whole
    // row in registers, so neither the fp32 P nor a second pass over it is needed
    void* Ph; float* Ph_inv; int64_t batch_inv_stride;
    // batch of equal-length clips (segments of `extended`): blockIdx.y = b
    int32_t n_batch; int64_t batch_sample_stride; // samples between the starts of consecutive clips
    int64_t batch_spec_stride;                    // elements between clips in X and V (= C*chan_stride)
    int64_t batch_mean_stride;                    // elements between clips in Vm/Vn/P (= Tpad*FS)
};
hipError_t launch_stft(const StftArgs& a, hipStream_t s);
// strict reference mode: X and |X| = NaN in every bin of the frames that hold an infinite sample (stft.hip)
hipError_t launch_infinite_frames_fix(const StftArgs& a, hipStream_t s);

// Cross-fade weight of sample n of a segment of `extended` (repet.py:380-414). The reference fades IN PLACE: every
// segment after the first multiplies what has been accumulated under its first `overlap` samples by the falling
// half of triang(2*overlap), multiplies its own first `overlap` samples by the rising half, and adds. A sample of
// segment j is therefore scaled by its own rise (n < fade_in) and by the fall of EVERY later segment q = 1..later
// (starting q*step samples further on) whose overlap zone covers it -- one factor with the default 50 % overlap,
// several when the step is shorter than the overlap.
__host__ __device__ inline float segment_weight(int64_t n, int64_t fade_in, int64_t overlap, int64_t step, int later) {
    float w = 1.f;
    // (numerator times the rounded reciprocal of the denominator, as segment_weight32 below: the two agree to the bit)
    if (n < fade_in) w = (float)(2 * n + 1) * (1.0f / (float)(2 * fade_in));
    if (overlap > 0 && step > 0) {
        const float inv_ov = 1.0f / (float)(2 * overlap);
        for (int64_t q = 1; q <= later && q * step <= n; ++q) {
            const int64_t r = n - q * step;       // position inside later segment q
            if (r < overlap) w *= (float)(2 * (overlap - r) - 1) * inv_ov;
        }
    }
    return w;
}

// a * b rounded to fp32 and never contracted into a following add (HIP's __fmul_rn is a plain product): the inverse STFT
// kernels multiply the spectrum by the mask plane with it, so that the products equal the ones a mask kernel stores
__device__ __forceinline__ float mul_rounded(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
// |x| of a spectrum bin, with the roundings pinned (x.y^2 rounded, one fma, then the hardware square root v_sqrt_f32, 1 ulp):
// the forward kernels store it as V and the inverse STFT of `original` / `extended` recomputes it from X instead of
// reading V -- the two must agree to the bit, which one instruction does by construction. sqrtf() is the correctly rounded
// one: twenty instructions per bin (two trial roundings with compare + select, denormal scaling, class fix-up) in kernels
// that are VALU-bound -- 340 of the 1 700 instructions of a forward transform. REPET_IEEE_SQRT (build flag): sqrtf().
#ifdef REPET_IEEE_SQRT
__device__ __forceinline__ float magnitude(float2 x) { return sqrtf(fmaf(x.x, x.x, mul_rounded(x.y, x.y))); }
#else
// v_sqrt_f32 takes no denormal input and the squares of a spectrum below 1e-19 underflow: such a bin (a frame that is silent
// but for noise at the bottom of the fp32 range) goes the long way round, scaled by 2^64 -- a compare and a branch that is
// practically never taken. (A purely real bin -- DC, Nyquist -- is sqrt(x * x), within
// one ulp of the |x| np.abs returns; the forward and the inverse kernels share this function, so they agree to the bit.)
__device__ __forceinline__ float magnitude(float2 x) {
    const float s = fmaf(x.x, x.x, mul_rounded(x.y, x.y));
    float r = __builtin_amdgcn_sqrtf(s);
    if (__builtin_expect(s < 1.0e-30f, 0)) {                    // (an exact zero takes this way too, and comes out as zero)
        const float a = x.x * 18446744073709551616.0f, b = x.y * 18446744073709551616.0f;
        r = sqrtf(fmaf(a, a, b * b)) * 5.421010862427522e-20f;
    }
    return r;
}
#endif

// w(n) of `extended` (segment_weight, common.h) for positions inside ONE segment, in 32-bit arithmetic; inv_in / inv_ov
// are 1.0f / (float)(2 fade_in) and 1.0f / (float)(2 overlap), computed once per kernel (a division per sample was 64
// correctly rounded divisions per hop and lane in the register inverse STFT: a third of its instructions on `extended`).
// Same values as segment_weight: the conversions are of the same integers, the reciprocals of the same floats.
__device__ __forceinline__ float segment_weight32(int n, int fade_in, int overlap, int step, int later, float inv_in, float inv_ov) {
    float w = 1.f;
    if (n < fade_in) w = (float)(2 * n + 1) * inv_in;
    if (overlap > 0 && step > 0) {
        for (int q = 1, base = step; q <= later && base <= n; ++q, base += step) {
            const int r = n - base;
            if (r < overlap) w *= (float)(2 * (overlap - r) - 1) * inv_ov;
        }
    }
    return w;
}

// The power of two that brings a row's largest magnitude m to [2^13, 2^14) (gram_f16.hip: rows of a general matrix are
// scaled one by one for the f16 split; exact to apply and to remove).
__device__ __forceinline__ float f16_row_scale(float m) {
    if (!(m > 0.f) || !(m < INFINITY)) return 1.f;
    int e;
    (void)frexpf(m, &e);                       // m = f * 2^e, 0.5 <= f < 1
    return ldexpf(1.f, 14 - e);
}
// one component x of such a row into the two f16 planes (the arithmetic of split4, gram_f16.hip)
__device__ __forceinline__ void store_split_f16_scaled_at(_Float16* p, float x, float sc) {      // p: the component's hi half
    const float v = x * sc;
    const _Float16 h = (_Float16)v;
    p[0] = h;
    p[32] = (_Float16)(v - (float)h);
}
__device__ __forceinline__ void store_split_f16_scaled(void* planes, int64_t e, float x, float sc) {
    store_split_f16_scaled_at(static_cast<_Float16*>(planes) + ((e >> 5) << 6) + (e & 31), x, sc);
}

// one component of a unit row into the two f16 planes (the arithmetic of split_f16_kernel, gram_f16.hip)
__device__ __forceinline__ void store_split_f16_at(_Float16* p, float x) {                          // p: the component's hi half
    const float v = x * 128.0f;
    const _Float16 h = (_Float16)v;
    p[0] = h;
    p[32] = (_Float16)(v - (float)h);
}
__device__ __forceinline__ void store_split_f16(void* planes, int64_t e, float x) {
    store_split_f16_at(static_cast<_Float16*>(planes) + ((e >> 5) << 6) + (e & 31), x);
}

// K9: masked spectrum -> inverse real FFT -> time frames yf[c][t][W] (scaled by 1/W like np.fft.ifft).
struct IstftArgs {
    const float2* Y; int64_t chan_stride; int32_t n_channels; int64_t T; int32_t FS; int32_t W;
    const float2* twiddle; float* frames;  // frames[c][t][W]
};
hipError_t launch_istft_frames(const IstftArgs& a, hipStream_t s);

// K9 fused (hop = W/2 only): masked spectrum -> inverse FFT -> overlap-add -> out[n][C], no frames buffer.
struct IstftOlaArgs {
    const float2* Y; int64_t chan_stride; int32_t n_channels; int64_t T; int32_t FS; int32_t W;
    const float2* twiddle; int64_t trim; float* out; int64_t n_out; int64_t out_offset; float scale;
    int32_t accumulate_weighted; int64_t fade_in, fade_out;
    int64_t seg_step; int32_t later;   // cross-fade of `extended`, filled by the kernels from the batch fields
    int64_t first_hop, last_hop;   // filled by the launcher
    // batch of segments (extended): blockIdx.y = slot, segment j = batch_first + slot*batch_step reads its
    // spectra at j_local*batch_spec_stride, writes at out_offset + j*batch_out_stride, and fades in unless
    // j == 0 / out unless j == batch_total-1 over `overlap` samples (fade_in/fade_out are ignored then)
    int32_t n_batch, batch_first, batch_step, batch_total, batch_local0; int64_t batch_spec_stride, batch_out_stride, overlap;
    // (nullable) the soft mask as a plane of its own, laid out like V (element strides of Y): the spectrum is multiplied
    // by it as it is fetched, so the mask kernels write 4 bytes per cell instead of reading and rewriting 8 + 8
    const float* M;
    // (nullable, original / extended on the register kernels) instead of M: the
    // repeating-segment model of every clip of the batch, model[clip][channel][q < period][FS], and the clips' periods
    // (device array, batch-local index): the mask of frame t is soft_mask(|Y|, model[t mod period]) -- what mask_period_kernel
    // would have written into M, computed where it is used (|Y| by magnitude(), the forward kernels' own V), so that kernel
    // only writes the model (a third of a plane or less), never reads V a second time, and the inverse reads no plane at all
    const float* model; const int32_t* periods; int64_t model_batch_stride, model_chan_stride; int32_t cutoff;
    // channel groups (launch_istft_ola splits a clip with more channels than one workgroup's LDS holds): this launch writes
    // channels [out_chan0, out_chan0 + n_channels) of an output interleaved over out_channels (0: n_channels, from 0)
    int32_t out_channels, out_chan0;
};
hipError_t launch_istft_ola(const IstftOlaArgs& a, hipStream_t s);
// register-resident variants for W = 2048 (stft_reg.hip); launch_stft / launch_istft_ola pick them themselves
bool reg_fft_supported(int W, int n_channels, bool inverse);
hipError_t launch_stft_reg(const StftArgs& a, hipStream_t s);
hipError_t launch_istft_ola_reg(const IstftOlaArgs& a, int64_t hops, hipStream_t s);
// true when launch_istft_ola will take the register kernel for these arguments (the only one that applies a model itself)
bool istft_reg_takes(const IstftOlaArgs& a);

// Overlap-add of frames[c][t][W] at hop H into out[n][C] (interleaved), out sample n takes padded
// position n + trim; multiplied by `scale` (1/sum(window[0:W:H])).
struct OlaArgs {
    const float* frames; int32_t n_channels; int64_t T; int32_t W, H; int64_t trim;
    float* out; int64_t n_out; int64_t out_offset; float scale; int32_t accumulate_weighted;
    // accumulate_weighted (extended, repet.py:380-414): out += w(n) * y instead of out = y, with the
    // triangular fade-in over [0,fade_in) and fade-out over [n_out-fade_out, n_out)
    int64_t fade_in, fade_out;
};
hipError_t launch_overlap_add(const OlaArgs& a, hipStream_t s);

// Tile list for the Gram kernels: upper-triangle 128x128 tiles with bj - bi < ndiag (ndiag = nb: all),
// ordered for XCD locality; returns the padded list length (a multiple of 8).
int gram_tile_list(int nb, int ndiag, std::vector<int2>* out);
inline int gram_band_diagonals(int n_lags) { return (n_lags + 126) / kTile + 1; }
// K3: S[T][TS] = A A^T for A[Tpad][FS] fp32 (rows >= T are zero), MFMA fp32, upper tiles mirrored.
hipError_t launch_gram_full(const float* A, int64_t T, int32_t FS, float* S, int64_t TS, const int2* tiles,
                            int32_t n_tiles, hipStream_t s);
// K3h (gram_f16.hip): the same S from the f16 matrix cores: the fp32 rows are split once into f16 halves hi, lo
// (`planes`: 2 * count halves, interleaved per 32 components) and S = hi hi' + hi lo' + lo hi': fp32-class accuracy
// at 3/16 of the fp32 MFMA time.
// launch_split_f16: unit rows (fixed scale 2^7). launch_split_f16_rows: a general matrix [n_rows][FS], every row scaled
// by its own power of two (row_inv[row] = 1 / scale, to be passed to the band launcher).
hipError_t launch_split_f16(const float* src, void* planes, int64_t count, hipStream_t s);
hipError_t launch_split_f16_rows(const float* src, void* planes, int64_t n_rows, int32_t FS, float* row_inv, hipStream_t s);
hipError_t launch_gram_full_f16(const void* planes, int64_t T, int32_t FS, float* S, int64_t TS,
                                const int2* tiles, int32_t n_tiles, hipStream_t s);
// the same full matrix with 256 x 256 tiles and LDS-DMA staging (gram_f16_big.hip); tiles from gram_tile_list(ceil(T / 256), all);
// `planes` must be readable up to row round_up(T, 256)
int gram_big_tile();
// seg (nullable): the epilogue also writes the segment records of S's rows (peaks.h: PeakArgs::seg), pitch seg_pitch
hipError_t launch_gram_full_f16_big(const void* planes, int64_t T, int32_t FS, float* S, int64_t TS,
                                    const int2* tiles, int32_t n_tiles, hipStream_t s, float* seg = nullptr, int32_t seg_pitch = 0);
// banded form: band[t][l] = row t . row t+l; plane_batch_stride in halves (2 * Tpad * FS); row_inv (nullable): per-row
// inverse scales of launch_split_f16_rows, inv_batch_stride floats between clips; lookback: band[j][l] = row j . row j-l
// instead (the layout the peak picking of simonline reads row by row: PeakArgs::mode 2)
hipError_t launch_gram_band_f16(const void* planes, int64_t T, int32_t FS, float* band, int32_t n_lags, int32_t LP,
                                const int2* tiles, int32_t n_tiles, int32_t n_batch, int64_t plane_batch_stride,
                                int64_t band_batch_stride, hipStream_t s, const float* row_inv = nullptr,
                                int64_t inv_batch_stride = 0, bool lookback = false);
// K6/K3b: band[t][l] = A[t] . A[t+l] for 0 <= l < n_lags (band pitch LP), zero where t+l >= T.
hipError_t launch_gram_band(const float* A, int64_t T, int32_t FS, float* band, int32_t n_lags, int32_t LP,
                            const int2* tiles, int32_t n_tiles, int32_t n_batch, int64_t a_batch_stride,
                            int64_t band_batch_stride, hipStream_t s);

// Stage exports only: out = A B^T (cosine similarity of two column sets) and the per-column autocorrelation.
hipError_t launch_matmul_nt(const float* A, int64_t TA, const float* B, int64_t TB, int32_t FS, float* out,
                            int64_t pitch, hipStream_t s);
hipError_t launch_acorr(const float* x, int32_t R, int32_t n_cols, int32_t pitch, float* ac, hipStream_t s);

// Windowed diagonal sums of the band: beat[w][l] = sum_{t=lo_w}^{hi_w - l} band[t][l] / ((len - l) * F)
//   window w covers frames [start0 + w*step, start0 + w*step + len) clipped to [0,T).
//   partial: scratch of n_batch * n_windows * band_window_chunks(T, len) * LP floats (chunk sums, fixed chunk size)
int band_window_chunks(int64_t T, int64_t len);
hipError_t launch_band_window_sum(const float* band, int64_t T, int32_t LP, int32_t n_lags, int32_t n_freq,
                                  int64_t start0, int64_t step, int64_t len, int32_t n_windows,
                                  float* beat, int32_t beat_pitch, int32_t n_batch, int64_t band_batch_stride,
                                  int64_t beat_batch_stride, float* partial, hipStream_t s);

// K7: period[c] = argmax(beat[c][lo:hi]) + 1 + lo (first max wins), hi = min(period_hi, n_lags/3).
hipError_t launch_periods(const float* beat, int32_t n_cols, int32_t pitch, int32_t n_lags, int32_t lo,
                          int32_t hi, int32_t* period, hipStream_t s);
// adaptive: expand per-window periods to per-frame periods with the reference's replicate-with-a-hole
// rule (repet.py:1194-1204): frame i+step-1 of every window keeps the all-zero column => lo+1.
hipError_t launch_expand_periods(const int32_t* win_period, int32_t n_windows, int32_t step, int64_t T,
                                 int32_t lo, int32_t* frame_period, hipStream_t s);

// K4: strict local maxima (+-d, clipped) of every row of M[n_rows][pitch], top `number` by value.
//   mode 0: row r is M[r][0..n_cols); indices are column numbers.
//   mode 1 (simonline): "row" j is the circular-buffer view of the band: element c is
//           band[j-l][l], l = (j-c) mod B, for c < B = n_cols; indices are FRAME numbers j-l.
//           `shift` (streaming): band row of frame f is f - shift, and the indices written are band rows.
//   refine (nullable): near-tie refinement. M must then hold cosine similarities of the fp32 unit rows
//           `unit_rows` (row of frame f at (f - shift) * pitch): every decision within `delta` of a tie is
//           re-taken from float64 similarities of those rows, so the index lists do not depend on the fp32
//           rounding of the Gram kernel. stats (nullable, 4 counters, added to): rows refined, near-tied
//           elements, decisions changed, flat rows left to fp32.
//           Second level (peaks_exact.hip; redo_list != nullptr): the float64 values above still carry the rounding of
//           the fp32 spectra. Rows with such a comparison closer than delta2 (and flat rows) are appended to
//           redo_list[2 k] = row, [2 k + 1] = clip (capacity rows x clips; count in stats[4]); redo_flag[clip * flag_stride
//           + row] == gen marks a listed row. launch_local_maxima_exact then decides those rows from float64 spectra.
constexpr int kRefineStats = 32;   // counters behind PeakRefine::stats: [0] rows refined, [1] near-tied elements, [2] decisions
                                   // changed, [3] flat rows, [4] rows handed to the second level, [5] its work cursor,
                                   // [6] elements given float64 spectra, [7] rows whose list the second level changed,
                                   // [8] largest |level 1 - level 2| in units of 1e-12, [9] float64 unit rows computed,
                                   // [10] frames queued for float64 spectra, [11] the queue's cursor, [12] rows with a record
                                   // (fast path), [13] their cursor, [14] rows the fast path handed on to the general one
// The counters that only COUNT (diagnostics: [0]-[3], [6]-[9], [14]) are kept in kStatShards copies, each on a cache line of
// its own, and added up when they are read (stat_total): thousands of wavefronts adding to ONE word inside a few
// microseconds are served at about 88 per microsecond, which a short kernel then waits for. The counters the device
// itself reads (list lengths and cursors: [4], [5], [10]-[13]) live in copy 0 only.
constexpr int kStatShards = 64;
constexpr int kStatWords = kRefineStats * (kStatShards + 1);
struct PeakRefine {
    const float* unit_rows; int32_t pitch; float delta; double min_value; unsigned int* stats;
    double delta2; int32_t* redo_list; unsigned int* redo_flag; unsigned int gen; int64_t flag_stride;
    // fast path of the wavefront kernel (peaks_wave.hip; nullable): records of the rows' lists, their list and flags, the
    // queue of frames whose float64 unit rows are needed (frame = clip * frame_clip_stride + frame row)
    unsigned char* records; int32_t record_bytes; int32_t* lite_list; unsigned int* lite_flag;
    int32_t* frame_list; unsigned int* frame_flag; int64_t frame_clip_stride;
    // (nullable) float64 norms of the fp32 unit rows, one per row of unit_rows (launch_unit_row_norms): with them the first
    // pass's float64 similarities are one dot product per item instead of two and a square root
    const double* unit_norms;
};
// float64 L2 norm of every fp32 row: norms[r] = sqrt(sum_k rows[r * pitch + k]^2), n_rows rows of `pitch` floats (pitch % 4 == 0)
hipError_t launch_unit_row_norms(const float* rows, int64_t n_rows, int32_t pitch, double* norms, hipStream_t s);
// batch (nullable): blockIdx.y = clip of a batch of equal-shape matrices; element strides between the clips
struct PeakBatch { int32_t n_batch; int64_t m_stride, idx_stride, cnt_stride, unit_stride; };
hipError_t launch_local_maxima(const float* M, int64_t n_rows, int64_t row0, int32_t n_cols, int64_t pitch,
                               int32_t mode, float min_value, int32_t d, int32_t number, int32_t* idx,
                               int32_t idx_pitch, int32_t* count, hipStream_t s, int64_t shift = 0,
                               const PeakRefine* refine = nullptr, const PeakBatch* batch = nullptr,
                               void* scratch = nullptr, const struct ExactSource* lite_src = nullptr,
                               const float* seg = nullptr, int32_t seg_pitch = 0);
//   seg (mode 0, nullable): the rows' segment records (peaks.h: launch_segment_maxima, or the Gram kernel's epilogue), row r
//           of M at seg + (row0 + r) * 3 * seg_pitch: the wavefront kernel then takes its candidates from them.
//   lite_src (second level, fast path): instead of the first pass, the rows it left records of (PeakRefine::records)
//           are taken up again with the float64 unit rows of lite_src (same matrix arguments as the first call).
bool local_maxima_wave_supported(int n_cols, int d, int* record_bytes);
// Segment records (peaks.h: PeakArgs::seg): for every row and every aligned run of kSegWidth columns the largest value, the
// largest of the run's other elements and the position of the largest, as three planes of seg_pitch entries per row.
// launch_segment_maxima computes them from a matrix (the Gram kernel of the long clips writes them in its epilogue);
// segment_pitch: the planes' pitch for rows of n_cols elements; local_maxima_segments_apply: whether the peak picking will
// use them for this shape (mode 0, one clip, 16-byte aligned rows, windows of at least kSegWidth - 1 on either side)
constexpr int kSegWidth = 32;
hipError_t launch_segment_maxima(const float* M, int64_t n_rows, int n_cols, int64_t pitch, float* seg, int seg_pitch, hipStream_t s);
int segment_pitch(int n_cols);
bool local_maxima_segments_apply(int n_cols, int d, int64_t pitch, int mode, int n_batch);
hipError_t launch_unit_rows_f64(const struct ExactSource& src, const PeakRefine* refine, hipStream_t s);
//   scratch (nullable): local_maxima_scratch_bytes(n_rows, n_cols, d) bytes (0 for rows that fit one workgroup). With
//           it, rows of any length are handled in segments; without it the limit is about 40 000 elements per row.
size_t local_maxima_scratch_bytes(int64_t n_rows, int32_t n_cols, int32_t d);

// K4, second level (peaks_exact.hip): where the float64 spectra of a frame row come from. Frame row fr of clip b covers the
// samples frame_sample0 + fr * H .. + W - 1 of hi (+ lo, nullable: the fp32 remainder of a float64 upload), zero outside
// [0, n_samples). u64[b * u64_clip_stride + fr * FS ..] caches the unit rows, valid where u64_gen[b * gen_clip_stride + fr]
// equals the launch's generation (PeakRefine::gen).
struct ExactSource {
    const float* hi; const float* lo; int64_t n_samples; int32_t n_channels; int64_t clip_stride;
    int64_t frame_sample0; int32_t W, H, F, FS;
    const double* window64; const double2* twiddle64;      // twiddle64[m] = exp(-2 pi i m / W), m <= W
    double* u64; int64_t u64_clip_stride; unsigned int* u64_gen; int64_t gen_clip_stride;
};
// Decides the rows listed by the first pass again (same matrix arguments as launch_local_maxima; fixed grid, the count is
// read on the device). scratch: local_maxima_exact_scratch_bytes(n_cols) bytes.
size_t local_maxima_exact_scratch_bytes(int32_t n_cols, int* grid_out = nullptr);
hipError_t launch_local_maxima_exact(const float* M, int64_t row0, int32_t n_cols, int64_t pitch, int32_t mode, float min_value,
                                     int32_t d, int32_t number, int32_t* idx, int32_t idx_pitch, int32_t* count, hipStream_t s,
                                     int64_t shift, const PeakRefine* refine, const PeakBatch* batch, const ExactSource& src,
                                     void* scratch);

// K5/K8/K8b: gather-median masks. V[c][t][FS] -> (optional) mask[c][t][FS]; if X != null it is
// multiplied in place by the mask after the high-pass override mask[1..cutoff] = 1 (repet.py:185).
// The soft mask of one bin (repet.py:1446 and the high-pass rule :1449-1451): shared by the mask kernels and by the inverse
// STFT that applies a repeating-segment model itself, so both give the same bits.
__device__ __forceinline__ float soft_mask(float v, float model, int f, int cutoff) {
#ifdef REPET_IEEE_SQRT
    const float m = (fminf(v, model) + kMaskEps) / (v + kMaskEps);
#else
    // the quotient by the hardware reciprocal (1 ulp) and one product; where the model is not below the magnitude the
    // quotient is x / x = 1 exactly in the reference, so it is 1 here too (the reciprocal alone could say 1 - 2^-24)
    const float q = fminf((model + kMaskEps) * __builtin_amdgcn_rcpf(v + kMaskEps), 1.0f);      // (never 1 + an ulp)
    const float m = (model >= v) ? 1.0f : q;
#endif
    // fminf drops a NaN model; np.minimum propagates it (empty similarity list -> NaN frame)
    const float mm = (model != model) ? model : m;
    return (f >= 1 && f <= cutoff) ? 1.0f : mm;
}

struct MaskArgs {
    const float* V; int64_t chan_stride; int32_t n_channels; int64_t T; int32_t F, FS;
    float2* X; float* mask; int32_t cutoff;
    float* model; int64_t model_batch_stride, model_chan_stride;   // mask_period only (nullable): write the medians [clip][channel][q][FS] and nothing else
    int64_t pad_row;   // rows pad_row / pad_row+1 of every channel of V hold -1.0f / +inf (median pads)
    int32_t n_batch; int64_t batch_stride;   // mask_period / mask_sim: blockIdx.z = clip, elements between clips in V and X
    int64_t idx_batch_stride, cnt_batch_stride;   // mask_sim: elements between the clips' index lists / list lengths
    int64_t frame0;                          // mask_sim only: first frame row handled by this launch (streaming window)
    int64_t frame_end;                       // mask_sim only: one past the last frame row of this launch (0 = T)
    // mask_sim, rank-domain median (rank.hip): 16-bit rank codes R[c][row][FS] (same row geometry as V, pad rows 0 /
    // 0x7C00 at pad_row, pad_row+1) and the rank -> value table Vs[c][f][vs_pitch] of the first n_rank_cols bins of
    // every channel. R == nullptr: select on the floats themselves.
    const unsigned short* R; int64_t r_chan_stride; const float* Vs; int64_t vs_pitch; int32_t n_rank_cols;
    // the same codes (minus their base) bit-sliced, for the bit-sliced selection (mask_bits.hip): P[t][plane][64] words,
    // bit b of word l = bit `plane` of the code of cell 64 b + l, cells numbered channel-major over the ranked bins.
    // P == nullptr: the packed network on R.
    const unsigned* P; int32_t n_planes;
    unsigned* median_codes;       // bit-sliced selection: one word per cell, V's geometry (upper code << 16 | lower code | flag)
};
constexpr int kPadRows = 8;       // rows kept behind the Tpad frame rows of V (2 used)
constexpr int kMinIdxPitch = 128; // index lists are readable up to the largest network size
hipError_t launch_fill_pad_rows(float* V, int64_t chan_stride, int32_t n_channels, int64_t pad_row, int32_t FS,
                                hipStream_t s, float* Z = nullptr, int64_t z_stride = 0, int64_t z_count = 0, int32_t n_z = 0,
                                unsigned int* stats = nullptr, float* Z2 = nullptr);
// max_count / min_period bound the list length so the launcher can pick the smallest compiled network.
// side/fork/join (nullable): second stream and two events to run the Nyquist-bin kernel beside the main one.
// parts: 1 = main kernel only, 2 = Nyquist-bin kernel only, 3 = both (chunked pipelines launch them separately).
hipError_t launch_mask_sim(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count,
                           int64_t first_frame, int32_t max_count, hipStream_t s, hipStream_t side = nullptr,
                           hipEvent_t fork = nullptr, hipEvent_t join = nullptr, int parts = 3, bool lookups_by_caller = false);
int median_network_instructions(int max_n, int* net_size);
// bit-sliced selection (mask_bits.hip): lists of at most 128 entries, at most 32 blocks of 64 ranked bins over all channels
int code_planes_for(int64_t T);                       // planes of the codes of a T-frame clip (bits of T - 1, at least 11)
bool mask_sim_bits_supported(int64_t T, int32_t n_channels, int32_t n_cols, int32_t max_count);
int mask_sim_bits_instructions(int32_t max_count, int32_t n_planes);
// the selection (codes of both medians per cell into m.median_codes), and the lookups + mask from those codes (the caller
// launches the second behind the first; launch_mask_sim does both unless told that the caller will)
hipError_t launch_mask_sim_bits(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count, int32_t max_count,
                                unsigned n_launch, hipStream_t s);
hipError_t launch_mask_from_codes(const MaskArgs& m, const int32_t* count, hipStream_t s);
hipError_t launch_mask_adaptive(const MaskArgs& m, const int32_t* periods, int32_t order, hipStream_t s);
hipError_t launch_mask_period(const MaskArgs& m, const int32_t* period_dev, int32_t period_host,
                              int32_t min_period, hipStream_t s);

// Rank transform of V for the rank-domain median of `sim` (rank.hip): every column (bin f of channel c over the T
// frames) is sorted once; R gets 0x0400 + the number of strictly smaller magnitudes of the column, Vs the sorted column.
constexpr int kRankCodeBase = 0x0400;     // codes are positive normal f16 bit patterns (u16 order == f16 order)
constexpr int kRankMinFrames = 1024;      // shorter clips: the selection on floats is cheap enough
constexpr int kRankMaxFrames = 30720;     // 0x0400 + T - 1 must stay below 0x7C00 (+inf, the high pad)
struct RankArgs {
    const float* V; int64_t chan_stride; int32_t n_channels; int64_t T; int32_t FS;
    int32_t n_cols;                       // bins [0, n_cols) of every channel are ranked (a multiple of 128)
    unsigned short* R; int64_t r_chan_stride;
    float* Vs; int64_t vs_pitch;          // Vs[c * n_cols + f][vs_pitch], vs_pitch = round_up(T, 32)
    unsigned short* codes;                // scratch: the codes column-major, [c * n_cols + f][vs_pitch]
    unsigned* P; int32_t n_planes;        // the codes bit-sliced INSTEAD of R (exactly one of R and P is set): MaskArgs::P
    // phases of the chain a launch_rank_columns call runs: bit 0 the transpose V -> columns, bit 1 the sort and what follows
    // (0 = both). The transpose needs V only and may run early (exec_sim, REPET_RANK_TRANSPOSE=early).
    int32_t phases;
};
bool rank_columns_supported(int64_t T);
// CPUs of the NUMA node device `dev` hangs off that this process may run on (hostio.hip; empty: unknown or nothing to choose)
std::vector<int> host_cpus_near_device(int dev);
// hook(user, step) is called behind every kernel of the chain (0 transpose, 1 sort + rank search, 2 code planes, 3 transpose back):
// the engine records a timing mark there when the chain runs alone on the main stream (REPET_RANK_OVERLAP=0)
typedef void (*RankStepHook)(void* user, int step);
hipError_t launch_rank_columns(const RankArgs& a, hipStream_t s, RankStepHook hook = nullptr, void* user = nullptr);
hipError_t launch_fill_rank_pad_rows(unsigned short* R, int64_t r_chan_stride, int32_t n_channels, int64_t pad_row,
                                     int32_t FS, hipStream_t s);

// Host <-> device staging (hostio.hip): a ring of pinned fp32 chunks per context; worker threads convert between the
// caller's array and the ring while earlier chunks are in flight.
struct StagingRing {
    static constexpr int kSlots = 6;
    static constexpr size_t kSlotElems = (size_t)1 << 20;      // 4 MB of fp32 per slot
    float* base = nullptr;
    float* base_lo = nullptr;      // second ring for the fp32 remainders of float64 uploads (ensure_lo)
    // deferred remainders (staged_upload with a second stream): the whole remainder plane is staged in pinned memory of its
    // own while the samples travel, and follows them on the other stream
    float* lo_full = nullptr; size_t lo_full_elems = 0; hipEvent_t hi_done = nullptr, lo_done = nullptr; bool lo_in_flight = false;
    hipEvent_t event[kSlots] = {};
    bool busy[kSlots] = {};
    hipError_t ensure();
    hipError_t ensure_lo();
    hipError_t ensure_lo_full(size_t elems);
    void release();
    ~StagingRing();
};
// dtype: 0 float32, 1 float64, 2 int16 (REPET_F32 / F64 / I16)
// s_lo (nullable): the remainders follow the samples on this stream, behind the last chunk of samples: what is enqueued on
// `s` after the call starts as soon as the SAMPLES are there, and whoever reads dst_lo waits for ring.lo_done first
hipError_t staged_upload(StagingRing& ring, const void* src, int dtype, float* dst, size_t count, hipStream_t s,
                         float* dst_lo = nullptr, bool* any_lo = nullptr, hipStream_t s_lo = nullptr, bool* not_finite = nullptr);
// not_finite (nullable): set when a sample of src is NaN or infinite
hipError_t staged_download(StagingRing& ring, const float* src, double* dst, size_t count, hipStream_t s);
hipError_t staged_upload_bytes(StagingRing& ring, const void* src, void* dst, size_t n_bytes, hipStream_t s);
// WAVE files (wav.hip): header parsing on the host, PCM decode + wavread's normalisation on the device
const char* wav_parse(const void* file, int64_t n_bytes, ::repet_wav_info* info);
int wav_itemsize(const ::repet_wav_info& w);
hipError_t launch_decode_pcm(const void* raw, int format, int width, float* dst, int64_t n, hipStream_t s);
int64_t wav_float_header(unsigned char* out, int sampling_frequency, int n_channels, int64_t n_samples, int item_bytes);
// pinned host buffers from a recycling pool (result arrays of the Python module); nullptr when the pool declines
void* host_alloc(size_t bytes);
void host_free(void* ptr);

// elementwise helpers
hipError_t launch_convert_in(const void* src, int dtype, float* dst, int64_t count, hipStream_t s, float* dst_lo = nullptr);
hipError_t launch_convert_out(const float* src, double* dst, int64_t count, hipStream_t s);
hipError_t launch_foreground(const float* audio, const float* background, double* dst, int64_t count, hipStream_t s);
hipError_t launch_foreground_f32(const float* audio, const float* background, float* dst, int64_t count, hipStream_t s);
hipError_t launch_channel_mean(const float* audio, const float* background, int which, int n_channels, float* dst,
                               int64_t n_samples, hipStream_t s);
hipError_t launch_square(const float* src, float* dst, int64_t count, hipStream_t s);
hipError_t launch_unit_rows(const float* src, float* dst, int64_t T, int32_t F, int32_t FS, hipStream_t s);

}  // namespace repet
