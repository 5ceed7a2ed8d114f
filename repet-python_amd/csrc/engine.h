// Shared by the translation units of the host orchestration (engine*.hip): the context, its workspaces, the error plumbing
// and the helpers one pipeline file calls in another.
//   engine.hip         contexts, tables, Gram / STFT wrappers, timing, uploads and downloads, the C ABI's core entry points
//   engine_period.hip  original, extended, adaptive (the period family)
//   engine_sim.hip     sim, simonline, the peak picking's refinement plumbing
//   engine_batch.hip   repet_run_batch over several devices, the RCCL transport
//   engine_stages.hip  stage-level exports and the accessors of a run's integer intermediates
//   engine_online.hip  the streaming handle
#pragma once
#include "../../include/repet_hip.h"
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

namespace repet_eng {
using namespace repet;

extern thread_local std::string g_last_error;
int fail(int code, const std::string& msg);



#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? REPET_ERR_OOM : REPET_ERR_HIP,                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                        \
        }                                                                                          \
    } while (0)

#define RP_TRY(expr)                                                                               \
    do {                                                                                           \
        int rc_ = (expr);                                                                          \
        if (rc_ != REPET_OK) return rc_;                                                           \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool borrowed = false;        // points into another context's buffer: never freed or grown here
    void borrow(void* ptr, size_t bytes) { p = ptr; cap = bytes; borrowed = true; }
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (borrowed) return hipErrorInvalidValue;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = (bytes + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p && !borrowed) (void)hipFree(p); p = nullptr; cap = 0; borrowed = false; }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct Tables {
    DevBuf window, twiddle;
    DevBuf window64, twiddle64;   // the same in float64 (second level of the peak picking, peaks_exact.hip): W and W + 1 entries
    double cola = 1.0;   // sum(window[0:W:H]) for H = W/2 (repet.py:1103)
};

}  // namespace repet_eng

using repet_eng::DevBuf;
using repet_eng::Tables;
using repet::StagingRing;

struct repet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // short independent kernels run beside the main stream
    hipStream_t copy_stream = nullptr;   // the remainder plane of a float64 upload follows the samples here (created on first use)
    std::vector<hipStream_t> ballast_streams;   // candidates that shared the main stream's hardware queue (pick_side_stream)
    hipEvent_t fork_event = nullptr, join_event = nullptr;
    hipEvent_t norms_fork = nullptr, norms_done = nullptr;      // exec_sim: the unit rows' float64 norms on the side stream beside the Gram kernel
    // resident clip
    DevBuf staging, audio, out, out64;
    StagingRing ring;             // pinned chunks the waveforms travel through (hostio.hip)
    int64_t n_samples = 0;        // per clip
    int32_t n_clips = 1;          // equal-shape clips back to back in `audio` / `out` (repet_ctx_upload_batch)
    int64_t clip_base = 0;        // first sample of the clip the single-clip pipelines currently work on
    // repet_ctx_set_window: the resident samples are [win_offset, win_offset + n_samples) of a clip of win_total samples
    // (multi-GPU `extended`: a rank holds only the samples of its own segment range); 0 = the resident clip is whole
    int64_t win_total = 0, win_offset = 0;
    bool win_skip_clear = false;  // exec_extended cleared `out` itself (window mode)
    DevBuf Mk;                    // the soft mask as a plane of its own (laid out like V), when the inverse STFT applies it
    DevBuf Wm;                    // original / extended: the repeating-segment models [clip][channel][q][FS] when the inverse STFT applies THEM
    bool mask_model = false;      // this pipeline's inverse STFT computes the mask from V and Wm (run_original)
    bool band_lookback = false;   // the last run_gram_band wrote band[j][l] = sim(j, j - l) (simonline on the f16-split kernel)
    bool mask_plane = false;      // the pipeline being enqueued keeps the mask apart instead of multiplying X in place
    bool ola_first_batch = false; // run_original: the first batch of equal segments of an `extended` run (class 0 may store)
    int32_t last_fs = 0;          // sampling frequency of the resident clip when it came from a WAVE file (for repet_ctx_result_wav)
    // `extended`: the longer last segment cannot join the batch of equal segments; its analysis (STFT .. mask) runs on
    // this auxiliary context's stream beside the batch and only its inverse STFT waits for the batch's
    repet_ctx* aux = nullptr;
    hipEvent_t aux_start = nullptr, aux_main_done = nullptr, aux_done = nullptr;
    std::function<int()> pre_synthesis;      // run_original calls it (once) right before its inverse STFT
    bool clip_loop = false;       // true while run_algo works through the clips one by one
    int32_t n_channels = 0;
    bool strict = true;           // samples that are not finite are let through as repet.py lets them (all five variants); false: REPET_FLAG_REFUSE_NONFINITE
    bool input_not_finite = false;   // the resident clip came from a host array that held such samples
    // the resident clip was never scanned (device planes -- the RCCL transport, torch tensors --, float WAVE payloads): under
    // `strict` the passes that reproduce the reference on such samples then run unconditionally (they cost microseconds)
    bool input_unscanned = false;
    bool nonfinite_passes() const { return input_not_finite || (strict && input_unscanned); }
    // workspaces
    DevBuf X, V, Vn, P, S, band, beat, idx, cnt, periods, win_periods, frames, tmp_a, tmp_b, tmp_c;
    DevBuf peak_scratch;          // per-segment candidates of long similarity rows (launch_local_maxima)
    DevBuf seg;                   // segment records of the similarity rows (PeakArgs::seg): [row][m1 | m2 | arg][seg_pitch]
    DevBuf beat_partial;          // chunk sums of the beat-spectrum windows (launch_band_window_sum)
    DevBuf amax;                  // inverse scale of every row of the matrix being split (scaled f16-split band Gram)
    DevBuf Vh;                    // f16 hi / lo halves of Vn for the split-precision Gram (gram_f16.hip)
    DevBuf refine_stats;          // kRefineStats counters of the last sim/simonline run (PeakRefine::stats)
    // second level of the peak picking (peaks_exact.hip): the fp32 remainders of a float64 upload (audio = hi, audio_lo = lo,
    // hi + lo = 48 bits of the caller's sample; empty when every remainder was zero or the input was not float64), the rows
    // handed over, the table of float64 unit rows with its generation stamps, the row workspaces of the fixed grid
    DevBuf audio_lo; bool has_lo = false;
    DevBuf redo_list, redo_flag, u64, u64_gen, exact_scratch;
    DevBuf unit_norms;            // float64 norms of the fp32 unit rows (PeakRefine::unit_norms)
    DevBuf lite_list, lite_flag, lite_records, frame_list, frame_flag;   // the wavefront kernel's fast path (peaks_wave.hip)
    unsigned int exact_gen = 0;
    bool refine_stats_cleared = false;   // ensure_spectra's housekeeping launch has zeroed them for the run being enqueued
    DevBuf R, Vs, rank_codes;     // rank codes of V, the sorted columns and the column-major codes (rank-domain median of `sim`, rank.hip)
    DevBuf code_planes, median_codes;   // the same codes bit-sliced, and the selected codes per cell (mask_bits.hip)
    // geometry for which the constant median-pad rows of R are in place (they survive every run of that geometry)
    const void* r_pads_ptr = nullptr; int64_t r_pads_stride = 0, r_pads_row = 0; int r_pads_channels = 0, r_pads_fs = 0;
    std::map<int, std::unique_ptr<Tables>> tables;
    DevBuf tiles;                 // Gram tile list of the last (nb, ndiag)
    int tiles_nb = -1, tiles_ndiag = -1, tiles_count = 0;
    DevBuf tiles_big;             // upper-triangle list of 256 x 256 tiles (gram_f16_big.hip)
    int tiles_big_nb = -1, tiles_big_count = 0;
    // last run
    int last_algo = -1;
    int64_t last_T = 0;
    int32_t last_n_periods = 0;
    int64_t last_idx_rows = 0;
    int32_t last_idx_pitch = 0;
    int32_t last_idx_number = 0;
    int32_t last_FS = 0; int64_t last_chan_stride = 0;     // sim: geometry of the last run's per-cell arrays
    int32_t last_median_path = 0;     // sim: 0 selection on the float magnitudes, 1 packed network on rank codes, 2 bit-sliced selection
    int32_t last_idx_batch = 1;   // clips whose lists sit back to back in idx / cnt (batch contexts)
    bool band_on_f16 = false;     // the last banded Gram ran on the f16-split kernel (stage label / roofline of bench.py)
    // timing
    std::vector<hipEvent_t> events;
    repet_timing* timing = nullptr;
    int n_marks = 0;
    // timing series (repet_ctx_timing_series_begin): every asynchronous run records its own block of events
    bool series_on = false; int series_cap = 0, series_steps = 0, series_marks = 0, event_base = 0;
    repet_timing series_timing{};
};

namespace repet_eng {

struct DeviceGuard {
    int prev = 0;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() { if (ok) (void)hipSetDevice(prev); }
};

struct Geo {
    int W, H, F, FS;
    int64_t T, Tpad, chan_stride;
    int C;
};

enum class MaskKind { period, adaptive, sim_float, sim_ranks };

struct MaskPlaneScope {            // the choice holds for one pipeline; stage exports and the streaming handle never see it
    repet_ctx* c;
    MaskPlaneScope(repet_ctx* ctx, bool on) : c(ctx) { c->mask_plane = on; }
    ~MaskPlaneScope() { c->mask_plane = false; }
};

struct ModelRef { const float* model; const int32_t* periods; int64_t batch_stride, chan_stride; int32_t cutoff; };

struct BatchInfo { int64_t transport = 0, clips_sent = 0, clips_with_remainders = 0, groups = 0; };
extern thread_local BatchInfo g_batch_info;

int logical_device_count(int physical);
int run_batch_impl(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                   const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices, int transport,
                   int one_device = -1);
void release_stream_contexts();
float peak_refine_delta(int FS, bool f16_gram);
double peak_exact_delta2();
int ensure_stamps(repet_ctx* c, DevBuf& buf, size_t count);
int make_refine(repet_ctx* c, const float* unit_rows, int FS, double threshold, PeakRefine* rf, int64_t rows = 0, int clips = 1,
                int n_cols = 0, int d = 0, int64_t frames = 0);
int run_exact_rows(repet_ctx* c, const Tables* tb, const Geo& g, const float* M, int64_t row0, int n_cols, int64_t pitch, int mode,
                   float min_value, int d, int number, int32_t* idx, int idx_pitch, int32_t* count, int64_t shift,
                   const PeakRefine& rf, const PeakBatch* batch, const float* hi, const float* lo, int64_t n_samples,
                   int64_t clip_stride, int64_t frame_sample0, int64_t n_frames, int clips);
bool rank_median_enabled();
int run_rank_columns(repet_ctx* c, const Geo& g, MaskArgs* m, hipStream_t stream, bool with_mark, int max_count, int phases = 0);
int exec_sim(repet_ctx* c, const repet_params* p);
int exec_simonline(repet_ctx* c, const repet_params* p);
int prepare_power_planes(repet_ctx* c, const Geo& g, int64_t T, int B);
IstftOlaArgs reg_probe(int W, int channels, bool weighted, int64_t n_out, int64_t out_stride, int64_t overlap);
int run_original(repet_ctx* c, const repet_params* p, int64_t offset, int64_t n, int B, int64_t hop,
                 int32_t* period_slots, bool weighted, int seg_first, int seg_total, int64_t overlap);
int exec_original(repet_ctx* c, const repet_params* p);
int64_t extended_segment_count(int64_t N, const repet_params* p);
int exec_extended(repet_ctx* c, const repet_params* p, int64_t first = 0, int64_t n_seg = -1);
int exec_extended_plan(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg, int64_t N);
int exec_adaptive(repet_ctx* c, const repet_params* p);
int check_params(const repet_params* p);
int h2d_pitched(repet_ctx* c, float* dst, int64_t dpitch, const float* src, int64_t rows, int64_t cols, int64_t rows_pad);
int d2h_pitched(repet_ctx* c, float* dst, const float* src, int64_t spitch, int64_t rows, int64_t cols);
int ctx_create(int device, repet_ctx** out, bool probe_side_stream);
int run_algo_one(repet_ctx* c, int algo, const repet_params* p);
int run_algo(repet_ctx* c, int algo, const repet_params* p);
int get_tables(repet_ctx* c, int W, Tables** out);
int upload_twiddle_only(repet_ctx* c, int W, const float2** tw);
int get_tiles(repet_ctx* c, int64_t T, int ndiag, const int2** tiles, int* count);
int run_band_window_sum(repet_ctx* c, const float* band, int64_t T, int LP, int n_lags, int n_freq, int64_t start0,
                        int64_t step, int64_t len, int n_windows, float* beat, int beat_pitch, int n_batch,
                        int64_t band_batch_stride, int64_t beat_batch_stride);
bool gram_f16_enabled();
bool gram_big_enabled();
int run_gram_full(repet_ctx* c, const float* A, int64_t T, int FS, float* S, int64_t TS, bool unit_rows = false,
                  bool planes_ready = false, float* seg = nullptr, int seg_pitch = 0, bool* seg_written = nullptr);
bool band_rows_on_f16(repet_ctx* c, int64_t T, int FS, int n_lags, int B, int64_t a_stride);
int run_gram_band(repet_ctx* c, const float* A, int64_t T, int FS, float* band, int n_lags, int LP, bool unit_rows = false,
                  int B = 1, int64_t a_stride = 0, int64_t band_stride = 0, bool planes_ready = false, bool lookback = false);
void mark(repet_ctx* c, const char* name, double bytes, double flops);
void begin_timing(repet_ctx* c, repet_timing* t);
void end_timing(repet_ctx* c);
Geo make_geo(int W, int H, int64_t T, int C);
bool split_in_stft(int B);
int mask_plane_forced();
bool mask_plane_wanted(MaskKind kind);
int ensure_spectra(repet_ctx* c, const Geo& g, bool want_vn, bool want_p, int B = 1, bool p_planes = false);
int run_stft(repet_ctx* c, const Geo& g, const Tables* tb, int64_t offset, int64_t n, int centred, bool vn, bool p,
             int B = 1, int64_t batch_sample_stride = 0, bool p_as_planes = false);
MaskArgs mask_args(repet_ctx* c, const Geo& g, int cutoff);
void apply_model(IstftOlaArgs& a, repet_ctx*, const ModelRef* mr);
int run_istft(repet_ctx* c, const Geo& g, const Tables* tb, int64_t trim, int64_t n_out, int64_t out_offset,
              bool weighted, int64_t fade_in, int64_t fade_out, const ModelRef* mr = nullptr);
bool split_in_stft(int B);
constexpr int kRankMinList = 24;     // shortest list bound for which the column sort is worth its time

}  // namespace repet_eng
