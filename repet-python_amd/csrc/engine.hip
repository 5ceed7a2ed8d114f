// Host orchestration and C ABI (include/repet_hip.h) of the gfx950 REPET engine.
//
// A repet_ctx owns one HIP stream, grow-only device workspaces and the per-window-length tables
// (periodic Hamming window, FFT twiddles). repet_ctx_execute chains the kernels of one variant on
// that stream with no host round trip in between (periods and index lists stay on the device).
#include "../../include/repet_hip.h"
#include "common.h"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

namespace repet {
hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex m;
    static std::map<std::pair<int, const void*>, int> granted;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(m);
    int& cur = granted[std::make_pair(dev, kernel)];
    if (bytes <= cur) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
}  // namespace repet

using namespace repet;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

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

}  // namespace

struct repet_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;   // short independent kernels run beside the main stream
    hipStream_t copy_stream = nullptr;   // the remainder plane of a float64 upload follows the samples here (created on first use)
    std::vector<hipStream_t> ballast_streams;   // candidates that shared the main stream's hardware queue (pick_side_stream)
    hipEvent_t fork_event = nullptr, join_event = nullptr;
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
    DevBuf lite_list, lite_flag, lite_records, frame_list, frame_flag;   // the wavefront kernel's fast path (peaks_wave.hip)
    unsigned int exact_gen = 0;
    bool refine_stats_cleared = false;   // ensure_spectra's housekeeping launch has zeroed them for the run being enqueued
    DevBuf R, Vs, rank_codes;     // rank codes of V, the sorted columns and the column-major codes (rank-domain median of `sim`, rank.hip)
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

namespace {

int ctx_create(int device, repet_ctx** out, bool probe_side_stream);

struct DeviceGuard {
    int prev = 0;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() { if (ok) (void)hipSetDevice(prev); }
};

int get_tables(repet_ctx* c, int W, Tables** out) {
    auto it = c->tables.find(W);
    if (it != c->tables.end()) { *out = it->second.get(); return REPET_OK; }
    if (W < 64 || W > 8192 || (W & (W - 1))) return fail(REPET_ERR_LIMIT, "window length must be a power of two in [64, 8192]");
    auto t = std::make_unique<Tables>();
    std::vector<float> win(W);
    std::vector<float2> tw(W);
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<double> wd(W);
    for (int n = 0; n < W; ++n) {
        wd[n] = 0.54 - 0.46 * std::cos(two_pi * n / W);   // scipy.signal.hamming(W, sym=False), repet.py:131
        win[n] = (float)wd[n];
        tw[n] = make_float2((float)std::cos(two_pi * n / W), (float)(-std::sin(two_pi * n / W)));
    }
    t->cola = wd[0] + wd[W / 2];
    std::vector<double2> tw64(W + 1);
    for (int n = 0; n <= W; ++n) tw64[n] = make_double2(std::cos(two_pi * n / W), -std::sin(two_pi * n / W));
    HIP_TRY(t->window64.ensure(W * sizeof(double)));
    HIP_TRY(t->twiddle64.ensure(tw64.size() * sizeof(double2)));
    HIP_TRY(hipMemcpyAsync(t->window64.p, wd.data(), W * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->twiddle64.p, tw64.data(), tw64.size() * sizeof(double2), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(t->window.ensure(W * sizeof(float)));
    HIP_TRY(t->twiddle.ensure(W * sizeof(float2)));
    HIP_TRY(hipMemcpyAsync(t->window.p, win.data(), W * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(t->twiddle.p, tw.data(), W * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out = t.get();
    c->tables[W] = std::move(t);
    return REPET_OK;
}

int upload_twiddle_only(repet_ctx* c, int W, const float2** tw) {
    Tables* t = nullptr;
    RP_TRY(get_tables(c, W, &t));
    *tw = t->twiddle.as<float2>();
    return REPET_OK;
}

// device tile list for nb tile rows and ndiag diagonals (cached per context)
int get_tiles(repet_ctx* c, int64_t T, int ndiag, const int2** tiles, int* count) {
    const int nb = (int)ceil_div(T, kTile);
    if (ndiag > nb) ndiag = nb;
    if (c->tiles_nb != nb || c->tiles_ndiag != ndiag) {
        std::vector<int2> host;
        const int n = gram_tile_list(nb, ndiag, &host);
        HIP_TRY(hipStreamSynchronize(c->stream));          // the previous list may still be in use
        HIP_TRY(c->tiles.ensure(std::max<size_t>(host.size() * sizeof(int2), 256)));
        HIP_TRY(hipMemcpyAsync(c->tiles.p, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->tiles_nb = nb; c->tiles_ndiag = ndiag; c->tiles_count = n;
    }
    *tiles = c->tiles.as<int2>();
    *count = c->tiles_count;
    return REPET_OK;
}

int run_band_window_sum(repet_ctx* c, const float* band, int64_t T, int LP, int n_lags, int n_freq, int64_t start0,
                        int64_t step, int64_t len, int n_windows, float* beat, int beat_pitch, int n_batch,
                        int64_t band_batch_stride, int64_t beat_batch_stride) {
    const int nb = n_batch > 0 ? n_batch : 1;
    const size_t need = (size_t)nb * std::max(n_windows, 1) * band_window_chunks(T, len) * LP * sizeof(float);
    HIP_TRY(c->beat_partial.ensure(std::max<size_t>(need, 256)));
    HIP_TRY(launch_band_window_sum(band, T, LP, n_lags, n_freq, start0, step, len, n_windows, beat, beat_pitch, n_batch,
                                   band_batch_stride, beat_batch_stride, c->beat_partial.as<float>(), c->stream));
    return REPET_OK;
}

// REPET_GRAM=f32 selects the exact-fp32 MFMA kernel for the similarity matrix of `sim` (default: the f16-split one)
bool gram_f16_enabled() {
    static const bool on = [] { const char* e = getenv("REPET_GRAM"); return !(e && e[0] == 'f' && e[1] == '3'); }();
    return on;
}

// REPET_GRAM_TILE=128 keeps the full similarity matrix on the 128 x 128-tile kernel (default: 256 x 256 tiles with LDS-DMA
// staging from 8 tile rows on, i.e. clips of about 45 s at 44.1 kHz)
bool gram_big_enabled() {
    static const bool on = [] { const char* e = getenv("REPET_GRAM_TILE"); return !(e && e[0] == '1'); }();
    return on;
}

// seg (nullable): segment records of S's rows for the peak picking (peaks.h), pitch seg_pitch: written by the 256 x 256
// kernel's epilogue, by a pass over S behind the other kernels
int run_gram_full(repet_ctx* c, const float* A, int64_t T, int FS, float* S, int64_t TS, bool unit_rows = false,
                  bool planes_ready = false, float* seg = nullptr, int seg_pitch = 0, bool* seg_written = nullptr) {
    if (seg_written) *seg_written = false;             // (the caller then runs launch_segment_maxima itself, as a stage of its own)
    // REPET_GRAM_SEGMENTS=0: the records by the pass over S also behind the 256 x 256 kernel (agreement test of the epilogue's)
    static const bool seg_in_epilogue = [] { const char* e = getenv("REPET_GRAM_SEGMENTS"); return !(e && e[0] == '0'); }();
    if (unit_rows && gram_f16_enabled() && gram_big_enabled() && T >= 8 * gram_big_tile()) {
        const int bt = gram_big_tile();
        const int nb = (int)ceil_div(T, bt);
        if (c->tiles_big_nb != nb) {
            std::vector<int2> host;
            const int n = gram_tile_list(nb, 1 << 30, &host);
            HIP_TRY(hipStreamSynchronize(c->stream));          // the previous list may still be in use
            HIP_TRY(c->tiles_big.ensure(std::max<size_t>(host.size() * sizeof(int2), 256)));
            HIP_TRY(hipMemcpyAsync(c->tiles_big.p, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->tiles_big_nb = nb; c->tiles_big_count = n;
        }
        // the planes of rows [round_up(T, 128), round_up(T, 256)) are read by the last tile row and never used: the
        // buffer only has to be that long
        const int64_t count = round_up(T, kTile) * FS;
        if (!planes_ready) {
            HIP_TRY(c->Vh.ensure((size_t)round_up(T, bt) * FS * 4));
            HIP_TRY(launch_split_f16(A, c->Vh.p, count, c->stream));
        }
        HIP_TRY(launch_gram_full_f16_big(c->Vh.p, T, FS, S, TS, c->tiles_big.as<int2>(), c->tiles_big_count, c->stream,
                                         seg_in_epilogue ? seg : nullptr, seg_pitch));
        if (seg_written) *seg_written = seg && seg_in_epilogue;
        else if (seg && !seg_in_epilogue) HIP_TRY(launch_segment_maxima(S, T, (int)T, TS, seg, seg_pitch, c->stream));
        return REPET_OK;
    }
    const int2* tiles; int n;
    RP_TRY(get_tiles(c, T, 1 << 30, &tiles, &n));
    if (unit_rows && gram_f16_enabled()) {      // rows are unit vectors (components in [0, 1]): safe for the f16 split
        const int64_t count = round_up(T, kTile) * FS;
        if (!planes_ready) {
            HIP_TRY(c->Vh.ensure((size_t)count * 4));
            HIP_TRY(launch_split_f16(A, c->Vh.p, count, c->stream));
        }
        HIP_TRY(launch_gram_full_f16(c->Vh.p, T, FS, S, TS, tiles, n, c->stream));
        if (seg && !seg_written) HIP_TRY(launch_segment_maxima(S, T, (int)T, TS, seg, seg_pitch, c->stream));
        return REPET_OK;
    }
    HIP_TRY(launch_gram_full(A, T, FS, S, TS, tiles, n, c->stream));
    if (seg && !seg_written) HIP_TRY(launch_segment_maxima(S, T, (int)T, TS, seg, seg_pitch, c->stream));
    return REPET_OK;
}
// unit_rows: A holds unit vectors (the similarity band of simonline), safe for the f16-split kernel; the beat-spectrum
// bands (power spectra, wide dynamic range) stay on the exact-fp32 one. B clips: a_stride / band_stride in elements.
// does the banded Gram of power spectra (beat spectrum) run on the f16-split kernel with row-scaled planes?
bool band_rows_on_f16(repet_ctx* c, int64_t T, int FS, int n_lags, int B, int64_t a_stride) {
    const int2* tiles; int n;
    if (get_tiles(c, T, gram_band_diagonals(n_lags), &tiles, &n) != REPET_OK) return false;
    return gram_f16_enabled() && (int64_t)n * B >= 512 && (B == 1 || a_stride == round_up(T, kTile) * FS);
}

// lookback (simonline): ask for band[j][l] = row j . row j - l; granted on the f16-split kernel only (c->band_lookback says so)
int run_gram_band(repet_ctx* c, const float* A, int64_t T, int FS, float* band, int n_lags, int LP, bool unit_rows = false,
                  int B = 1, int64_t a_stride = 0, int64_t band_stride = 0, bool planes_ready = false, bool lookback = false) {
    c->band_lookback = false;
    const int2* tiles; int n;
    RP_TRY(get_tiles(c, T, gram_band_diagonals(n_lags), &tiles, &n));
    // power spectra (beat spectrum): any range, so the split is scaled by the matrix's largest magnitude. Two extra
    // passes over the matrix (max, split): worth it from about two rounds of tiles on (the batched segments of
    // `extended`: 0.58 -> 0.43 ms at cfg 3), not for one clip's narrow band (0.16 -> 0.17 ms at cfg 2 / cfg 4 sizes)
    c->band_on_f16 = false;
    if (!unit_rows && (planes_ready || band_rows_on_f16(c, T, FS, n_lags, B, a_stride))) {
        c->band_on_f16 = true;
        const int64_t per_clip = round_up(T, kTile) * FS;
        const int64_t count = per_clip * B;
        const int64_t rows_per_clip = round_up(T, kTile);
        if (!planes_ready) {                                                         // (else: written by the STFT itself)
            HIP_TRY(c->Vh.ensure((size_t)count * 4));
            HIP_TRY(c->amax.ensure((size_t)rows_per_clip * B * sizeof(float)));      // one inverse scale per row
            HIP_TRY(launch_split_f16_rows(A, c->Vh.p, rows_per_clip * B, FS, c->amax.as<float>(), c->stream));
        }
        HIP_TRY(launch_gram_band_f16(c->Vh.p, T, FS, band, n_lags, LP, tiles, n, B, 2 * per_clip, band_stride, c->stream,
                                     c->amax.as<float>(), rows_per_clip));
        return REPET_OK;
    }
    if (unit_rows && gram_f16_enabled()) {
        c->band_on_f16 = true;
        const int64_t per_clip = round_up(T, kTile) * FS;
        if (B > 1 && a_stride != per_clip) return fail(REPET_ERR_BAD_ARG, "internal: batch stride of the unit rows");
        const int64_t count = per_clip * B;
        if (!planes_ready) {
            HIP_TRY(c->Vh.ensure((size_t)count * 4));
            HIP_TRY(launch_split_f16(A, c->Vh.p, count, c->stream));
        }
        c->band_lookback = lookback;
        HIP_TRY(launch_gram_band_f16(c->Vh.p, T, FS, band, n_lags, LP, tiles, n, B, 2 * per_clip, band_stride, c->stream, nullptr, 0,
                                     c->band_lookback));
        return REPET_OK;
    }
    HIP_TRY(launch_gram_band(A, T, FS, band, n_lags, LP, tiles, n, B, a_stride, band_stride, c->stream));
    return REPET_OK;
}

void mark(repet_ctx* c, const char* name, double bytes, double flops) {
    if (!c->timing) return;
    if (c->n_marks >= REPET_MAX_STAGES) return;
    while ((int)c->events.size() < c->event_base + REPET_MAX_STAGES + 1) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->events.push_back(e);
    }
    (void)hipEventRecord(c->events[c->event_base + c->n_marks + 1], c->stream);
    std::snprintf(c->timing->stage_name[c->n_marks], sizeof(c->timing->stage_name[0]), "%s", name);
    c->timing->stage_bytes[c->n_marks] = bytes;
    c->timing->stage_flops[c->n_marks] = flops;
    c->n_marks++;
}

void begin_timing(repet_ctx* c, repet_timing* t) {
    c->timing = t;
    c->n_marks = 0;
    if (!t) return;
    std::memset(t, 0, sizeof(*t));
    if (t != &c->series_timing) c->event_base = 0;
    while ((int)c->events.size() < c->event_base + REPET_MAX_STAGES + 1) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { c->timing = nullptr; return; }
        c->events.push_back(e);
    }
    (void)hipEventRecord(c->events[c->event_base], c->stream);
}

void end_timing(repet_ctx* c) {
    if (!c->timing) return;
    repet_timing* t = c->timing;
    t->n_stages = c->n_marks;
    const int b = c->event_base;
    for (int i = 0; i < c->n_marks; ++i) (void)hipEventElapsedTime(&t->stage_ms[i], c->events[b + i], c->events[b + i + 1]);
    if (c->n_marks > 0) (void)hipEventElapsedTime(&t->total_ms, c->events[b], c->events[b + c->n_marks]);
    c->timing = nullptr;
}

// Geometry shared by every variant.
struct Geo {
    int W, H, F, FS;
    int64_t T, Tpad, chan_stride;
    int C;
};

Geo make_geo(int W, int H, int64_t T, int C) {
    Geo g;
    g.W = W; g.H = H; g.F = W / 2 + 1; g.FS = (int)round_up(g.F, kFreqAlign);
    g.T = T; g.Tpad = round_up(T > 0 ? T : 1, kTile); g.chan_stride = (g.Tpad + kPadRows) * g.FS; g.C = C;
    return g;
}

// B: number of equal-geometry clips handled together (segments of `extended`); buffers are [B][C][rows][FS]
// The STFT epilogue can write the f16 planes of the unit rows itself instead of a separate pass over Vn. Measured at cfg 2
// (one clip): the split pass disappears (-0.011 ms) and the STFT grows by as much (+0.012 ms: two 2-byte stores per
// component from a thread that owns every 256th bin) -- no gain. Measured at cfg 5 (64 clips of 30 s): the split pass is
// 0.136 ms there, the STFT grows by 0.065: step 2.54 -> 2.47 ms. So: batches yes, single clips no.
bool split_in_stft(int B) { return B > 1 && gram_f16_enabled(); }

// The mask kernels read V, read X and write X: 20 bytes per cell, and the inverse STFT reads X again. With the mask as a
// plane of its own they write 4 bytes and the inverse STFT multiplies while it fetches (8 + 4): 20 instead of 28 bytes per
// cell over the two stages, the same products bit for bit (mul_rounded). The inverse kernel pays for its extra loads
// about what the byte count says, so it depends on the mask kernel whether the sum gains (mask + inverse, ms, same box):
//   extended cfg 3 (period mask, HBM-bound)      0.415 + 0.461 -> 0.249 + 0.498   default: plane
//   simonline cfg 5 (ten similar frames: HBM)    0.763 + 0.544 -> 0.603 + 0.608   default: plane
//   adaptive cfg 4                                0.077 + 0.055 -> 0.060 + 0.068   default: in place
//   sim cfg 2 (selection-bound mask)              0.50 + 0.072 -> 0.50 + 0.091     default: in place
// REPET_MASK_PLANE=0 / 1 / p: never / in every variant (with the repeating-segment model where a variant has one) / the same
// as a plain plane, without the model.
enum class MaskKind { period, adaptive, sim_float, sim_ranks };
int mask_plane_forced() {
    static const int forced = [] { const char* e = getenv("REPET_MASK_PLANE"); return e ? (e[0] == '0' ? 0 : (e[0] == 'p' ? 2 : 1)) : -1; }();
    return forced;
}
bool mask_plane_wanted(MaskKind kind) {
    return mask_plane_forced() >= 0 ? mask_plane_forced() != 0 : (kind == MaskKind::period || kind == MaskKind::sim_float);
}

struct MaskPlaneScope {            // the choice holds for one pipeline; stage exports and the streaming handle never see it
    repet_ctx* c;
    MaskPlaneScope(repet_ctx* ctx, bool on) : c(ctx) { c->mask_plane = on; }
    ~MaskPlaneScope() { c->mask_plane = false; }
};

// p_planes: the forward STFT will write the row-scaled f16 planes of the power spectra (prepare_power_planes): their pad
// rows [T, Tpad) of every clip are zeroed by the same housekeeping launch (as a 2-D memset they were 61 us at cfg 3)
int ensure_spectra(repet_ctx* c, const Geo& g, bool want_vn, bool want_p, int B = 1, bool p_planes = false) {
    if (g.W > 4096) c->mask_plane = false;       // the 8192-sample inverse kernel has no registers to spare for the mask
    HIP_TRY(c->X.ensure((size_t)B * g.C * g.chan_stride * sizeof(float2)));
    HIP_TRY(c->V.ensure((size_t)B * g.C * g.chan_stride * sizeof(float)));
    if (c->mask_plane && !c->mask_model) HIP_TRY(c->Mk.ensure((size_t)B * g.C * g.chan_stride * sizeof(float)));
    if ((size_t)g.chan_stride * 4 >= (size_t)1 << 31) return fail(REPET_ERR_LIMIT, "clip too long: one channel's spectrogram must stay below 2 GiB");
    const size_t mean_elems = (size_t)g.Tpad * g.FS;
    if (want_vn) HIP_TRY(c->Vn.ensure(B * mean_elems * sizeof(float)));
    HIP_TRY(c->refine_stats.ensure(kStatWords * sizeof(unsigned int)));
    // one launch: the pad rows of V, zeros over the rows [T, Tpad) of every clip's unit spectra (the Gram tiles read
    // them), the counters of the peak refinement (make_refine then skips its own clear)
    if (p_planes && !want_vn) HIP_TRY(c->Vh.ensure((size_t)B * mean_elems * 4));
    // the f16 planes of the unit rows, written by the STFT beside Vn (same bytes per row: 2 planes x 2 bytes); the big-tile
    // Gram kernel reads (and ignores) up to round_up(T, 256) rows of a single clip
    const bool unit_planes = want_vn && split_in_stft(B);
    if (unit_planes) HIP_TRY(c->Vh.ensure((B == 1 ? (size_t)round_up(g.T, 256) * g.FS : B * mean_elems) * sizeof(float)));
    const bool planes_pad = (unit_planes || (p_planes && !want_vn)) && g.Tpad > g.T;
    HIP_TRY(launch_fill_pad_rows(c->V.as<float>(), g.chan_stride, B * g.C, g.Tpad, g.FS, c->stream,
                                 want_vn ? c->Vn.as<float>() + g.T * g.FS : nullptr, (int64_t)mean_elems,
                                 (int64_t)(g.Tpad - g.T) * g.FS, B, c->refine_stats.as<unsigned int>(),
                                 planes_pad ? c->Vh.as<float>() + g.T * g.FS : nullptr));      // (one launch: a 2-D memset of the planes' pad rows was 45-61 us)
    c->refine_stats_cleared = true;
    if (want_p) {
        HIP_TRY(c->P.ensure(B * mean_elems * sizeof(float)));
        if (B == 1) HIP_TRY(hipMemsetAsync(c->P.as<float>() + g.T * g.FS, 0, (size_t)(g.Tpad - g.T) * g.FS * sizeof(float), c->stream));
        else HIP_TRY(hipMemsetAsync(c->P.p, 0, B * mean_elems * sizeof(float), c->stream));
    }
    return REPET_OK;
}

int run_stft(repet_ctx* c, const Geo& g, const Tables* tb, int64_t offset, int64_t n, int centred, bool vn, bool p,
             int B = 1, int64_t batch_sample_stride = 0, bool p_as_planes = false) {
    StftArgs a{};
    a.audio = c->audio.as<float>(); a.n_samples = n; a.n_channels = g.C; a.sample_offset = c->clip_base + offset;
    a.window = tb->window.as<float>(); a.twiddle = tb->twiddle.as<float2>();
    a.W = g.W; a.H = g.H; a.T = g.T; a.FS = g.FS; a.centred = centred;
    a.X = c->X.as<float2>(); a.V = c->V.as<float>(); a.chan_stride = g.chan_stride;
    a.Vm = nullptr; a.Vn = vn ? c->Vn.as<float>() : nullptr; a.P = (p && !p_as_planes) ? c->P.as<float>() : nullptr;
    if (p_as_planes) { a.Ph = c->Vh.p; a.Ph_inv = c->amax.as<float>(); a.batch_inv_stride = g.Tpad; }
    a.Vh = (vn && split_in_stft(B)) ? c->Vh.p : nullptr;
    a.n_batch = B; a.batch_sample_stride = batch_sample_stride; a.batch_spec_stride = (int64_t)g.C * g.chan_stride;
    a.batch_mean_stride = g.Tpad * g.FS;
    HIP_TRY(launch_stft(a, c->stream));
    const double in_b = 4.0 * n * g.C, spec_b = (8.0 + 4.0) * g.F * g.T * g.C, mean_b = 4.0 * g.F * g.T;
    mark(c, "stft", B * (in_b + spec_b + mean_b), 0);
    return REPET_OK;
}

MaskArgs mask_args(repet_ctx* c, const Geo& g, int cutoff) {
    MaskArgs m{};
    m.V = c->V.as<float>(); m.chan_stride = g.chan_stride; m.n_channels = g.C; m.T = g.T; m.F = g.F; m.FS = g.FS;
    m.X = c->mask_plane ? nullptr : c->X.as<float2>(); m.mask = c->mask_plane ? c->Mk.as<float>() : nullptr;
    m.cutoff = cutoff; m.pad_row = g.Tpad;
    m.n_batch = 1; m.batch_stride = (int64_t)g.C * g.chan_stride;
    return m;
}

// masked spectrum -> inverse FFT + overlap-add (one fused kernel) -> c->out
// the repeating-segment models of a batch for the inverse STFT (IstftOlaArgs::model)
struct ModelRef { const float* model; const int32_t* periods; int64_t batch_stride, chan_stride; int32_t cutoff; };
void apply_model(IstftOlaArgs& a, repet_ctx*, const ModelRef* mr) {
    if (!mr) return;
    a.M = nullptr; a.model = mr->model; a.periods = mr->periods;
    a.model_batch_stride = mr->batch_stride; a.model_chan_stride = mr->chan_stride; a.cutoff = mr->cutoff;
}

int run_istft(repet_ctx* c, const Geo& g, const Tables* tb, int64_t trim, int64_t n_out, int64_t out_offset,
              bool weighted, int64_t fade_in, int64_t fade_out, const ModelRef* mr = nullptr) {
    IstftOlaArgs a{};
    a.Y = c->X.as<float2>(); a.M = c->mask_plane ? c->Mk.as<float>() : nullptr; a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.W = g.W;
    a.twiddle = tb->twiddle.as<float2>(); a.trim = trim; a.out = c->out.as<float>(); a.n_out = n_out;
    a.out_offset = c->clip_base + out_offset; a.scale = (float)(1.0 / tb->cola);
    a.accumulate_weighted = weighted ? 1 : 0; a.fade_in = fade_in; a.fade_out = fade_out;
    apply_model(a, c, mr);
    hipError_t e = launch_istft_ola(a, c->stream);
    if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "too many channels for the fused inverse STFT");
    HIP_TRY(e);
    mark(c, "istft_ola", (mr ? 8.0 + 4.0 / 3 : c->mask_plane ? 12.0 : 8.0) * g.F * g.T * g.C + 4.0 * n_out * g.C, 0);      // (a model: a third of a plane at most)
    return REPET_OK;
}

// ---- original on B equal-length clips of the resident signal: clip b covers samples
// [offset + b*hop, offset + b*hop + n). B = 1, hop = 0 is repet.original itself; B > 1 are segments
// seg_first .. seg_first+B-1 of `extended` (of seg_total), whose outputs are cross-faded into c->out.
// The forward STFT's waves write the row-scaled f16 planes of the power spectra themselves (stft_reg.hip): workspace, the
// pad rows of every clip (zero planes, inverse scale 1: what the separate pass makes of zero rows).
int prepare_power_planes(repet_ctx* c, const Geo& g, int64_t T, int B) {
    const int64_t mean_stride = g.Tpad * g.FS;
    HIP_TRY(c->Vh.ensure((size_t)B * mean_stride * 4));            // (pad rows: zeroed by ensure_spectra's housekeeping launch)
    HIP_TRY(c->amax.ensure((size_t)B * g.Tpad * sizeof(float)));
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->amax.p), 0x3f800000, (size_t)B * g.Tpad, c->stream));
    return REPET_OK;
}

// the arguments of a fused inverse STFT as far as istft_reg_takes() looks at them
IstftOlaArgs reg_probe(int W, int channels, bool weighted, int64_t n_out, int64_t out_stride, int64_t overlap) {
    IstftOlaArgs a{};
    a.W = W; a.n_channels = channels; a.accumulate_weighted = weighted ? 1 : 0; a.n_out = n_out; a.batch_out_stride = out_stride;
    a.overlap = overlap; a.fade_in = overlap; a.fade_out = overlap;
    return a;
}

int run_original(repet_ctx* c, const repet_params* p, int64_t offset, int64_t n, int B, int64_t hop,
                 int32_t* period_slots, bool weighted, int seg_first, int seg_total, int64_t overlap) {
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, p->window_length, &tb));
    const int64_t T = repet_frame_count(n, p->window_length, p->step_length, 1);
    const Geo g = make_geo(p->window_length, p->step_length, T, c->n_channels);
    const int hi = (int)std::min<int64_t>(p->period_hi, T / 3);
    if (hi <= p->period_lo) return fail(REPET_ERR_TOO_SHORT, "attempt to get argmax of an empty sequence (clip too short for the period range)");
    const int LP = (int)round_up(hi, 64);
    const int64_t mean_stride = g.Tpad * g.FS, band_stride = g.Tpad * LP;
    // When the beat spectrum's Gram runs on the f16-split kernel (many segments) and the forward STFT is the
    // wave-per-frame kernel, the wave that owns a frame writes the row-scaled f16 planes of P itself: no fp32 P, no
    // second pass over it (extended 600 s: 0.34 -> 0.24 ms for the Gram stage).
    const bool p_planes = gram_f16_enabled() && g.Tpad == round_up(T, kTile) && reg_fft_supported(g.W, g.C, false) &&
                          (band_rows_on_f16(c, T, g.FS, hi, B, mean_stride) || (B == 1 && T >= 2048));   // (a long single clip: as in exec_adaptive)
    // The mask of a cell is soft_mask(V, W[frame mod period]) with W the medians over the repetitions -- [period][F] per clip
    // and channel, a third of a plane at most. On the register inverse STFT the mask kernel writes only W and the inverse
    // computes the mask where it multiplies it in, from |X| (magnitude(): the forward kernel's own V, bit for bit): no mask
    // plane written and read back, no second read of V by the mask kernel, no read of V by the inverse. cfg 3: mask_period
    // 0.24 -> 0.11 ms, inverse 0.48 -> 0.51, step 1.53 -> 1.42. REPET_MASK_PLANE=p: the plane.
    const bool model_wanted = mask_plane_forced() != 2;
    struct ModelScope { repet_ctx* c; ~ModelScope() { c->mask_model = false; } } model_scope{c};
    // (the launcher's own test, not a copy of it: only the register kernel applies a model, and launch_istft_ola refuses
    // one on the others)
    c->mask_model = model_wanted && c->mask_plane && istft_reg_takes(reg_probe(g.W, g.C, weighted, n, hop, overlap));
    const int model_rows = hi + 1;
    ModelRef model_ref{};
    RP_TRY(ensure_spectra(c, g, false, !p_planes, B, p_planes));
    if (c->mask_model) {
        HIP_TRY(c->Wm.ensure((size_t)B * g.C * model_rows * g.FS * sizeof(float)));
        model_ref = ModelRef{c->Wm.as<float>(), period_slots, (int64_t)g.C * model_rows * g.FS, (int64_t)model_rows * g.FS, p->cutoff_bins};
    }
    const ModelRef* mr = c->mask_model ? &model_ref : nullptr;
    if (p_planes) RP_TRY(prepare_power_planes(c, g, T, B));
    RP_TRY(run_stft(c, g, tb, offset, n, 1, false, true, B, hop, p_planes));
    HIP_TRY(c->band.ensure((size_t)B * band_stride * sizeof(float)));
    HIP_TRY(c->beat.ensure((size_t)B * LP * sizeof(float)));
    RP_TRY(run_gram_band(c, c->P.as<float>(), T, g.FS, c->band.as<float>(), hi, LP, false, B, mean_stride, band_stride, p_planes));
    mark(c, c->band_on_f16 ? "gram_band_f16x3" : "gram_band", B * (4.0 * g.F * T + 4.0 * T * hi), B * 2.0 * g.F * T * hi);
    RP_TRY(run_band_window_sum(c, c->band.as<float>(), T, LP, hi, g.F, 0, 0, T, 1, c->beat.as<float>(), LP, B, band_stride, LP));
    HIP_TRY(launch_periods(c->beat.as<float>(), B, LP, (int)T, p->period_lo, p->period_hi, period_slots, c->stream));
    mark(c, "beat_period", B * 4.0 * T * hi, 0);
    MaskArgs m = mask_args(c, g, p->cutoff_bins);
    m.n_batch = B;
    if (mr) { m.X = nullptr; m.mask = nullptr; m.model = c->Wm.as<float>(); m.model_batch_stride = mr->batch_stride; m.model_chan_stride = mr->chan_stride; }
    HIP_TRY(launch_mask_period(m, period_slots, 0, p->period_lo + 1, c->stream));
    if (mr) mark(c, "mask_period", B * (4.0 + 4.0 / 3) * g.F * T * g.C, 0);                        // the gathers of V, the model (a third of a plane at most)
    else mark(c, "mask_period", B * (4.0 + 4.0 + (c->mask_plane ? 4.0 : 16.0)) * g.F * T * g.C, 0);     // V, the gathers, the mask plane or X in place
    if (c->pre_synthesis) {
        std::function<int()> hook;
        hook.swap(c->pre_synthesis);
        RP_TRY(hook());
    }
    if (!weighted && B == 1) {
        RP_TRY(run_istft(c, g, tb, g.W - g.H, n, offset, false, 0, 0, mr));
    } else if (!weighted) {
        // independent clips of a batch context: clip b is written at offset + b*hop, no cross-fade
        IstftOlaArgs a{};
        a.Y = c->X.as<float2>(); a.M = c->mask_plane ? c->Mk.as<float>() : nullptr; a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.W = g.W;
        a.twiddle = tb->twiddle.as<float2>(); a.trim = g.W - g.H; a.out = c->out.as<float>(); a.n_out = n;
        a.out_offset = c->clip_base + offset; a.scale = (float)(1.0 / tb->cola); a.accumulate_weighted = 0;
        a.n_batch = B; a.batch_first = 0; a.batch_step = 1; a.batch_total = B; a.batch_local0 = 0;
        a.batch_spec_stride = (int64_t)g.C * g.chan_stride; a.batch_out_stride = hop; a.overlap = 0;
        apply_model(a, c, mr);
        hipError_t e = launch_istft_ola(a, c->stream);
        if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "too many channels for the fused inverse STFT");
        HIP_TRY(e);
        mark(c, "istft_ola", B * ((mr ? 8.0 + 4.0 / 3 : c->mask_plane ? 12.0 : 8.0) * g.F * g.T * g.C + 4.0 * n * g.C), 0);
    } else {
        // segments that overlap in the output must not be accumulated concurrently: one launch per residue
        // class modulo ceil(n / hop) (2 for the default 10 s / 5 s), each class writes disjoint samples
        const int classes = hop > 0 ? (int)ceil_div(n, hop) : 1;
        for (int k = 0; k < classes && k < B; ++k) {
            IstftOlaArgs a{};
            a.Y = c->X.as<float2>(); a.M = c->mask_plane ? c->Mk.as<float>() : nullptr; a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.W = g.W;
            a.twiddle = tb->twiddle.as<float2>(); a.trim = g.W - g.H; a.out = c->out.as<float>(); a.n_out = n;
            a.out_offset = c->clip_base; a.scale = (float)(1.0 / tb->cola);
            // class 0 of the first batch tiles its span of the cleared output exactly when the segment length is a whole
            // number of steps, and nothing has been added there yet: it stores, the other classes add
            a.accumulate_weighted = (k == 0 && c->ola_first_batch && hop > 0 && n == (int64_t)classes * hop) ? 2 : 1;
            a.n_batch = (B - k + classes - 1) / classes; a.batch_first = seg_first + k; a.batch_step = classes;
            a.batch_total = seg_total; a.batch_local0 = k; a.batch_spec_stride = (int64_t)g.C * g.chan_stride;
            a.batch_out_stride = hop > 0 ? hop : 0; a.overlap = overlap;
            if (hop == 0) a.out_offset = c->clip_base + offset;   // single (last) segment: explicit offset, j = seg_first
            apply_model(a, c, mr);
            hipError_t e = launch_istft_ola(a, c->stream);
            if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "too many channels for the fused inverse STFT");
            HIP_TRY(e);
        }
        mark(c, "istft_ola", B * ((mr ? 8.0 + 4.0 / 3 : c->mask_plane ? 12.0 : 8.0) * g.F * g.T * g.C + 4.0 * n * g.C), 0);
    }
    c->last_T = T;
    return REPET_OK;
}

int exec_original(repet_ctx* c, const repet_params* p) {
    MaskPlaneScope plane(c, mask_plane_wanted(MaskKind::period));
    // a batch context at its base runs all clips together (one launch per stage); otherwise the current clip
    const int nb = c->clip_loop ? 1 : c->n_clips;
    HIP_TRY(c->periods.ensure((size_t)nb * sizeof(int32_t)));
    RP_TRY(run_original(c, p, 0, c->n_samples, nb, nb > 1 ? c->n_samples : 0, c->periods.as<int32_t>(), false, 0, 1, 0));
    c->last_n_periods = nb;
    return REPET_OK;
}

int64_t extended_segment_count(int64_t N, const repet_params* p) {
    const int64_t L = p->seg_len_samples, Hs = p->seg_step_samples;
    if (L <= 0 || Hs <= 0) return -1;
    if (N < L + Hs) return 1;                             // repet.py:271-275: a single segment, whatever the step
    if (Hs > L) return -1;                                // several segments with a negative overlap: triang() raises
    return 1 + (N - L) / Hs;                              // repet.py:277-281
}

// segments [first, first+n_seg) of the resident clip; contributions of other segments are left zero,
// so partial results of disjoint ranges simply add up (repet.py:380-414 is linear in the segments).
// All segments but the last have the same length and run as ONE batch per stage; the last one
// (it absorbs the remainder, repet.py:320-322) runs on its own.
int exec_extended_plan(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg, int64_t N);

int exec_extended(repet_ctx* c, const repet_params* p, int64_t first = 0, int64_t n_seg = -1) {
    if (c->win_total <= 0) return exec_extended_plan(c, p, first, n_seg, c->n_samples);
    // a window of a longer clip: the plan is the whole clip's, sample s of it lives at s - win_offset here. clip_base is
    // the (signed) origin every read and write of the single-clip pipelines is relative to.
    const int64_t N = c->win_total, L = p->seg_len_samples, Hs = p->seg_step_samples;
    const int64_t count = extended_segment_count(N, p);
    if (count < 0) return fail(REPET_ERR_BAD_ARG, "extended: bad segment length/step (Window length M must be a non-negative integer)");
    if (n_seg < 0) n_seg = count - first;
    if (first < 0 || n_seg < 1 || first + n_seg > count) return fail(REPET_ERR_BAD_ARG, "extended: segment range outside the plan");
    const int64_t lo = count == 1 ? 0 : first * Hs;
    const int64_t hi = (first + n_seg == count) ? N : (first + n_seg - 1) * Hs + L;
    if (lo < c->win_offset || hi > c->win_offset + c->n_samples)
        return fail(REPET_ERR_BAD_ARG, "extended: the resident window does not hold the samples of this segment range");
    HIP_TRY(hipMemsetAsync(c->out.p, 0, (size_t)c->n_samples * c->n_channels * sizeof(float), c->stream));
    c->clip_base = -c->win_offset;
    c->win_skip_clear = true;
    const int rc = exec_extended_plan(c, p, first, n_seg, N);
    c->win_skip_clear = false;
    c->clip_base = 0;
    return rc;
}

int exec_extended_plan(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg, int64_t N) {
    MaskPlaneScope plane(c, mask_plane_wanted(MaskKind::period));
    const int64_t L = p->seg_len_samples, Hs = p->seg_step_samples;
    const int64_t count = extended_segment_count(N, p);
    if (count < 0) return fail(REPET_ERR_BAD_ARG, "extended: bad segment length/step (Window length M must be a non-negative integer)");
    if (n_seg < 0) n_seg = count - first;
    if (first < 0 || n_seg < 0 || first + n_seg > count) return fail(REPET_ERR_BAD_ARG, "extended: segment range outside the plan");
    if (count == 1) {                                               // repet.py:271
        if (n_seg == 1) return exec_original(c, p);
        if (!c->win_skip_clear)
            HIP_TRY(hipMemsetAsync(c->out.as<float>() + c->clip_base * c->n_channels, 0, (size_t)N * c->n_channels * sizeof(float), c->stream));
        return REPET_OK;
    }
    const int64_t O = L - Hs;
    HIP_TRY(c->periods.ensure((size_t)std::max<int64_t>(n_seg, 1) * sizeof(int32_t)));
    const int64_t last = count - 1;
    const int64_t uniform = std::min(first + n_seg, last) - first;  // equal-length segments in the range
    constexpr int64_t kMaxSegmentBatch = 256;
    if (!c->win_skip_clear) {
        // Class 0 of the first batch STORES its span (run_original: accumulate_weighted = 2 when the segment length is a whole
        // number of steps) -- with the register inverse STFT, which honours that mode, those samples need no clearing: at
        // cfg 3 that is all but the last 441 000 of 26 460 000 samples (212 MB of memset, 40 us). Other kernels add onto
        // the cleared output whatever the mode says, so they get the whole clear.
        int64_t s0 = 0, s1 = 0;                                     // [s0, s1): stored by class 0 of the first batch
        if (uniform > 0 && Hs > 0 && L == ceil_div(L, Hs) * Hs &&
            istft_reg_takes(reg_probe(p->window_length, c->n_channels, true, L, Hs, O))) {
            const int64_t classes = ceil_div(L, Hs), nb0 = std::min(uniform, kMaxSegmentBatch);
            const int64_t n_class0 = (nb0 + classes - 1) / classes;
            s0 = first * Hs;
            s1 = (first + (n_class0 - 1) * classes) * Hs + L;
        }
        float* o = c->out.as<float>() + c->clip_base * c->n_channels;
        if (s0 > 0) HIP_TRY(hipMemsetAsync(o, 0, (size_t)s0 * c->n_channels * sizeof(float), c->stream));
        if (s1 < N) HIP_TRY(hipMemsetAsync(o + s1 * c->n_channels, 0, (size_t)(N - s1) * c->n_channels * sizeof(float), c->stream));
    }
    // the equal-length segments go through every stage as ONE batch -- in bounded batches, so that the workspaces of an
    // hours-long recording stay at a few GB (a segment's spectra are about 15 MB at 44.1 kHz stereo)
    auto run_uniform = [&]() -> int {
        for (int64_t done = 0; done < uniform; done += kMaxSegmentBatch) {
            const int64_t nb = std::min(kMaxSegmentBatch, uniform - done);
            repet_timing* timing = c->timing;
            if (done > 0) c->timing = nullptr;                          // stages are listed once, for the first batch
            c->ola_first_batch = done == 0;
            const int rc = run_original(c, p, (first + done) * Hs, L, (int)nb, Hs, c->periods.as<int32_t>() + done, true,
                                        (int)(first + done), (int)count, O);
            c->ola_first_batch = false;
            c->timing = timing;
            if (rc != REPET_OK) return rc;
        }
        return REPET_OK;
    };
    const bool with_last = first + n_seg == count;                  // the longer last segment, repet.py:320-322
    if (with_last && uniform > 0) {
        // One small clip through eight kernels is a chain of launch latencies (0.22 ms at cfg 3) -- beside the batch it
        // is free: its analysis is enqueued on the auxiliary stream FIRST, the batch follows on the main stream, and
        // only the last segment's inverse STFT (it accumulates into samples the batch also writes) waits for the batch.
        if (!c->aux) {
            RP_TRY(ctx_create(c->device, &c->aux, false));     // (an auxiliary context runs nothing on its side stream: no probe)
            HIP_TRY(hipEventCreateWithFlags(&c->aux_start, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->aux_main_done, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->aux_done, hipEventDisableTiming));
        }
        repet_ctx* x = c->aux;
        x->audio.borrow(c->audio.p, c->audio.cap);
        x->out.borrow(c->out.p, c->out.cap);
        x->n_samples = c->n_samples; x->n_channels = c->n_channels; x->n_clips = 1; x->clip_base = c->clip_base;
        x->timing = nullptr;
        MaskPlaneScope aux_plane(x, c->mask_plane);
        HIP_TRY(hipEventRecord(c->aux_start, c->stream));              // the clip is resident, `out` is cleared
        HIP_TRY(hipStreamWaitEvent(x->stream, c->aux_start, 0));
        int batch_rc = REPET_OK;
        hipStream_t aux_stream = x->stream;
        x->pre_synthesis = [&]() -> int {
            batch_rc = run_uniform();
            if (batch_rc != REPET_OK) return batch_rc;
            // The last segment's inverse STFT goes on the MAIN stream, behind the batch's (it accumulates into samples the
            // batch also writes): the main stream waits for the analysis on the auxiliary one -- long finished -- and the
            // launch follows the batch directly. (On the auxiliary stream it was two more stream hops: main -> aux before
            // it, aux -> main behind it.)
            if (hipEventRecord(c->aux_main_done, aux_stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->aux_main_done, 0) != hipSuccess)
                return fail(REPET_ERR_HIP, "extended: stream ordering of the last segment");
            x->stream = c->stream;
            return REPET_OK;
        };
        const int rc = run_original(x, p, last * Hs, N - last * Hs, 1, 0, c->periods.as<int32_t>() + uniform, true, (int)last, (int)count, O);
        x->pre_synthesis = nullptr;
        x->stream = aux_stream;
        if (rc != REPET_OK) { (void)hipStreamSynchronize(aux_stream); (void)hipStreamSynchronize(c->stream); return rc; }
        mark(c, "last_segment", 0, 0);
    } else {
        if (uniform > 0) RP_TRY(run_uniform());
        if (with_last) {
            repet_timing* timing = c->timing;                           // its stages are not listed separately
            if (uniform > 0) c->timing = nullptr;
            int rc = run_original(c, p, last * Hs, N - last * Hs, 1, 0, c->periods.as<int32_t>() + uniform, true, (int)last, (int)count, O);
            c->timing = timing;
            if (rc != REPET_OK) return rc;
            if (uniform > 0) mark(c, "last_segment", 0, 0);
        }
    }
    c->last_n_periods = (int32_t)n_seg;
    return REPET_OK;
}

int exec_adaptive(repet_ctx* c, const repet_params* p) {
    MaskPlaneScope plane(c, mask_plane_wanted(MaskKind::adaptive));
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, p->window_length, &tb));
    const int64_t N = c->n_samples;
    const int64_t T = repet_frame_count(N, p->window_length, p->step_length, 1);
    const Geo g = make_geo(p->window_length, p->step_length, T, c->n_channels);
    const int Ls = p->seg_len_frames, Hs = p->seg_step_frames;
    if (Ls <= 0 || Hs <= 0) return fail(REPET_ERR_BAD_ARG, "adaptive: bad segment length/step");
    const int hi = std::min(p->period_hi, Ls / 3);
    if (hi <= p->period_lo) return fail(REPET_ERR_TOO_SHORT, "attempt to get argmax of an empty sequence (segment too short for the period range)");
    if (p->filter_order < 1) return fail(REPET_ERR_BAD_ARG, "adaptive: filter_order must be >= 1");
    // One long clip's narrow band did not pay for the two extra passes of the f16 split (0.16 -> 0.17 ms at cfg 4); with the
    // planes written by the forward STFT's own waves there are no extra passes.
    const bool p_planes = gram_f16_enabled() && g.Tpad == round_up(T, kTile) &&
                          reg_fft_supported(g.W, g.C, false) && T >= 2048;
    RP_TRY(ensure_spectra(c, g, false, !p_planes, 1, p_planes));
    if (p_planes) RP_TRY(prepare_power_planes(c, g, T, 1));
    RP_TRY(run_stft(c, g, tb, 0, N, 1, false, true, 1, 0, p_planes));
    const int LP = (int)round_up(hi, 64);
    const int n_win = (int)ceil_div(T, Hs);
    HIP_TRY(c->band.ensure((size_t)g.Tpad * LP * sizeof(float)));
    HIP_TRY(c->beat.ensure((size_t)n_win * LP * sizeof(float)));
    HIP_TRY(c->win_periods.ensure((size_t)n_win * sizeof(int32_t)));
    HIP_TRY(c->periods.ensure((size_t)T * sizeof(int32_t)));
    RP_TRY(run_gram_band(c, c->P.as<float>(), T, g.FS, c->band.as<float>(), hi, LP, false, 1, 0, 0, p_planes));
    mark(c, c->band_on_f16 ? "gram_band_f16x3" : "gram_band", 4.0 * g.F * T + 4.0 * T * hi, 2.0 * g.F * T * hi);
    const int64_t left = (Ls - 1 + 1) / 2;    // ceil((Ls-1)/2), repet.py:1182
    RP_TRY(run_band_window_sum(c, c->band.as<float>(), T, LP, hi, g.F, -left, Hs, Ls, n_win, c->beat.as<float>(), LP, 1, 0, 0));
    HIP_TRY(launch_periods(c->beat.as<float>(), n_win, LP, Ls, p->period_lo, p->period_hi, c->win_periods.as<int32_t>(), c->stream));
    HIP_TRY(launch_expand_periods(c->win_periods.as<int32_t>(), n_win, Hs, T, p->period_lo, c->periods.as<int32_t>(), c->stream));
    mark(c, "beat_periods", 4.0 * n_win * (double)Ls * hi, 0);
    HIP_TRY(launch_mask_adaptive(mask_args(c, g, p->cutoff_bins), c->periods.as<int32_t>(), p->filter_order, c->stream));
    mark(c, "mask_adaptive", (4.0 + 4.0 * p->filter_order + 16.0) * g.F * T * g.C, 0);
    RP_TRY(run_istft(c, g, tb, g.W - g.H, N, 0, false, 0, 0));
    c->last_T = T;
    c->last_n_periods = (int32_t)T;
    return REPET_OK;
}

// Near-tie refinement of the peak picking (peaks.hip): the tolerance inside which an fp32 similarity is not
// trusted, delta = scale * sqrt(FS) * 2^-24 (an error random walk over the FS products of unit-vector components).
// Measured against float64 on MI355X at FS = 1056 (tools/refine_probe.py --ambiguity):
//   exact-fp32 MFMA chain : rms 3.8e-7, max 5.9e-6  -> scale 4 (7.7e-6); 4x, 8x, 16x give identical index lists
//                           (0, 0, 0, 1 of 8062 rows differ from the float64 oracle; plain fp32: 71, 31, 35, 110),
//                           2x loses one more row
//   f16-split Gram kernel : rms 1.3e-7, max 1.1e-6  -> scale 2 (3.9e-6); 1x, 2x and 4x give identical lists
// The cost grows with delta (cfg 2, f16 Gram: peaks 0.28 / 0.30 / 0.34 ms at 1x / 2x / 4x).
float peak_refine_delta(int FS, bool f16_gram) { return (f16_gram ? 2.0f : 4.0f) * sqrtf((float)FS) * 5.9604645e-8f; }

// Second level (peaks_exact.hip): a float64 comparison of the fp32 spectra closer than this is decided again from float64
// spectra. The level-1 values are off by up to 9.3e-8 against the float64 reference (rms 1.2e-8: fp32 FFT, magnitudes and
// unit rows; tools/level_error_probe.py; on the device `level2_max_diff` of repet_ctx_last_exact_stats reports the largest
// difference met), a comparison of two of them by up to twice that; DESIGN.md 1 derives the band from the error of the fp32
// spectra. REPET_PEAK_EXACT=0 turns the second level off.
double peak_exact_delta2() {
    static const double v = [] { const char* off = getenv("REPET_PEAK_EXACT"); return (off && off[0] == '0') ? 0.0 : 2.5e-7; }();
    return v;
}

// A flag array of `count` generation stamps: grown (and cleared) when too small; stamps of earlier runs never match a new
// generation, so it is not cleared between runs.
int ensure_stamps(repet_ctx* c, DevBuf& buf, size_t count) {
    if (buf.cap >= count * sizeof(unsigned int)) return REPET_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(buf.ensure(count * sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(buf.p, 0, buf.cap, c->stream));
    return REPET_OK;
}

// rows x clips: the rows one launch_local_maxima call may hand to the second level (0: no second level for this call);
// n_cols, d: the call's row length and window (the wavefront kernel's fast path applies to its shapes); frames: frame
// rows per clip (the float64 unit-row table)
int make_refine(repet_ctx* c, const float* unit_rows, int FS, double threshold, PeakRefine* rf, int64_t rows = 0, int clips = 1,
                int n_cols = 0, int d = 0, int64_t frames = 0) {
    HIP_TRY(c->refine_stats.ensure(kStatWords * sizeof(unsigned int)));
    if (!c->refine_stats_cleared) HIP_TRY(hipMemsetAsync(c->refine_stats.p, 0, kStatWords * sizeof(unsigned int), c->stream));
    c->refine_stats_cleared = false;
    *rf = PeakRefine{};
    rf->unit_rows = unit_rows; rf->pitch = FS; rf->delta = peak_refine_delta(FS, gram_f16_enabled()); rf->min_value = threshold;
    rf->stats = c->refine_stats.as<unsigned int>();
    if (rows > 0 && rf->delta > 0.0f && peak_exact_delta2() > 0.0) {
        const size_t total = (size_t)rows * clips;
        HIP_TRY(c->redo_list.ensure(total * 2 * sizeof(int32_t)));
        RP_TRY(ensure_stamps(c, c->redo_flag, total));
        rf->delta2 = peak_exact_delta2(); rf->redo_list = c->redo_list.as<int32_t>(); rf->redo_flag = c->redo_flag.as<unsigned int>();
        rf->gen = ++c->exact_gen; rf->flag_stride = rows;
        int record_bytes = 0;
        if (frames > 0 && local_maxima_wave_supported(n_cols, d, &record_bytes)) {
            HIP_TRY(c->lite_list.ensure(total * 2 * sizeof(int32_t)));
            RP_TRY(ensure_stamps(c, c->lite_flag, total));
            HIP_TRY(c->lite_records.ensure(total * (size_t)record_bytes));
            HIP_TRY(c->frame_list.ensure((size_t)frames * clips * sizeof(int32_t)));
            RP_TRY(ensure_stamps(c, c->frame_flag, (size_t)frames * clips));
            rf->records = c->lite_records.as<unsigned char>(); rf->record_bytes = record_bytes;
            rf->lite_list = c->lite_list.as<int32_t>(); rf->lite_flag = c->lite_flag.as<unsigned int>();
            rf->frame_list = c->frame_list.as<int32_t>(); rf->frame_flag = c->frame_flag.as<unsigned int>();
            rf->frame_clip_stride = frames;
        }
    }
    return REPET_OK;
}

// The second level behind a launch_local_maxima call with the same matrix arguments: float64 spectra of frame row fr of clip
// b start at sample frame_sample0 + fr * H of `hi` (+ `lo`), clips clip_stride elements apart, n_frames rows per clip.
int run_exact_rows(repet_ctx* c, const Tables* tb, const Geo& g, const float* M, int64_t row0, int n_cols, int64_t pitch, int mode,
                   float min_value, int d, int number, int32_t* idx, int idx_pitch, int32_t* count, int64_t shift,
                   const PeakRefine& rf, const PeakBatch* batch, const float* hi, const float* lo, int64_t n_samples,
                   int64_t clip_stride, int64_t frame_sample0, int64_t n_frames, int clips) {
    if (!rf.redo_list) return REPET_OK;
    hipStream_t stream = c->stream;
    if (lo && c->ring.lo_in_flight) HIP_TRY(hipStreamWaitEvent(stream, c->ring.lo_done, 0));       // the remainder plane has arrived
    ExactSource src{};
    src.hi = hi; src.lo = lo; src.n_samples = n_samples; src.n_channels = g.C; src.clip_stride = clip_stride;
    src.frame_sample0 = frame_sample0; src.W = g.W; src.H = g.H; src.F = g.F; src.FS = g.FS;
    src.window64 = tb->window64.as<double>(); src.twiddle64 = tb->twiddle64.as<double2>();
    const size_t rows = (size_t)n_frames * clips;
    HIP_TRY(c->u64.ensure(rows * g.FS * sizeof(double)));
    if (c->u64_gen.cap < rows * sizeof(unsigned int)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(c->u64_gen.ensure(rows * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(c->u64_gen.p, 0, c->u64_gen.cap, c->stream));
    }
    src.u64 = c->u64.as<double>(); src.u64_clip_stride = n_frames * (int64_t)g.FS;
    src.u64_gen = c->u64_gen.as<unsigned int>(); src.gen_clip_stride = n_frames;
    if (rf.lite_list) {
        // fast path: the float64 unit rows of the queued frames, then the recorded rows again with them
        HIP_TRY(launch_unit_rows_f64(src, &rf, stream));
        HIP_TRY(launch_local_maxima(M, 0, row0, n_cols, pitch, mode, min_value, d, number, idx, idx_pitch, count, stream, shift,
                                    &rf, batch, nullptr, &src));
    }
    // general path: flat rows, rows of the workgroup kernel, rows the fast path handed on
    HIP_TRY(c->exact_scratch.ensure(local_maxima_exact_scratch_bytes(n_cols)));
    HIP_TRY(launch_local_maxima_exact(M, row0, n_cols, pitch, mode, min_value, d, number, idx, idx_pitch, count, stream, shift,
                                      &rf, batch, src, c->exact_scratch.p));
    return REPET_OK;
}

// REPET_MEDIAN=f32 keeps the selection of `sim` on the float magnitudes; default: the rank-domain form (rank.hip) when
// the clip is long enough for it to pay (the column sort is a fixed cost, the saving grows with the list length).
bool rank_median_enabled() {
    static const bool on = [] { const char* e = getenv("REPET_MEDIAN"); return !(e && e[0] == 'f'); }();
    return on;
}
constexpr int kRankMinList = 24;     // shortest list bound for which the column sort is worth its time

// Sort every column of V and fill m's rank fields (bins [0, F-1); the lone Nyquist bin stays on the float kernel).
int run_rank_columns(repet_ctx* c, const Geo& g, MaskArgs* m, hipStream_t stream, bool with_mark) {
    const int n_cols = g.F - 1;
    const int64_t vs_pitch = round_up(g.T, 32);
    HIP_TRY(c->R.ensure((size_t)g.C * g.chan_stride * sizeof(unsigned short)));
    HIP_TRY(c->Vs.ensure((size_t)g.C * n_cols * vs_pitch * sizeof(float)));
    HIP_TRY(c->rank_codes.ensure((size_t)g.C * n_cols * vs_pitch * sizeof(unsigned short)));
    if (c->r_pads_ptr != c->R.p || c->r_pads_stride != g.chan_stride || c->r_pads_row != g.Tpad || c->r_pads_channels != g.C ||
        c->r_pads_fs != g.FS) {
        HIP_TRY(launch_fill_rank_pad_rows(c->R.as<unsigned short>(), g.chan_stride, g.C, g.Tpad, g.FS, stream));
        c->r_pads_ptr = c->R.p; c->r_pads_stride = g.chan_stride; c->r_pads_row = g.Tpad; c->r_pads_channels = g.C; c->r_pads_fs = g.FS;
    }
    RankArgs a{};
    a.V = c->V.as<float>(); a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.n_cols = n_cols;
    a.R = c->R.as<unsigned short>(); a.r_chan_stride = g.chan_stride; a.Vs = c->Vs.as<float>(); a.vs_pitch = vs_pitch;
    a.codes = c->rank_codes.as<unsigned short>();
    HIP_TRY(launch_rank_columns(a, stream));
    m->R = a.R; m->r_chan_stride = a.r_chan_stride; m->Vs = a.Vs; m->vs_pitch = vs_pitch; m->n_rank_cols = n_cols;
    // V read, columns written / read twice / written sorted, codes written column-major, read, written frame-major
    if (with_mark) mark(c, "rank_columns", (4.0 + 4.0 + 8.0 + 4.0 + 2.0 + 2.0 + 2.0) * n_cols * (double)g.T * g.C, 0);
    return REPET_OK;
}

int exec_sim(repet_ctx* c, const repet_params* p) {
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, p->window_length, &tb));
    const int64_t N = c->n_samples;
    const int64_t T = repet_frame_count(N, p->window_length, p->step_length, 1);
    const Geo g = make_geo(p->window_length, p->step_length, T, c->n_channels);
    if (p->sim_number < 1) return fail(REPET_ERR_BAD_ARG, "similarity_number must be >= 1");
    // (the same test as below: the median on rank codes multiplies X in place, the float path keeps the mask apart)
    const bool ranks_ahead = rank_median_enabled() && g.F > 128 && ((g.F - 1) & 127) == 0 && rank_columns_supported(T) &&
                             std::min<int64_t>(p->sim_number, ceil_div(T, p->sim_distance_frames + 1)) >= kRankMinList &&
                             std::min<int64_t>(p->sim_number, ceil_div(T, p->sim_distance_frames + 1)) <= 128;
    MaskPlaneScope plane(c, mask_plane_wanted(ranks_ahead ? MaskKind::sim_ranks : MaskKind::sim_float));
    RP_TRY(ensure_spectra(c, g, true, false));
    RP_TRY(run_stft(c, g, tb, 0, N, 1, true, false));
    const int64_t TS = round_up(T, 64);
    HIP_TRY(c->S.ensure((size_t)T * TS * sizeof(float)));
    // segment records of S's rows: the peak picking takes its candidates from them instead of scanning S (peaks_wave.hip)
    const bool with_seg = local_maxima_segments_apply((int)T, p->sim_distance_frames, TS, 0, 1);
    const int seg_pitch = segment_pitch((int)TS);
    if (with_seg) HIP_TRY(c->seg.ensure((size_t)T * 3 * seg_pitch * sizeof(float)));
    float* seg = with_seg ? c->seg.as<float>() : nullptr;
    bool seg_written = false;
    RP_TRY(run_gram_full(c, c->Vn.as<float>(), T, g.FS, c->S.as<float>(), TS, true, split_in_stft(1), seg, seg_pitch, &seg_written));
    {
        // flops as EXECUTED: upper-triangle 128 x 128 tiles over the padded K = FS, three f16 products per term on the
        // split kernel (hi hi' + hi lo' + lo hi'); bench.py prices them against the f16 (or fp32) matrix peak and
        // states the algorithmic 2 F T^2 beside it
        const bool f16 = gram_f16_enabled();
        const int edge = (f16 && gram_big_enabled() && T >= 8 * gram_big_tile()) ? gram_big_tile() : kTile;
        const double n_tiles = 0.5 * (double)ceil_div(T, edge) * (double)(ceil_div(T, edge) + 1);
        mark(c, f16 ? "similarity_gemm_f16x3" : "similarity_gemm", 4.0 * g.F * T + 4.0 * T * T,
             (f16 ? 3.0 : 1.0) * 2.0 * g.FS * n_tiles * edge * edge);
    }
    if (seg && !seg_written) {                          // (short clips, the other Gram kernels: a pass over S)
        HIP_TRY(launch_segment_maxima(c->S.as<float>(), T, (int)T, TS, seg, seg_pitch, c->stream));
        mark(c, "segment_maxima", 4.0 * T * T + 12.0 * T * seg_pitch, 0);
    }
    const int K = p->sim_number, KP = std::max(K, kMinIdxPitch);
    HIP_TRY(c->idx.ensure((size_t)T * KP * sizeof(int32_t)));
    HIP_TRY(c->cnt.ensure((size_t)T * sizeof(int32_t)));
    // peaks are more than d frames apart: at most ceil(T/(d+1)) of them, whatever similarity_number says
    const int max_peaks = (int)std::min<int64_t>(K, ceil_div(T, p->sim_distance_frames + 1));
    PeakRefine rf{};
    RP_TRY(make_refine(c, c->Vn.as<float>(), g.FS, p->sim_threshold, &rf, T, 1, (int)T, p->sim_distance_frames, T));
    {
        MaskArgs m = mask_args(c, g, p->cutoff_bins);
        const bool use_rank = rank_median_enabled() && g.F > 128 && ((g.F - 1) & 127) == 0 && rank_columns_supported(T) &&
                              max_peaks >= kRankMinList && max_peaks <= 128;
        // The column sort needs nothing of the similarity matrix: it runs on the side stream BESIDE the peak picking, whose
        // rows take 30 .. 140 us each -- the second half of that launch is a tail of fewer and fewer waves (spans of every
        // row: tools/peak_stamps.py), which the sort's workgroups fill. (Beside the Gram kernel it does not pay: a sort
        // workgroup on a CU keeps the Gram's 139 KB workgroup off it -- and so does the memory-bound transpose that opens
        // the sort, although its 17 KB of LDS fit beside a Gram workgroup: Gram 0.209 -> 0.244 ms for 0.015 ms saved
        // afterwards.)
        const bool beside = use_rank;
        // (Measured and dropped: starting the sort behind the first pass of the peak picking, beside its second level --
        // peaks + sort 0.446 against 0.419 ms: the second level's kernels hold a whole register file per wave and do not share
        // a CU with the sort any better than the first pass does.)
        // (Measured and dropped: starting the sort behind the first pass of the peak picking, beside its second level --
        // peaks + sort 0.416 against 0.373 ms: the sort fills the first pass's tail better than it shares the GPU with the
        // one-wave-per-SIMD kernels of the second level.)
        if (beside) {
            HIP_TRY(hipEventRecord(c->fork_event, c->stream));          // V is complete (so is S)
            HIP_TRY(hipStreamWaitEvent(c->side_stream, c->fork_event, 0));
            RP_TRY(run_rank_columns(c, g, &m, c->side_stream, false));
            HIP_TRY(hipEventRecord(c->join_event, c->side_stream));
        }
        const size_t scratch = local_maxima_scratch_bytes(T, (int)T, p->sim_distance_frames);
        if (scratch > 0) HIP_TRY(c->peak_scratch.ensure(scratch));
        hipError_t e = launch_local_maxima(c->S.as<float>(), T, 0, (int)T, TS, 0, (float)p->sim_threshold,
                                           p->sim_distance_frames, K, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), c->stream, 0, &rf,
                                           nullptr, scratch > 0 ? c->peak_scratch.p : nullptr, nullptr, seg, seg_pitch);
        if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "sim: clip has too many frames for the peak-picking kernel's LDS row");
        HIP_TRY(e);
        // The second level of the peak picking: float64 spectra for the rows the fp32 spectra cannot settle (a few hundred of
        // 7 753 at cfg 2). Measured and dropped: running it on the side stream BESIDE the median mask of all the other rows
        // and masking its rows afterwards -- its kernels hold whole register files (one wave per SIMD) and the issue-bound
        // mask kernel loses more than the chain takes in line (1.256 against 1.178 ms per step).
        RP_TRY(run_exact_rows(c, tb, g, c->S.as<float>(), 0, (int)T, TS, 0, (float)p->sim_threshold, p->sim_distance_frames, K,
                              c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, rf, nullptr,
                              c->audio.as<float>() + c->clip_base * g.C, c->has_lo ? c->audio_lo.as<float>() + c->clip_base * g.C : nullptr,
                              N, 0, -(int64_t)(g.W / 2), T, 1));
        if (beside) {
            HIP_TRY(hipStreamWaitEvent(c->stream, c->join_event, 0));
            // one figure for the two concurrent launches: their bytes added up (S read once + the sort's passes over V)
            mark(c, "peaks+rank_columns", 4.0 * T * T + 4.0 * K * T + (4.0 + 4.0 + 8.0 + 4.0 + 2.0 + 2.0 + 2.0) * (g.F - 1) * (double)g.T * g.C, 0);
        } else {
            mark(c, "local_maxima", 4.0 * T * T + 4.0 * K * T, 0);
            if (use_rank) RP_TRY(run_rank_columns(c, g, &m, c->stream, true));
        }
        HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, max_peaks, c->stream, c->side_stream,
                                c->fork_event, c->join_event));
        mark(c, "mask_sim", (4.0 + 4.0 * K + (c->mask_plane ? 4.0 : 16.0)) * g.F * T * g.C, 0);
    }
    RP_TRY(run_istft(c, g, tb, g.W - g.H, N, 0, false, 0, 0));
    c->last_T = T; c->last_idx_rows = T; c->last_idx_pitch = KP; c->last_idx_number = K;
    return REPET_OK;
}

int exec_simonline(repet_ctx* c, const repet_params* p) {
    MaskPlaneScope plane(c, mask_plane_wanted(MaskKind::sim_float));
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, p->window_length, &tb));
    const int64_t N = c->n_samples;
    const int W = p->window_length, H = p->step_length, B = p->buffer_frames;
    if (B < 1) return fail(REPET_ERR_BAD_ARG, "buffer length must be >= 1 frame");
    if (N < (int64_t)(B - 2) * H + W)   // the warm-up slices B-1 whole frames (repet.py:795-810)
        return fail(REPET_ERR_TOO_SHORT, "operands could not be broadcast together (signal shorter than the buffer)");
    const int64_t T = repet_frame_count(N, W, H, 0);
    const Geo g = make_geo(W, H, T, c->n_channels);
    if (p->sim_number < 1) return fail(REPET_ERR_BAD_ARG, "similarity_number must be >= 1");
    // nb equal-shape clips (repet_ctx_upload_batch) go through every stage together: one launch per stage
    const int nb = c->clip_loop ? 1 : c->n_clips;
    RP_TRY(ensure_spectra(c, g, true, false, nb));
    RP_TRY(run_stft(c, g, tb, 0, N, 0, true, false, nb, N));
    const int LP = (int)round_up(B, 64);
    const int64_t mean_stride = g.Tpad * g.FS, band_stride = g.Tpad * LP, spec_stride = (int64_t)g.C * g.chan_stride;
    HIP_TRY(c->band.ensure((size_t)nb * band_stride * sizeof(float)));
    RP_TRY(run_gram_band(c, c->Vn.as<float>(), T, g.FS, c->band.as<float>(), B, LP, true, nb, mean_stride, band_stride, split_in_stft(nb), true));
    const int peak_mode = c->band_lookback ? 2 : 1;
    mark(c, c->band_on_f16 ? "similarity_band_f16x3" : "similarity_band", nb * (4.0 * g.F * T + 4.0 * T * B), nb * 2.0 * g.F * (double)T * B);
    const int K = p->sim_number, KP = std::max(K, kMinIdxPitch);
    const int64_t rows = T >= B ? T - B + 1 : 0;
    const int64_t rows_alloc = std::max<int64_t>(rows, 1);
    HIP_TRY(c->idx.ensure((size_t)nb * rows_alloc * KP * sizeof(int32_t)));
    HIP_TRY(c->cnt.ensure((size_t)nb * rows_alloc * sizeof(int32_t)));
    PeakRefine rf{};
    RP_TRY(make_refine(c, c->Vn.as<float>(), g.FS, p->sim_threshold, &rf, rows, nb, B, p->sim_distance_frames, T));
    const PeakBatch pb{nb, band_stride, rows_alloc * KP, rows_alloc, mean_stride};
    hipError_t e = launch_local_maxima(c->band.as<float>(), rows, B - 1, B, LP, peak_mode, (float)p->sim_threshold,
                                       p->sim_distance_frames, K, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), c->stream, 0, &rf,
                                       nb > 1 ? &pb : nullptr);
    if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "simonline: buffer too long for the peak-picking kernel");
    HIP_TRY(e);
    if (rows > 0)
        RP_TRY(run_exact_rows(c, tb, g, c->band.as<float>(), B - 1, B, LP, peak_mode, (float)p->sim_threshold, p->sim_distance_frames, K,
                              c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, rf, nb > 1 ? &pb : nullptr,
                              c->audio.as<float>() + c->clip_base * g.C, c->has_lo ? c->audio_lo.as<float>() + c->clip_base * g.C : nullptr,
                              N, N * g.C, 0, T, nb));
    mark(c, "local_maxima", nb * (4.0 * rows * B + 4.0 * K * rows), 0);
    const int max_peaks = (int)std::min<int64_t>(K, ceil_div(B, p->sim_distance_frames + 1));
    MaskArgs m = mask_args(c, g, p->cutoff_bins);
    m.n_batch = nb; m.batch_stride = spec_stride; m.idx_batch_stride = rows_alloc * KP; m.cnt_batch_stride = rows_alloc;
    HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), B - 1, max_peaks, c->stream, c->side_stream,
                            c->fork_event, c->join_event));
    mark(c, "mask_sim", nb * (4.0 + 4.0 * K + (c->mask_plane ? 4.0 : 16.0)) * g.F * (double)rows * g.C, 0);
    if (nb == 1) {
        RP_TRY(run_istft(c, g, tb, 0, N, 0, false, 0, 0));
    } else {
        IstftOlaArgs a{};
        a.Y = c->X.as<float2>(); a.M = c->mask_plane ? c->Mk.as<float>() : nullptr; a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.W = g.W;
        a.twiddle = tb->twiddle.as<float2>(); a.trim = 0; a.out = c->out.as<float>(); a.n_out = N;
        a.out_offset = 0; a.scale = (float)(1.0 / tb->cola); a.accumulate_weighted = 0;
        a.n_batch = nb; a.batch_first = 0; a.batch_step = 1; a.batch_total = nb; a.batch_local0 = 0;
        a.batch_spec_stride = spec_stride; a.batch_out_stride = N; a.overlap = 0;
        hipError_t e2 = launch_istft_ola(a, c->stream);
        if (e2 == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "too many channels for the fused inverse STFT");
        HIP_TRY(e2);
        mark(c, "istft_ola", nb * ((c->mask_plane ? 12.0 : 8.0) * g.F * g.T * g.C + 4.0 * N * g.C), 0);
    }
    c->last_T = T; c->last_idx_rows = rows; c->last_idx_pitch = KP; c->last_idx_number = K;
    c->last_idx_batch = rows >= 1 ? nb : 1;       // rows_alloc == rows then: the clips' lists are contiguous
    return REPET_OK;
}

int check_params(const repet_params* p) {
    if (!p) return fail(REPET_ERR_BAD_ARG, "params is null");
    if (p->window_length < 64 || p->window_length > 8192 || (p->window_length & (p->window_length - 1)))
        return fail(REPET_ERR_LIMIT, "window length must be a power of two in [64, 8192]");
    if (p->step_length * 2 != p->window_length) return fail(REPET_ERR_BAD_ARG, "step length must be half the window length");
    if (p->period_lo < 0 || p->cutoff_bins < 0 || p->sim_distance_frames < 0) return fail(REPET_ERR_BAD_ARG, "negative parameter");
    return REPET_OK;
}

// copy a dense host matrix [rows][cols] into a pitched device matrix (pad columns zeroed)
int h2d_pitched(repet_ctx* c, float* dst, int64_t dpitch, const float* src, int64_t rows, int64_t cols, int64_t rows_pad) {
    HIP_TRY(hipMemsetAsync(dst, 0, (size_t)rows_pad * dpitch * sizeof(float), c->stream));
    if (rows > 0)
        HIP_TRY(hipMemcpy2DAsync(dst, dpitch * sizeof(float), src, cols * sizeof(float), cols * sizeof(float), rows,
                                 hipMemcpyHostToDevice, c->stream));
    return REPET_OK;
}
int d2h_pitched(repet_ctx* c, float* dst, const float* src, int64_t spitch, int64_t rows, int64_t cols) {
    if (rows > 0)
        HIP_TRY(hipMemcpy2DAsync(dst, cols * sizeof(float), src, spitch * sizeof(float), cols * sizeof(float), rows,
                                 hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

// Contexts of the one-shot entry points (repet_run), one per host thread and device. The holder destroys them when the
// thread exits (streams, events and the grow-only workspaces -- for `sim` that includes the T x T similarity matrix);
// repet_release_thread_ctx does it on request.
struct ThreadContexts {
    std::map<int, repet_ctx*> by_device;
    ~ThreadContexts();
};
thread_local ThreadContexts g_thread_ctx;

}  // namespace

extern "C" {

int repet_abi_version(void) { return REPET_ABI_VERSION; }

int repet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* repet_last_error(void) { return g_last_error.c_str(); }

int64_t repet_frame_count(int64_t n, int32_t W, int32_t H, int32_t centred) {
    if (H <= 0) return 0;
    if (centred) {
        const int64_t pad = W / 2;                                   // repet.py:1018
        const int64_t num = n + 2 * pad - W;                         // repet.py:1024
        const int64_t q = num >= 0 ? (num + H - 1) / H : -((-num) / H);
        return q + 1;
    }
    const int64_t num = n - W;                                       // repet.py:781
    const int64_t q = num >= 0 ? (num + H - 1) / H : -((-num) / H);
    return q + 1;
}

namespace {

__global__ void queue_probe_kernel(unsigned long long ticks) {          // ticks of the 100 MHz clock; 0: nothing
    if (ticks == 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < (1 << 20) && __builtin_amdgcn_s_memrealtime() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(16);
}

hipError_t pick_side_stream(repet_ctx* c, bool probe_wanted) {
    const bool probe = probe_wanted;
    hipError_t e = hipSuccess;
    hipEvent_t t_begin = nullptr, t_end = nullptr;           // device-side timing: a loaded host (profiler, sanitizer build) must
    if (probe) {                                             // not make every candidate look serialised
        e = hipEventCreate(&t_begin);
        if (e == hipSuccess) e = hipEventCreate(&t_end);
        // the kernel's first launch (code object load) is not part of the test
        if (e == hipSuccess) { hipLaunchKernelGGL(queue_probe_kernel, dim3(1), dim3(64), 0, c->stream, 1ull); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    for (int attempt = 0; e == hipSuccess && attempt < 8; ++attempt) {
        hipStream_t cand = nullptr;
        e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) break;
        bool overlaps = true;
        if (probe) {
            // the pattern of a run: fork by event, two dependent kernels beside one, join by event -- 100 + 100 us of waiting
            // on the candidate beside 200 us on the main stream: about 0.2 ms when they overlap, 0.4 ms when they do not
            auto launch = [&](hipStream_t st, unsigned long long ticks) {
                if (e != hipSuccess) return;
                hipLaunchKernelGGL(queue_probe_kernel, dim3(1), dim3(64), 0, st, ticks);
                e = hipGetLastError();
            };
            e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipEventRecord(t_begin, c->stream);
            if (e == hipSuccess) e = hipEventRecord(c->fork_event, c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(cand, c->fork_event, 0);
            launch(cand, 10000ull);
            launch(cand, 10000ull);
            if (e == hipSuccess) e = hipEventRecord(c->join_event, cand);
            launch(c->stream, 20000ull);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->join_event, 0);
            if (e == hipSuccess) e = hipEventRecord(t_end, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, t_begin, t_end);
            overlaps = ms < 0.32f;
        }
        if (overlaps || attempt == 7) { c->side_stream = cand; break; }
        c->ballast_streams.push_back(cand);
    }
    if (t_begin) (void)hipEventDestroy(t_begin);
    if (t_end) (void)hipEventDestroy(t_end);
    return e;
}

}  // namespace

namespace {
int ctx_create(int device, repet_ctx** out, bool probe_side_stream) {
    if (!out) return fail(REPET_ERR_BAD_ARG, "out is null");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(REPET_ERR_BAD_ARG, "no such device");
    DeviceGuard guard(device);
    auto* c = new repet_ctx();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork_event, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join_event, hipEventDisableTiming);
    // The two streams must sit on DIFFERENT hardware queues, or the kernels meant to run side by side (column sort | peak
    // picking, Nyquist bins | mask) run one after the other. The runtime deals its GPU_MAX_HW_QUEUES = 4 queues by use
    // count: in a process that has already opened several streams (PyTorch with an RCCL communicator: seven) two streams
    // created back to back can both be given the same, least-used queue (rocprofv3 Queue_Id, tools/queue_trace.sh) --
    // repet.sim 1.02 -> 1.12 ms in every rank of a torch.distributed job. HIP does not tell which queue a stream has, and
    // independent kernels of two streams on one queue still overlap -- it is the fork / dependent kernels / join pattern of
    // a run that does not. So the context times exactly that pattern with wait kernels (pick_side_stream): 0.2 ms when the
    // candidate overlaps the main stream, 0.4 ms when not. A candidate that does not is kept open (it raises its queue's
    // use count, the next one goes elsewhere) until the context is destroyed.
    if (e == hipSuccess) e = pick_side_stream(c, probe_side_stream);
    if (e != hipSuccess) { repet_ctx_destroy(c); return fail(REPET_ERR_HIP, hipGetErrorString(e)); }
    *out = c;
    return REPET_OK;
}
}  // namespace

int repet_ctx_create(int device, repet_ctx** out) { return ctx_create(device, out, true); }

int repet_ctx_destroy(repet_ctx* c) {
    if (!c) return REPET_OK;
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->aux) {
        c->aux->audio.release(); c->aux->out.release();           // borrowed: just forgotten
        repet_ctx_destroy(c->aux);
        c->aux = nullptr;
        for (hipEvent_t e : {c->aux_start, c->aux_main_done, c->aux_done}) if (e) (void)hipEventDestroy(e);
    }
    c->ring.release();
    for (DevBuf* b : {&c->staging, &c->audio, &c->out, &c->out64, &c->X, &c->V, &c->Mk, &c->Wm, &c->Vn, &c->Vh, &c->amax, &c->beat_partial, &c->peak_scratch, &c->P, &c->S, &c->band, &c->beat,
                      &c->refine_stats, &c->R, &c->Vs, &c->rank_codes, &c->tiles_big,
                      &c->audio_lo, &c->redo_list, &c->redo_flag, &c->u64, &c->u64_gen, &c->exact_scratch,
                      &c->lite_list, &c->lite_flag, &c->lite_records, &c->frame_list, &c->frame_flag,
                      &c->seg,
                      &c->idx, &c->cnt, &c->periods, &c->win_periods, &c->frames, &c->tmp_a, &c->tmp_b, &c->tmp_c, &c->tiles})
        b->release();
    for (auto& kv : c->tables) { kv.second->window.release(); kv.second->twiddle.release(); kv.second->window64.release(); kv.second->twiddle64.release(); }
    for (hipEvent_t e : c->events) (void)hipEventDestroy(e);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipStream_t b : c->ballast_streams) (void)hipStreamDestroy(b);
    if (c->fork_event) (void)hipEventDestroy(c->fork_event);
    if (c->join_event) (void)hipEventDestroy(c->join_event);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return REPET_OK;
}

void repet_default_settings(repet_settings* s) {
    if (!s) return;
    s->cutoff_frequency = 100.0; s->period_range[0] = 1.0; s->period_range[1] = 10.0;       // repet.py:42-46
    s->segment_length = 10.0; s->segment_step = 5.0; s->filter_order = 5;                   // repet.py:50-54
    s->similarity_threshold = 0.0; s->similarity_distance = 1.0; s->similarity_number = 100; // repet.py:57-60
    s->buffer_length = 10.0;                                                                // repet.py:63
}

int repet_derive_params(const repet_settings* settings, double fs, repet_params* out) {
    if (!out) return fail(REPET_ERR_BAD_ARG, "params is null");
    if (!(fs > 0.0)) return fail(REPET_ERR_BAD_ARG, "sampling frequency must be positive");
    repet_settings d;
    repet_default_settings(&d);
    const repet_settings& s = settings ? *settings : d;
    // nearbyint under the default rounding mode is round-half-to-even, like Python's round() and np.round(). The
    // conversions to integers are checked: a value that is not finite or does not fit is an argument error, not a cast
    // with undefined behaviour (tools/asan_host_check.c runs this under UBSan).
    bool fits = true;
    auto rnd = [&fits](double x, double limit) -> int64_t {
        const double r = std::nearbyint(x);
        if (!(std::fabs(r) < limit)) { fits = false; return 0; }
        return (int64_t)r;
    };
    constexpr double k31 = 2147483648.0, k62 = 4611686018427387904.0;
    std::memset(out, 0, sizeof(*out));
    const double log_w = std::ceil(std::log2(0.04 * fs));                                    // repet.py:130
    if (!(log_w >= 1.0)) return fail(REPET_ERR_BAD_ARG, "sampling frequency too low: the 40-ms window has fewer than two samples");
    if (!(log_w <= 24.0)) return fail(REPET_ERR_LIMIT, "sampling frequency too high: window above 2^24 samples");
    const int w = 1 << (int)log_w;
    const int h = w / 2;                                                                     // repet.py:132
    out->window_length = w;
    out->step_length = h;
    out->period_lo = (int32_t)rnd(s.period_range[0] * fs / h, k31);                          // repet.py:165
    out->period_hi = (int32_t)rnd(s.period_range[1] * fs / h, k31);
    out->cutoff_bins = (int32_t)rnd(s.cutoff_frequency * w / fs, k31);                       // repet.py:173
    out->filter_order = s.filter_order;
    out->seg_len_frames = (int32_t)rnd(s.segment_length * fs / h, k31);                      // repet.py:519
    out->seg_step_frames = (int32_t)rnd(s.segment_step * fs / h, k31);                       // repet.py:520
    out->sim_distance_frames = (int32_t)rnd(s.similarity_distance * fs / h, k31);            // repet.py:670
    out->sim_number = s.similarity_number;
    out->buffer_frames = (int32_t)rnd((s.buffer_length * fs) / h, k31);                      // repet.py:787
    out->seg_len_samples = rnd(s.segment_length * fs, k62);                                  // repet.py:266
    out->seg_step_samples = rnd(s.segment_step * fs, k62);                                   // repet.py:267
    if (!fits) {
        std::memset(out, 0, sizeof(*out));
        return fail(REPET_ERR_BAD_ARG, "a setting times the sampling frequency is not a finite number that fits an integer");
    }
    out->sim_threshold = s.similarity_threshold;
    return REPET_OK;
}

int repet_ctx_upload_batch(repet_ctx* c, const void* audio, int dtype, int64_t n, int32_t ch, int32_t n_clips) {
    if (!c || !audio) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n < 0 || ch < 1) return fail(REPET_ERR_BAD_ARG, "audio_signal must be (number_samples, number_channels)");
    if (n_clips < 1) return fail(REPET_ERR_BAD_ARG, "n_clips must be >= 1");
    if (dtype < REPET_F32 || dtype > REPET_I16) return fail(REPET_ERR_BAD_ARG, "unsupported dtype");
    DeviceGuard guard(c->device);
    const int64_t count = n * ch * n_clips;
    HIP_TRY(c->audio.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    HIP_TRY(c->out.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    // narrowed to fp32 by host threads into the pinned ring, chunk by chunk, each chunk DMA'd while the next is
    // converted; the caller's array has been read completely when this returns (the last DMAs may still be in flight
    // on the context's stream, which every later operation of the context is ordered behind)
    // float64 input: the fp32 remainders travel too where they are not zero (peaks_exact.hip takes its float64 spectra from
    // sample + remainder)
    c->has_lo = false;
    float* lo_dst = nullptr;
    if (dtype == REPET_F64 && count > 0) {
        HIP_TRY(c->audio_lo.ensure((size_t)count * sizeof(float)));
        lo_dst = c->audio_lo.as<float>();
    }
    // The remainders are read by the second level of the peak picking only -- behind the STFT, the Gram matrix and the first
    // pass: they follow the samples on a stream of their own, and what is enqueued next starts when the SAMPLES are there
    // (run_exact_rows waits for ring.lo_done). Half of a float64 upload's bytes thus cross PCIe beside the computation.
    if (lo_dst && !c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    bool not_finite = false;
    HIP_TRY(staged_upload(c->ring, audio, dtype, c->audio.as<float>(), (size_t)count, c->stream, lo_dst, &c->has_lo, lo_dst ? c->copy_stream : nullptr,
                          &not_finite));
    if (not_finite) {
        // repet.py computes on, and NaN spreads from the frames that hold it through whatever is global in the variant (the
        // beat spectrum of original / extended / adaptive: whole segments or clips of NaN); host arrays with such samples are
        // refused instead ("Where the drop-in differs on purpose")
        c->n_channels = 0;
        return fail(REPET_ERR_BAD_ARG, "audio_signal contains NaN or infinite samples");
    }
    c->n_samples = n;
    c->n_channels = ch;
    c->n_clips = n_clips;
    c->clip_base = 0;
    c->win_total = 0; c->win_offset = 0;
    return REPET_OK;
}

int repet_ctx_upload_device_split(repet_ctx* c, const float* dev_audio, const float* dev_audio_lo, int64_t n, int32_t ch, int32_t n_clips) {
    if (!c || !dev_audio) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n < 0 || ch < 1 || n_clips < 1) return fail(REPET_ERR_BAD_ARG, "audio_signal must be (number_samples, number_channels)");
    DeviceGuard guard(c->device);
    const int64_t count = n * ch * n_clips;
    HIP_TRY(c->audio.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    HIP_TRY(c->out.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    // device -> device (peer memory works as well); the sources may be reused when this returns
    HIP_TRY(hipMemcpyAsync(c->audio.p, dev_audio, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    c->has_lo = false;
    if (dev_audio_lo && count > 0) {
        // the fp32 remainders of a float64 waveform (x - (double)(float)x): with them the second level of the peak picking
        // sees the 48 bits the single-GPU call sees
        HIP_TRY(c->audio_lo.ensure((size_t)count * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(c->audio_lo.p, dev_audio_lo, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        c->has_lo = true;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_samples = n; c->n_channels = ch; c->n_clips = n_clips; c->clip_base = 0;
    c->win_total = 0; c->win_offset = 0;
    return REPET_OK;
}

int repet_ctx_upload_device(repet_ctx* c, const float* dev_audio, int64_t n, int32_t ch, int32_t n_clips) {
    return repet_ctx_upload_device_split(c, dev_audio, nullptr, n, ch, n_clips);
}

int repet_ctx_download_device(repet_ctx* c, float* dev_out) {
    if (!c || !dev_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    const int64_t count = c->n_samples * c->n_channels * c->n_clips;
    if (count == 0) return REPET_OK;
    HIP_TRY(hipMemcpyAsync(dev_out, c->out.p, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int repet_ctx_set_window(repet_ctx* c, int64_t n_total, int64_t sample0) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    if (c->n_channels < 1 || c->n_clips != 1) return fail(REPET_ERR_BAD_ARG, "set_window applies to a single resident clip");
    if (n_total == 0 && sample0 == 0) { c->win_total = 0; c->win_offset = 0; return REPET_OK; }
    if (sample0 < 0 || n_total < sample0 + c->n_samples) return fail(REPET_ERR_BAD_ARG, "window outside the clip");
    c->win_total = n_total; c->win_offset = sample0;
    return REPET_OK;
}

int repet_ctx_upload(repet_ctx* c, const void* audio, int dtype, int64_t n, int32_t ch) {
    return repet_ctx_upload_batch(c, audio, dtype, n, ch, 1);
}

namespace {
int run_algo_one(repet_ctx* c, int algo, const repet_params* p) {
    // (the pipelines that never reach make_refine must not leave "cleared by the housekeeping launch" standing for a later
    // caller -- the streaming handle's make_refine -- to trust)
    struct StatsFlagScope { repet_ctx* c; ~StatsFlagScope() { c->refine_stats_cleared = false; } } stats_flag_scope{c};
    switch (algo) {
        case REPET_ORIGINAL: return exec_original(c, p);
        case REPET_EXTENDED: return exec_extended(c, p);
        case REPET_ADAPTIVE: return exec_adaptive(c, p);
        case REPET_SIM: return exec_sim(c, p);
        case REPET_SIMONLINE: return exec_simonline(c, p);
        default: return fail(REPET_ERR_BAD_ARG, "unknown algorithm");
    }
}

// A batch context (n_clips > 1): simonline and original run every stage once over all clips; the others work through
// the resident clips one after the other (their intermediates -- periods, index lists -- are those of the last).
int run_algo(repet_ctx* c, int algo, const repet_params* p) {
    c->clip_base = 0;
    if (c->win_total > 0) return fail(REPET_ERR_BAD_ARG, "the resident samples are a window of a longer clip: only repet_ctx_execute_extended_range applies");
    if (c->n_clips <= 1 || algo == REPET_SIMONLINE || algo == REPET_ORIGINAL) return run_algo_one(c, algo, p);
    repet_timing* timing = c->timing;
    c->timing = nullptr;                       // per-stage marks would repeat per clip: only the total is reported
    int rc = REPET_OK;
    c->clip_loop = true;
    for (int b = 0; b < c->n_clips && rc == REPET_OK; ++b) {
        c->clip_base = (int64_t)b * c->n_samples;
        rc = run_algo_one(c, algo, p);
    }
    c->clip_loop = false;
    c->clip_base = 0;
    c->timing = timing;
    if (rc == REPET_OK) mark(c, "clips", 0, 0);
    return rc;
}
}  // namespace

int repet_ctx_execute(repet_ctx* c, int algo, const repet_params* p, repet_timing* timing) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    RP_TRY(check_params(p));
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    DeviceGuard guard(c->device);
    begin_timing(c, timing);
    c->last_algo = algo;
    c->last_n_periods = 0;
    c->last_idx_rows = 0;
    c->last_idx_batch = 1;
    int rc = run_algo(c, algo, p);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (rc == REPET_OK && e != hipSuccess) rc = fail(REPET_ERR_HIP, std::string("execute: ") + hipGetErrorString(e));
    if (rc == REPET_OK) end_timing(c);
    c->timing = nullptr;
    return rc;
}

int repet_ctx_execute_async(repet_ctx* c, int algo, const repet_params* p) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    RP_TRY(check_params(p));
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    DeviceGuard guard(c->device);
    c->timing = nullptr;
    const bool timed = c->series_on && c->series_steps < c->series_cap;
    if (timed) {
        c->event_base = c->series_steps * (REPET_MAX_STAGES + 1);
        begin_timing(c, &c->series_timing);
    }
    c->last_algo = algo;
    c->last_n_periods = 0;
    c->last_idx_rows = 0;
    c->last_idx_batch = 1;
    const int rc = run_algo(c, algo, p);
    if (timed && c->timing) {
        c->series_marks = c->n_marks;
        c->series_steps++;
    }
    c->timing = nullptr;
    return rc;
}

int repet_ctx_timing_series_begin(repet_ctx* c, int32_t n_steps) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    if (n_steps < 1 || n_steps > 4096) return fail(REPET_ERR_BAD_ARG, "timing series: 1 to 4096 steps");
    DeviceGuard guard(c->device);
    while ((int64_t)c->events.size() < (int64_t)n_steps * (REPET_MAX_STAGES + 1)) {      // created here, not inside the timed region
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        c->events.push_back(e);
    }
    c->series_on = true;
    c->series_cap = n_steps;
    c->series_steps = 0;
    c->series_marks = 0;
    return REPET_OK;
}

int repet_ctx_timing_series_end(repet_ctx* c, repet_timing* mean, int32_t* n_steps) {
    if (!c || !mean) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    c->series_on = false;
    HIP_TRY(hipStreamSynchronize(c->stream));
    *mean = c->series_timing;                                  // names, bytes, flops of the last run
    mean->n_stages = c->series_marks;
    mean->total_ms = 0.f;
    for (int i = 0; i < REPET_MAX_STAGES; ++i) mean->stage_ms[i] = 0.f;
    const int steps = c->series_steps;
    for (int k = 0; k < steps; ++k) {
        const int b = k * (REPET_MAX_STAGES + 1);
        for (int i = 0; i < c->series_marks; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->events[b + i], c->events[b + i + 1]));
            mean->stage_ms[i] += ms / steps;
        }
        if (c->series_marks > 0) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->events[b], c->events[b + c->series_marks]));
            mean->total_ms += ms / steps;
        }
    }
    if (n_steps) *n_steps = steps;
    c->event_base = 0;
    return REPET_OK;
}

int repet_ctx_synchronize(repet_ctx* c) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int64_t repet_extended_segment_count(int64_t n_samples, const repet_params* p) {
    if (!p) return -1;
    return extended_segment_count(n_samples, p);
}

int repet_ctx_execute_extended_range(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg,
                                     repet_timing* timing) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    RP_TRY(check_params(p));
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    if (n_seg < 0) return fail(REPET_ERR_BAD_ARG, "negative segment count");
    if (c->n_clips > 1) return fail(REPET_ERR_BAD_ARG, "segment ranges apply to a single resident clip, not to a batch context");
    DeviceGuard guard(c->device);
    begin_timing(c, timing);
    c->last_algo = REPET_EXTENDED;
    c->last_n_periods = 0;
    int rc = exec_extended(c, p, first, n_seg);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (rc == REPET_OK && e != hipSuccess) rc = fail(REPET_ERR_HIP, std::string("execute: ") + hipGetErrorString(e));
    if (rc == REPET_OK) end_timing(c);
    c->timing = nullptr;
    return rc;
}

int repet_ctx_download(repet_ctx* c, double* out) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    const int64_t count = c->n_samples * c->n_channels * c->n_clips;
    if (count == 0) return REPET_OK;
    HIP_TRY(staged_download(c->ring, c->out.as<float>(), out, (size_t)count, c->stream));
    return REPET_OK;
}

int repet_wav_parse(const void* file_bytes, int64_t n_bytes, repet_wav_info* info) {
    const char* err = wav_parse(file_bytes, n_bytes, info);
    return err ? fail(REPET_ERR_BAD_ARG, err) : REPET_OK;
}

int repet_ctx_upload_wav(repet_ctx* c, const void* file_bytes, int64_t n_bytes, repet_wav_info* info_out) {
    if (!c || !file_bytes) return fail(REPET_ERR_BAD_ARG, "null argument");
    repet_wav_info w;
    RP_TRY(repet_wav_parse(file_bytes, n_bytes, &w));
    if (info_out) *info_out = w;
    DeviceGuard guard(c->device);
    const int64_t count = w.n_samples * w.n_channels;
    const size_t raw_bytes = (size_t)count * w.bytes_per_sample;
    HIP_TRY(c->audio.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    HIP_TRY(c->out.ensure(std::max<size_t>((size_t)count * sizeof(float), 256)));
    HIP_TRY(c->staging.ensure(std::max<size_t>(raw_bytes, 256)));
    HIP_TRY(staged_upload_bytes(c->ring, static_cast<const unsigned char*>(file_bytes) + w.data_offset, c->staging.p, raw_bytes, c->stream));
    HIP_TRY(launch_decode_pcm(c->staging.p, w.format, w.bytes_per_sample, c->audio.as<float>(), count, c->stream));
    c->has_lo = false;
    c->n_samples = w.n_samples; c->n_channels = w.n_channels; c->n_clips = 1; c->clip_base = 0;
    c->win_total = 0; c->win_offset = 0;
    c->last_fs = w.sampling_frequency;
    return REPET_OK;
}

int repet_ctx_result_wav(repet_ctx* c, int which, int dtype, void* file_out, int64_t capacity, int64_t* n_written) {
    if (!c || !file_out || !n_written) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (which != 1 && which != 2) return fail(REPET_ERR_BAD_ARG, "which must be 1 (background) or 2 (foreground)");
    if (dtype != REPET_F64 && dtype != REPET_F32) return fail(REPET_ERR_BAD_ARG, "float64 or float32 results");
    if (c->last_algo < 0 || c->n_clips != 1) return fail(REPET_ERR_BAD_ARG, "no separation of a single clip has been run on this context");
    if (c->last_fs <= 0) return fail(REPET_ERR_BAD_ARG, "the sampling frequency is unknown: the clip did not come from repet_ctx_upload_wav (set it with repet_ctx_set_sampling_frequency)");
    const int item = dtype == REPET_F64 ? 8 : 4;
    const int64_t count = c->n_samples * c->n_channels;
    const int64_t need = 58 + count * item;
    if (need > (int64_t)0xFFFFFFFF) return fail(REPET_ERR_LIMIT, "result too large for a RIFF/WAVE file (32-bit sizes)");
    if (capacity < need) return fail(REPET_ERR_BAD_ARG, "capacity too small for the file image");
    DeviceGuard guard(c->device);
    unsigned char* out = static_cast<unsigned char*>(file_out);
    const int64_t hdr = wav_float_header(out, c->last_fs, c->n_channels, c->n_samples, item);
    if (count > 0) {
        if (dtype == REPET_F64) {
            // the header is 58 bytes, so the samples are 2-byte aligned in the image: widen into an aligned bounce of the
            // pinned pool and copy (the copy is cheap next to the transfer)
            double* tmp = static_cast<double*>(host_alloc((size_t)count * 8));
            double* dst = tmp ? tmp : static_cast<double*>(malloc((size_t)count * 8));
            if (!dst) return fail(REPET_ERR_OOM, "host memory");
            int rc = REPET_OK;
            if (which == 1) {
                hipError_t e = staged_download(c->ring, c->out.as<float>(), dst, (size_t)count, c->stream);
                if (e != hipSuccess) rc = fail(REPET_ERR_HIP, hipGetErrorString(e));
            } else {
                rc = repet_ctx_download_foreground(c, dst);
            }
            if (rc == REPET_OK) std::memcpy(out + hdr, dst, (size_t)count * 8);
            if (tmp) host_free(tmp); else free(dst);
            if (rc != REPET_OK) return rc;
        } else {
            const float* src = c->out.as<float>();
            if (which == 2) {
                HIP_TRY(c->tmp_a.ensure((size_t)count * sizeof(float)));
                HIP_TRY(launch_foreground_f32(c->audio.as<float>(), c->out.as<float>(), c->tmp_a.as<float>(), count, c->stream));
                src = c->tmp_a.as<float>();
            }
            HIP_TRY(hipMemcpyAsync(out + hdr, src, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    *n_written = need;
    return REPET_OK;
}

int repet_ctx_set_sampling_frequency(repet_ctx* c, int32_t sampling_frequency) {
    if (!c || sampling_frequency <= 0) return fail(REPET_ERR_BAD_ARG, "bad argument");
    c->last_fs = sampling_frequency;
    return REPET_OK;
}

void* repet_host_alloc(size_t bytes) { return host_alloc(bytes); }
void repet_host_free(void* ptr) { host_free(ptr); }

int repet_ctx_download_foreground(repet_ctx* c, double* out) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (c->last_algo < 0) return fail(REPET_ERR_BAD_ARG, "no separation has been run on this context");
    DeviceGuard guard(c->device);
    const int64_t count = c->n_samples * c->n_channels * c->n_clips;
    if (count == 0) return REPET_OK;
    HIP_TRY(c->out64.ensure((size_t)count * sizeof(double)));
    HIP_TRY(launch_foreground(c->audio.as<float>(), c->out.as<float>(), c->out64.as<double>(), count, c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->out64.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int repet_ctx_spectrogram(repet_ctx* c, int which, int32_t window_length, float* out, int64_t n_frames) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (which < 0 || which > 2) return fail(REPET_ERR_BAD_ARG, "which must be 0 (mixture), 1 (background) or 2 (foreground)");
    if (which != 0 && c->last_algo < 0) return fail(REPET_ERR_BAD_ARG, "no separation has been run on this context");
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    DeviceGuard guard(c->device);
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, window_length, &tb));
    const int W = window_length, H = W / 2;
    const int64_t N = c->n_samples, T = repet_frame_count(N, W, H, 1);
    if (T != n_frames) return fail(REPET_ERR_BAD_ARG, "n_frames does not match repet_frame_count");
    const Geo g = make_geo(W, H, T, 1);
    HIP_TRY(c->tmp_a.ensure(std::max<size_t>((size_t)N * sizeof(float), 256)));
    HIP_TRY(launch_channel_mean(c->audio.as<float>(), c->out.as<float>(), which, c->n_channels, c->tmp_a.as<float>(), N, c->stream));
    // the spectra workspaces are reused: a later execute() recomputes them anyway
    HIP_TRY(c->X.ensure((size_t)g.chan_stride * sizeof(float2)));
    HIP_TRY(c->V.ensure((size_t)g.chan_stride * sizeof(float)));
    StftArgs a{};
    a.audio = c->tmp_a.as<float>(); a.n_samples = N; a.n_channels = 1; a.sample_offset = 0;
    a.window = tb->window.as<float>(); a.twiddle = tb->twiddle.as<float2>(); a.W = W; a.H = H; a.T = T; a.FS = g.FS; a.centred = 1;
    a.X = c->X.as<float2>(); a.V = c->V.as<float>(); a.chan_stride = g.chan_stride;
    HIP_TRY(launch_stft(a, c->stream));
    return d2h_pitched(c, out, c->V.as<float>(), g.FS, T, g.F);
}

static int thread_ctx(int device, repet_ctx** out) {
    auto it = g_thread_ctx.by_device.find(device);
    if (it != g_thread_ctx.by_device.end()) { *out = it->second; return REPET_OK; }
    repet_ctx* c = nullptr;
    RP_TRY(repet_ctx_create(device, &c));
    g_thread_ctx.by_device[device] = c;
    *out = c;
    return REPET_OK;
}

int repet_median_network_info(int32_t list_bound, int32_t* network_size, int32_t* instructions) {
    int size = 0;
    const int n = median_network_instructions(list_bound, &size);
    if (network_size) *network_size = size;
    if (instructions) *instructions = n;
    return REPET_OK;
}

int repet_release_thread_ctx(void) {
    for (auto& kv : g_thread_ctx.by_device) repet_ctx_destroy(kv.second);
    g_thread_ctx.by_device.clear();
    return REPET_OK;
}

}  // extern "C"
ThreadContexts::~ThreadContexts() {
    for (auto& kv : by_device) repet_ctx_destroy(kv.second);
}
extern "C" {

int repet_run(int algo, const void* audio, int dtype, int64_t n, int32_t ch, const repet_params* p, double* out,
              int device, repet_timing* timing) {
    repet_ctx* c = nullptr;
    RP_TRY(thread_ctx(device, &c));
    RP_TRY(repet_ctx_upload(c, audio, dtype, n, ch));
    // (without a timing request nothing waits between the last kernel and the first copy back: the download is ordered behind
    // the run on the context's stream, and an error of the run surfaces there)
    if (timing) RP_TRY(repet_ctx_execute(c, algo, p, timing));
    else RP_TRY(repet_ctx_execute_async(c, algo, p));
    return repet_ctx_download(c, out);
}

namespace {

// Logical devices (test switch): REPET_LOGICAL_DEVICES=n lets repet_run_batch deal its clips over n "devices" although
// fewer GPUs are visible -- logical device d runs on physical device d % visible, in its own thread, context and stream.
// Dealing, per-device threads and result placement of the multi-GPU path can then be exercised on a one-GPU box.
// what the last repet_run_batch / repet_run_batch_rccl call of this thread did (repet_last_batch_info)
struct BatchInfo { int64_t transport = 0, clips_sent = 0, clips_with_remainders = 0, groups = 0; };
thread_local BatchInfo g_batch_info;

int logical_device_count(int physical) {
    const char* e = getenv("REPET_LOGICAL_DEVICES");
    const int n = e ? atoi(e) : 0;
    return n > physical ? n : physical;
}

// ---- RCCL over xGMI, inside the library --------------------------------------------------------------
// librccl is opened on first use (dlopen by soname: a process that has PyTorch's RCCL loaded gets that one) -- the library
// carries no link-time dependency on it. One communicator per physical device from ncclCommInitAll, one process.
struct Rccl {
    using comm_t = void*;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*CommAbort)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    static constexpr int kFloat = 7;          // ncclFloat32
    static Rccl& get() {
        static Rccl r = [] {
            Rccl x;
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            auto sym = [&](const char* name) { return dlsym(h, name); };
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
            x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(sym("ncclCommAbort"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
            x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
            x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
            x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
            x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv;
            return x;
        }();
        return r;
    }
};

#define NCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        const int r_ = (expr);                                                                                   \
        if (r_ != 0) return fail(REPET_ERR_HIP, std::string(#expr) + ": " + (rc.GetErrorString ? rc.GetErrorString(r_) : "RCCL error")); \
    } while (0)

// transport 0: every device's worker thread uploads its own clips from the caller's host arrays and downloads its own
// results (with the data in host RAM this uses every device's own PCIe link: SURVEY 8e's "honest comparison").
// transport 1: the clips enter through device 0, travel to their devices as ONE group of ncclSend / ncclRecv over xGMI
// (fp32, interleaved), are separated there from the received device buffers, and the results come back the same way.
int run_batch_impl(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                   const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices, int transport) {
    if (n_clips < 0 || (n_clips > 0 && (!audio || !n_samples || !n_channels || !out)))
        return fail(REPET_ERR_BAD_ARG, "null argument");
    const int physical = repet_device_count();
    if (physical < 1) return fail(REPET_ERR_HIP, "no HIP device");
    const int avail = transport == 1 ? physical : logical_device_count(physical);     // (RCCL needs distinct physical devices)
    if (n_devices < 1 || n_devices > avail) return fail(REPET_ERR_BAD_ARG, "n_devices out of range");
    // longest first, dealt round-robin: clip order[i] -> device i % n_devices
    std::vector<int> order(n_clips);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_samples[a] > n_samples[b]; });
    std::vector<int> device_of(n_clips);
    for (int i = 0; i < n_clips; ++i) device_of[order[i]] = i % n_devices;
    std::vector<int> rcs(n_devices, REPET_OK);
    std::vector<std::string> msgs(n_devices);
    auto run_threads = [&](const std::function<void(int)>& worker) {
        if (n_devices == 1) { worker(0); return; }
        std::vector<std::thread> th;
        for (int d = 0; d < n_devices; ++d) th.emplace_back(worker, d);
        for (auto& t : th) t.join();
    };
    auto first_error = [&]() -> int {
        for (int d = 0; d < n_devices; ++d)
            if (rcs[d] != REPET_OK) return fail(rcs[d], msgs[d]);
        return REPET_OK;
    };

    if (transport != 1) {
        g_batch_info = BatchInfo{};
        run_threads([&](int dev) {
            repet_ctx* c = nullptr;
            int rc = repet_ctx_create(dev % physical, &c);
            for (int i = dev; rc == REPET_OK && i < n_clips; i += n_devices) {
                const int k = order[i];
                rc = repet_ctx_upload(c, audio[k], dtype, n_samples[k], n_channels[k]);
                if (rc == REPET_OK) rc = repet_ctx_execute(c, algo, p, nullptr);
                if (rc == REPET_OK) rc = repet_ctx_download(c, out[k]);
            }
            if (rc != REPET_OK) msgs[dev] = g_last_error;
            rcs[dev] = rc;
            repet_ctx_destroy(c);
        });
        return first_error();
    }

    // ---- transport 1 -------------------------------------------------------------------------------------------------
    // One call at a time (the communicators are shared, cached per device count and kept until the process ends: creating
    // them costs hundreds of milliseconds). The clips are worked through in ROUNDS of one clip per device: while the devices
    // separate round r, the root narrows / uploads round r + 1 and its scatter group is already enqueued; the results of
    // round r return in their own group and their buffers are freed before round r + 2 is staged -- the root never holds more
    // than two rounds. A float64 clip travels as TWO fp32 planes, samples and remainders (x - (double)(float)x, only where
    // one is not zero): the second level of the peak picking then decides on the same 48 bits as the single-GPU call.
    // REPET_RCCL_SELF=1 with n_devices == 1 (test switch): every clip takes the send / receive path, device 0 to itself
    // inside the group, so that the transport's lines run on a one-GPU box.
    Rccl& rc = Rccl::get();
    if (!rc.ok) return fail(REPET_ERR_HIP, "librccl could not be loaded (RCCL transport of repet_run_batch)");
    static std::mutex call_mu;
    static std::map<int, std::vector<Rccl::comm_t>> comm_cache;
    std::lock_guard<std::mutex> call_lock(call_mu);
    const bool self_test = n_devices == 1 && [] { const char* e = getenv("REPET_RCCL_SELF"); return e && e[0] == '1'; }();
    g_batch_info = BatchInfo{};
    g_batch_info.transport = 1;
    auto it = comm_cache.find(n_devices);
    if (it == comm_cache.end()) {
        std::vector<int> devs(n_devices);
        std::iota(devs.begin(), devs.end(), 0);
        std::vector<Rccl::comm_t> fresh(n_devices, nullptr);
        NCCL_TRY(rc.CommInitAll(fresh.data(), n_devices, devs.data()));
        it = comm_cache.emplace(n_devices, std::move(fresh)).first;
    }
    std::vector<Rccl::comm_t>& comms = it->second;
    bool comms_broken = false;

    struct ClipBufs { float *in_root = nullptr, *lo_root = nullptr, *out_root = nullptr, *in_dev = nullptr, *lo_dev = nullptr, *out_dev = nullptr; bool has_lo = false; };
    std::vector<ClipBufs> bufs(n_clips);
    std::vector<repet_ctx*> ctx(n_devices, nullptr);
    repet_ctx* io = nullptr;                          // the root's own context for staging and transport (ctx[0] computes)
    auto travels = [&](int k) { return device_of[k] != 0 || self_test; };
    auto free_clip = [&](int k) {
        ClipBufs& q = bufs[k];
        { DeviceGuard g(0); for (float** ptr : {&q.in_root, &q.lo_root, &q.out_root}) if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; } }
        { DeviceGuard g(device_of[k]); for (float** ptr : {&q.in_dev, &q.lo_dev, &q.out_dev}) if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; } }
    };
    struct Xfer { const float* src; int src_dev; float* dst; int dst_dev; size_t count; };
    // One group of sends and receives. The group is CLOSED whatever happens inside it (an error between ncclGroupStart and
    // ncclGroupEnd used to leave it open under the communicators' destruction); a failed group marks the communicators broken.
    auto exchange = [&](const std::vector<Xfer>& xs) -> int {
        if (xs.empty()) return REPET_OK;
        int err = rc.GroupStart();
        if (err != 0) { comms_broken = true; return fail(REPET_ERR_HIP, std::string("ncclGroupStart: ") + (rc.GetErrorString ? rc.GetErrorString(err) : "RCCL error")); }
        const char* what = nullptr;
        for (const Xfer& x : xs) {
            hipStream_t send_stream = x.src_dev == 0 ? io->stream : ctx[x.src_dev]->stream;
            hipStream_t recv_stream = x.dst_dev == 0 ? io->stream : ctx[x.dst_dev]->stream;
            err = rc.Send(x.src, x.count, Rccl::kFloat, x.dst_dev, comms[x.src_dev], send_stream);
            if (err != 0) { what = "ncclSend"; break; }
            err = rc.Recv(x.dst, x.count, Rccl::kFloat, x.src_dev, comms[x.dst_dev], recv_stream);
            if (err != 0) { what = "ncclRecv"; break; }
        }
        const int end = rc.GroupEnd();
        if (err == 0 && end != 0) { err = end; what = "ncclGroupEnd"; }
        if (err != 0) { comms_broken = true; return fail(REPET_ERR_HIP, std::string(what) + ": " + (rc.GetErrorString ? rc.GetErrorString(err) : "RCCL error")); }
        ++g_batch_info.groups;
        return REPET_OK;
    };
    const int n_rounds = (n_clips + n_devices - 1) / n_devices;
    auto round_clips = [&](int r) { std::vector<int> ks; for (int i = r * n_devices; i < std::min(n_clips, (r + 1) * n_devices); ++i) ks.push_back(order[i]); return ks; };
    // A(r): the round's clips enter through the root (fp32 samples + remainders), the travelling ones leave in one group
    auto stage_round = [&](int r) -> int {
        std::vector<Xfer> xs;
        for (int k : round_clips(r)) {
            ClipBufs& q = bufs[k];
            const size_t count = (size_t)n_samples[k] * n_channels[k];
            const size_t bytes = std::max<size_t>(count * sizeof(float), 256);
            const bool want_lo = dtype == REPET_F64 && count > 0;
            {
                DeviceGuard g(0);
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.in_root), bytes));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.out_root), bytes));
                if (want_lo) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.lo_root), bytes));
                bool not_finite = false;
                HIP_TRY(staged_upload(io->ring, audio[k], dtype, q.in_root, count, io->stream, q.lo_root, &q.has_lo, nullptr, &not_finite));
                if (not_finite) return fail(REPET_ERR_BAD_ARG, "audio_signal contains NaN or infinite samples");
            }
            if (q.has_lo) ++g_batch_info.clips_with_remainders;
            if (!travels(k) || count == 0) continue;
            const int g = device_of[k];
            {
                DeviceGuard gd(g);
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.in_dev), bytes));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.out_dev), bytes));
                if (q.has_lo) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.lo_dev), bytes));
            }
            xs.push_back({q.in_root, 0, q.in_dev, g, count});
            if (q.has_lo) xs.push_back({q.lo_root, 0, q.lo_dev, g, count});
            ++g_batch_info.clips_sent;
        }
        RP_TRY(exchange(xs));
        DeviceGuard g(0);
        HIP_TRY(hipStreamSynchronize(io->stream));          // the root's copies are complete (and the sends have been matched)
        return REPET_OK;
    };
    // B(r): every device separates its clip of the round from device memory; the result stays on the device
    auto compute_clip = [&](int k) -> int {
        ClipBufs& q = bufs[k];
        const int dev = device_of[k];
        const bool moved = travels(k) && (size_t)n_samples[k] * n_channels[k] > 0;
        RP_TRY(repet_ctx_upload_device_split(ctx[dev], moved ? q.in_dev : q.in_root, q.has_lo ? (moved ? q.lo_dev : q.lo_root) : nullptr,
                                             n_samples[k], n_channels[k], 1));
        RP_TRY(repet_ctx_execute(ctx[dev], algo, p, nullptr));
        return repet_ctx_download_device(ctx[dev], moved ? q.out_dev : q.out_root);
    };
    // C(r): the travelling results return in one group; the root widens them into the caller's arrays; the round is freed
    auto finish_round = [&](int r) -> int {
        std::vector<Xfer> xs;
        for (int k : round_clips(r)) {
            const size_t count = (size_t)n_samples[k] * n_channels[k];
            if (travels(k) && count > 0) xs.push_back({bufs[k].out_dev, device_of[k], bufs[k].out_root, 0, count});
        }
        RP_TRY(exchange(xs));
        {
            DeviceGuard g(0);
            for (int k : round_clips(r))
                HIP_TRY(staged_download(io->ring, bufs[k].out_root, out[k], (size_t)n_samples[k] * n_channels[k], io->stream));
            HIP_TRY(hipStreamSynchronize(io->stream));
        }
        for (int d = 1; d < n_devices; ++d) { DeviceGuard g(d); HIP_TRY(hipStreamSynchronize(ctx[d]->stream)); }     // (their sends)
        for (int k : round_clips(r)) free_clip(k);
        return REPET_OK;
    };
    auto body = [&]() -> int {
        RP_TRY(repet_ctx_create(0, &io));
        for (int d = 0; d < n_devices; ++d) RP_TRY(repet_ctx_create(d, &ctx[d]));
        if (n_rounds > 0) RP_TRY(stage_round(0));
        for (int r = 0; r < n_rounds; ++r) {
            // the devices work on round r in their own threads while this one stages round r + 1
            const std::vector<int> ks = round_clips(r);
            std::vector<int> round_rc(ks.size(), REPET_OK);
            std::vector<std::string> round_msg(ks.size());
            std::vector<std::thread> th;
            for (size_t i = 0; i < ks.size(); ++i)
                th.emplace_back([&, i] { round_rc[i] = compute_clip(ks[i]); if (round_rc[i] != REPET_OK) round_msg[i] = g_last_error; });
            const int staged = r + 1 < n_rounds ? stage_round(r + 1) : REPET_OK;
            const std::string staged_msg = g_last_error;
            for (auto& t : th) t.join();
            for (size_t i = 0; i < ks.size(); ++i) if (round_rc[i] != REPET_OK) return fail(round_rc[i], round_msg[i]);
            if (staged != REPET_OK) return fail(staged, staged_msg);
            RP_TRY(finish_round(r));
        }
        return REPET_OK;
    };
    const int status = body();
    const std::string keep = g_last_error;
    for (int d = 0; d < n_devices; ++d) if (ctx[d]) { DeviceGuard g(d); (void)hipStreamSynchronize(ctx[d]->stream); }
    if (io) { DeviceGuard g(0); (void)hipStreamSynchronize(io->stream); }
    for (int k = 0; k < n_clips; ++k) free_clip(k);
    for (int d = 0; d < n_devices; ++d) if (ctx[d]) repet_ctx_destroy(ctx[d]);
    if (io) repet_ctx_destroy(io);
    if (comms_broken) {                               // do not hand a communicator with a failed group to the next call
        for (Rccl::comm_t cm : comms) if (cm) (void)(rc.CommAbort ? rc.CommAbort(cm) : rc.CommDestroy(cm));
        comm_cache.erase(n_devices);
    }
    if (status != REPET_OK) g_last_error = keep;
    return status;
}

}  // namespace

int repet_last_batch_info(int64_t out[4]) {
    if (!out) return fail(REPET_ERR_BAD_ARG, "null argument");
    out[0] = g_batch_info.transport; out[1] = g_batch_info.clips_sent; out[2] = g_batch_info.clips_with_remainders; out[3] = g_batch_info.groups;
    return REPET_OK;
}

int repet_run_batch(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                    const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices) {
    return run_batch_impl(algo, n_clips, audio, dtype, n_samples, n_channels, p, out, n_devices, 0);
}

int repet_run_batch_rccl(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                         const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices) {
    return run_batch_impl(algo, n_clips, audio, dtype, n_samples, n_channels, p, out, n_devices, 1);
}

// ---- stage-level exports ---------------------------------------------------------------------------

int repet_stft(repet_ctx* c, const float* x, int64_t n, const float* window, int32_t W, int32_t H, int32_t centred,
               float* spec_out, int64_t n_frames) {
    if (!c || !x || !window || !spec_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (H < 1) return fail(REPET_ERR_BAD_ARG, "step length must be >= 1");
    DeviceGuard guard(c->device);
    const float2* tw = nullptr;
    RP_TRY(upload_twiddle_only(c, W, &tw));
    const int64_t T = repet_frame_count(n, W, H, centred);
    if (T != n_frames) return fail(REPET_ERR_BAD_ARG, "n_frames does not match repet_frame_count");
    const Geo g = make_geo(W, H, T, 1);
    HIP_TRY(c->tmp_a.ensure(std::max<size_t>((size_t)n * sizeof(float), 256)));
    HIP_TRY(c->tmp_b.ensure((size_t)W * sizeof(float)));
    HIP_TRY(c->X.ensure((size_t)g.chan_stride * sizeof(float2)));
    HIP_TRY(c->V.ensure((size_t)g.chan_stride * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tmp_b.p, window, (size_t)W * sizeof(float), hipMemcpyHostToDevice, c->stream));
    StftArgs a{};
    a.audio = c->tmp_a.as<float>(); a.n_samples = n; a.n_channels = 1; a.sample_offset = 0;
    a.window = c->tmp_b.as<float>(); a.twiddle = tw; a.W = W; a.H = H; a.T = T; a.FS = g.FS; a.centred = centred;
    a.X = c->X.as<float2>(); a.V = c->V.as<float>(); a.chan_stride = g.chan_stride;
    HIP_TRY(launch_stft(a, c->stream));
    if (T > 0)
        HIP_TRY(hipMemcpy2DAsync(spec_out, (size_t)g.F * sizeof(float2), c->X.p, (size_t)g.FS * sizeof(float2),
                                 (size_t)g.F * sizeof(float2), T, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int repet_istft(repet_ctx* c, const float* spec, int64_t T, const float* window, int32_t W, int32_t H, float* y_out,
                int64_t n_out) {
    if (!c || !spec || !window || !y_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (H < 1 || H > W) return fail(REPET_ERR_BAD_ARG, "bad step length");
    DeviceGuard guard(c->device);
    const float2* tw = nullptr;
    RP_TRY(upload_twiddle_only(c, W, &tw));
    const int64_t want = T * H - (W - H);                           // repet.py:1079,1098
    if (n_out != want) return fail(REPET_ERR_BAD_ARG, "n_out must be T*H - (W-H)");
    const Geo g = make_geo(W, H, T, 1);
    HIP_TRY(c->X.ensure((size_t)g.chan_stride * sizeof(float2)));
    HIP_TRY(hipMemsetAsync(c->X.p, 0, (size_t)g.chan_stride * sizeof(float2), c->stream));
    HIP_TRY(hipMemcpy2DAsync(c->X.p, (size_t)g.FS * sizeof(float2), spec, (size_t)g.F * sizeof(float2),
                             (size_t)g.F * sizeof(float2), T, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c->frames.ensure((size_t)T * W * sizeof(float)));
    HIP_TRY(c->tmp_a.ensure(std::max<size_t>((size_t)n_out * sizeof(float), 256)));
    IstftArgs ia{};
    ia.Y = c->X.as<float2>(); ia.chan_stride = g.chan_stride; ia.n_channels = 1; ia.T = T; ia.FS = g.FS; ia.W = W;
    ia.twiddle = tw; ia.frames = c->frames.as<float>();
    HIP_TRY(launch_istft_frames(ia, c->stream));
    double cola = 0;
    for (int i = 0; i < W; i += H) cola += window[i];
    OlaArgs oa{};
    oa.frames = c->frames.as<float>(); oa.n_channels = 1; oa.T = T; oa.W = W; oa.H = H; oa.trim = W - H;
    oa.out = c->tmp_a.as<float>(); oa.n_out = n_out; oa.out_offset = 0; oa.scale = (float)(1.0 / cola);
    HIP_TRY(launch_overlap_add(oa, c->stream));
    HIP_TRY(hipMemcpyAsync(y_out, c->tmp_a.p, (size_t)n_out * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

static int stage_matrix_in(repet_ctx* c, DevBuf& buf, const float* host, int64_t T, int F, int FS, int64_t Tpad) {
    HIP_TRY(buf.ensure((size_t)Tpad * FS * sizeof(float)));
    return h2d_pitched(c, buf.as<float>(), FS, host, T, F, Tpad);
}

int repet_selfsim(repet_ctx* c, const float* v, int64_t T, int32_t F, float* s_out) {
    if (!c || !v || !s_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign);
    const int64_t Tpad = round_up(T, kTile), TS = round_up(T, 64);
    HIP_TRY(c->tmp_a.ensure((size_t)T * F * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, v, (size_t)T * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c->Vn.ensure((size_t)Tpad * FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->Vn.p, 0, (size_t)Tpad * FS * sizeof(float), c->stream));
    HIP_TRY(launch_unit_rows(c->tmp_a.as<float>(), c->Vn.as<float>(), T, F, FS, c->stream));
    HIP_TRY(c->S.ensure((size_t)T * TS * sizeof(float)));
    RP_TRY(run_gram_full(c, c->Vn.as<float>(), T, FS, c->S.as<float>(), TS, true));   // unit rows: same kernel as `sim`
    return d2h_pitched(c, s_out, c->S.as<float>(), TS, T, T);
}

int repet_similarity(repet_ctx* c, const float* a, int64_t TA, const float* b, int64_t TB, int32_t F, float* s_out) {
    if (!c || !a || !b || !s_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (TA < 1 || TB < 1 || F < 1) return fail(REPET_ERR_BAD_ARG, "bad size");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign);
    const int64_t TApad = round_up(TA, kTile), TBpad = round_up(TB, kTile), pitch = round_up(TB, 4);
    HIP_TRY(c->tmp_a.ensure((size_t)std::max(TA, TB) * F * sizeof(float)));
    HIP_TRY(c->Vn.ensure((size_t)TApad * FS * sizeof(float)));
    HIP_TRY(c->P.ensure((size_t)TBpad * FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->Vn.p, 0, (size_t)TApad * FS * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(c->P.p, 0, (size_t)TBpad * FS * sizeof(float), c->stream));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, a, (size_t)TA * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_unit_rows(c->tmp_a.as<float>(), c->Vn.as<float>(), TA, F, FS, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, b, (size_t)TB * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_unit_rows(c->tmp_a.as<float>(), c->P.as<float>(), TB, F, FS, c->stream));
    HIP_TRY(c->S.ensure((size_t)TA * pitch * sizeof(float)));
    HIP_TRY(launch_matmul_nt(c->Vn.as<float>(), TA, c->P.as<float>(), TB, FS, c->S.as<float>(), pitch, c->stream));
    return d2h_pitched(c, s_out, c->S.as<float>(), pitch, TA, TB);
}

int repet_acorr(repet_ctx* c, const float* x, int32_t n_rows, int32_t n_cols, float* ac_out) {
    if (!c || !x || !ac_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n_rows < 1 || n_cols < 1) return fail(REPET_ERR_BAD_ARG, "bad size");
    DeviceGuard guard(c->device);
    const size_t bytes = (size_t)n_rows * n_cols * sizeof(float);
    HIP_TRY(c->tmp_a.ensure(bytes));
    HIP_TRY(c->tmp_c.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, x, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_acorr(c->tmp_a.as<float>(), n_rows, n_cols, n_cols, c->tmp_c.as<float>(), c->stream));
    HIP_TRY(hipMemcpyAsync(ac_out, c->tmp_c.p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int repet_beat_spectrum(repet_ctx* c, const float* p, int64_t T, int32_t F, float* beat_out, int32_t n_lags) {
    if (!c || !p || !beat_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n_lags < 1 || n_lags > T) return fail(REPET_ERR_BAD_ARG, "n_lags must be in [1, T]");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign);
    const int64_t Tpad = round_up(T, kTile);
    const int LP = (int)round_up(n_lags, 64);
    RP_TRY(stage_matrix_in(c, c->P, p, T, F, FS, Tpad));
    HIP_TRY(c->band.ensure((size_t)Tpad * LP * sizeof(float)));
    HIP_TRY(c->beat.ensure((size_t)LP * sizeof(float)));
    RP_TRY(run_gram_band(c, c->P.as<float>(), T, FS, c->band.as<float>(), n_lags, LP));
    RP_TRY(run_band_window_sum(c, c->band.as<float>(), T, LP, n_lags, F, 0, 0, T, 1, c->beat.as<float>(), LP, 1, 0, 0));
    return d2h_pitched(c, beat_out, c->beat.as<float>(), LP, 1, n_lags);
}

int repet_beat_spectrogram(repet_ctx* c, const float* p, int64_t T, int32_t F, int32_t Ls, int32_t Hs, float* beat_out) {
    if (!c || !p || !beat_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (Ls < 1 || Hs < 1) return fail(REPET_ERR_BAD_ARG, "bad segment length/step");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign);
    const int64_t Tpad = round_up(T, kTile);
    const int LP = (int)round_up(Ls, 64);
    const int n_win = (int)ceil_div(T, Hs);
    RP_TRY(stage_matrix_in(c, c->P, p, T, F, FS, Tpad));
    HIP_TRY(c->band.ensure((size_t)Tpad * LP * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->band.p, 0, (size_t)Tpad * LP * sizeof(float), c->stream));
    HIP_TRY(c->beat.ensure((size_t)n_win * LP * sizeof(float)));
    RP_TRY(run_gram_band(c, c->P.as<float>(), T, FS, c->band.as<float>(), Ls, LP));
    const int64_t left = Ls / 2;                                     // ceil((Ls-1)/2)
    RP_TRY(run_band_window_sum(c, c->band.as<float>(), T, LP, Ls, F, -left, Hs, Ls, n_win, c->beat.as<float>(), LP, 1, 0, 0));
    std::vector<float> win((size_t)n_win * Ls);
    RP_TRY(d2h_pitched(c, win.data(), c->beat.as<float>(), LP, n_win, Ls));
    // replicate with the reference's hole (repet.py:1194-1204): frame i+Hs-1 of each step stays zero
    std::memset(beat_out, 0, (size_t)T * Ls * sizeof(float));
    for (int w = 0; w < n_win; ++w) {
        const int64_t i = (int64_t)w * Hs;
        const int64_t end = std::min<int64_t>(i + Hs - 1, T);
        std::memcpy(beat_out + i * Ls, win.data() + (size_t)w * Ls, (size_t)Ls * sizeof(float));
        for (int64_t t = i; t < end; ++t) std::memcpy(beat_out + t * Ls, win.data() + (size_t)w * Ls, (size_t)Ls * sizeof(float));
    }
    return REPET_OK;
}

int repet_periods(repet_ctx* c, const float* beat, int32_t n_cols, int32_t n_lags, int32_t lo, int32_t hi, int32_t* out) {
    if (!c || !beat || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (std::min(hi, n_lags / 3) <= lo) return fail(REPET_ERR_TOO_SHORT, "attempt to get argmax of an empty sequence");
    DeviceGuard guard(c->device);
    HIP_TRY(c->beat.ensure((size_t)n_cols * n_lags * sizeof(float)));
    HIP_TRY(c->periods.ensure((size_t)n_cols * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(c->beat.p, beat, (size_t)n_cols * n_lags * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_periods(c->beat.as<float>(), n_cols, n_lags, n_lags, lo, hi, c->periods.as<int32_t>(), c->stream));
    HIP_TRY(hipMemcpyAsync(out, c->periods.p, (size_t)n_cols * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

int repet_local_maxima(repet_ctx* c, const float* m, int32_t n_rows, int32_t n_cols, float min_value, int32_t d,
                       int32_t number, int32_t* idx_out, int32_t* count_out) {
    if (!c || !m || !idx_out || !count_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n_rows < 1 || n_cols < 1 || number < 1 || d < 0) return fail(REPET_ERR_BAD_ARG, "bad size");
    DeviceGuard guard(c->device);
    const int64_t pitch = round_up(n_cols, 4);
    HIP_TRY(c->S.ensure((size_t)n_rows * pitch * sizeof(float)));
    RP_TRY(h2d_pitched(c, c->S.as<float>(), pitch, m, n_rows, n_cols, n_rows));
    HIP_TRY(c->idx.ensure((size_t)n_rows * number * sizeof(int32_t)));
    HIP_TRY(c->cnt.ensure((size_t)n_rows * sizeof(int32_t)));
    float* seg = nullptr;
    const int seg_pitch = segment_pitch((int)pitch);
    if (local_maxima_segments_apply(n_cols, d, pitch, 0, 1)) {
        HIP_TRY(c->seg.ensure((size_t)n_rows * 3 * seg_pitch * sizeof(float)));
        seg = c->seg.as<float>();
        HIP_TRY(launch_segment_maxima(c->S.as<float>(), n_rows, n_cols, pitch, seg, seg_pitch, c->stream));
    }
    hipError_t e = launch_local_maxima(c->S.as<float>(), n_rows, 0, n_cols, pitch, 0, min_value, d, number,
                                       c->idx.as<int32_t>(), number, c->cnt.as<int32_t>(), c->stream, 0, nullptr, nullptr, nullptr,
                                       nullptr, seg, seg_pitch);
    if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "row too long for the peak-picking kernel");
    HIP_TRY(e);
    HIP_TRY(hipMemcpyAsync(idx_out, c->idx.p, (size_t)n_rows * number * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(count_out, c->cnt.p, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
}

static int stage_mask_common(repet_ctx* c, const float* v, int64_t T, int F, MaskArgs* m, int* FS_out) {
    const int FS = (int)round_up(F, kFreqAlign);
    RP_TRY(stage_matrix_in(c, c->V, v, T, F, FS, T + kPadRows));
    HIP_TRY(launch_fill_pad_rows(c->V.as<float>(), (T + kPadRows) * FS, 1, T, FS, c->stream));
    HIP_TRY(c->tmp_c.ensure((size_t)T * FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->tmp_c.p, 0, (size_t)T * FS * sizeof(float), c->stream));
    *m = MaskArgs{};
    m->V = c->V.as<float>(); m->chan_stride = (T + kPadRows) * FS; m->n_channels = 1; m->T = T; m->F = F; m->FS = FS;
    m->X = nullptr; m->mask = c->tmp_c.as<float>(); m->cutoff = 0; m->pad_row = T;
    *FS_out = FS;
    return REPET_OK;
}

int repet_mask_period(repet_ctx* c, const float* v, int64_t T, int32_t F, int32_t period, float* mask_out) {
    if (!c || !v || !mask_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (period < 1) return fail(REPET_ERR_BAD_ARG, "period must be >= 1");
    DeviceGuard guard(c->device);
    MaskArgs m; int FS;
    RP_TRY(stage_mask_common(c, v, T, F, &m, &FS));
    HIP_TRY(launch_mask_period(m, nullptr, period, period, c->stream));
    return d2h_pitched(c, mask_out, c->tmp_c.as<float>(), FS, T, F);
}

int repet_mask_adaptive(repet_ctx* c, const float* v, int64_t T, int32_t F, const int32_t* periods, int32_t order,
                        float* mask_out) {
    if (!c || !v || !periods || !mask_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (order < 1) return fail(REPET_ERR_BAD_ARG, "filter_order must be >= 1");
    DeviceGuard guard(c->device);
    MaskArgs m; int FS;
    RP_TRY(stage_mask_common(c, v, T, F, &m, &FS));
    HIP_TRY(c->periods.ensure((size_t)T * sizeof(int32_t)));
    HIP_TRY(hipMemcpyAsync(c->periods.p, periods, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_mask_adaptive(m, c->periods.as<int32_t>(), order, c->stream));
    return d2h_pitched(c, mask_out, c->tmp_c.as<float>(), FS, T, F);
}

int repet_mask_sim(repet_ctx* c, const float* v, int64_t T, int32_t F, const int32_t* idx, const int32_t* count,
                   int32_t number, float* mask_out) {
    if (!c || !v || !idx || !count || !mask_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    MaskArgs m; int FS;
    RP_TRY(stage_mask_common(c, v, T, F, &m, &FS));
    const int KP = std::max(number, kMinIdxPitch);
    HIP_TRY(c->idx.ensure((size_t)T * KP * sizeof(int32_t)));
    HIP_TRY(c->cnt.ensure((size_t)T * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->idx.p, 0, (size_t)T * KP * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemcpy2DAsync(c->idx.p, (size_t)KP * sizeof(int32_t), idx, (size_t)number * sizeof(int32_t),
                             (size_t)number * sizeof(int32_t), T, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->cnt.p, count, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, number, c->stream));
    return d2h_pitched(c, mask_out, c->tmp_c.as<float>(), FS, T, F);
}

int repet_rank_columns(repet_ctx* c, const float* v, int64_t T, int32_t F, uint16_t* codes_out, float* sorted_out) {
    if (!c || !v || !codes_out || !sorted_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (F < 128) return fail(REPET_ERR_BAD_ARG, "needs at least 128 bins");
    if (!rank_columns_supported(T)) return fail(REPET_ERR_LIMIT, "rank transform: 1024 < n_frames <= 30720");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign), n_cols = F & ~127;
    const int64_t rows = T + kPadRows, vs_pitch = round_up(T, 32);
    RP_TRY(stage_matrix_in(c, c->V, v, T, F, FS, rows));
    HIP_TRY(c->R.ensure((size_t)rows * FS * sizeof(unsigned short)));
    c->r_pads_ptr = nullptr;                              // this export lays R out differently
    HIP_TRY(c->Vs.ensure((size_t)n_cols * vs_pitch * sizeof(float)));
    RankArgs a{};
    a.V = c->V.as<float>(); a.chan_stride = rows * FS; a.n_channels = 1; a.T = T; a.FS = FS; a.n_cols = n_cols;
    a.R = c->R.as<unsigned short>(); a.r_chan_stride = rows * FS; a.Vs = c->Vs.as<float>(); a.vs_pitch = vs_pitch;
    HIP_TRY(c->rank_codes.ensure((size_t)n_cols * vs_pitch * sizeof(unsigned short)));
    a.codes = c->rank_codes.as<unsigned short>();
    HIP_TRY(launch_rank_columns(a, c->stream));
    HIP_TRY(hipMemcpy2DAsync(codes_out, (size_t)n_cols * sizeof(uint16_t), c->R.p, (size_t)FS * sizeof(uint16_t),
                             (size_t)n_cols * sizeof(uint16_t), T, hipMemcpyDeviceToHost, c->stream));
    return d2h_pitched(c, sorted_out, c->Vs.as<float>(), vs_pitch, n_cols, T);
}

int repet_ctx_last_periods(repet_ctx* c, int32_t* out, int32_t capacity, int32_t* n_written) {
    if (!c || !out || !n_written) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    const int n = std::min(capacity, c->last_n_periods);
    if (n > 0) HIP_TRY(hipMemcpy(out, c->periods.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    *n_written = n;
    return REPET_OK;
}

int repet_ctx_last_sim_indices(repet_ctx* c, int32_t* idx_out, int32_t* count_out, int32_t n_rows, int32_t number) {
    if (!c || !idx_out || !count_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    // a batch context holds the lists of its clips back to back: n_rows may be rows-per-clip (first clip) or all of them
    if ((n_rows != c->last_idx_rows && n_rows != c->last_idx_rows * c->last_idx_batch) || number != c->last_idx_number)
        return fail(REPET_ERR_BAD_ARG, "shape does not match the last run");
    DeviceGuard guard(c->device);
    if (n_rows > 0) {
        HIP_TRY(hipMemcpy2D(idx_out, (size_t)number * sizeof(int32_t), c->idx.p, (size_t)c->last_idx_pitch * sizeof(int32_t),
                            (size_t)number * sizeof(int32_t), n_rows, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(count_out, c->cnt.p, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return REPET_OK;
}

int repet_ctx_last_frame_count(repet_ctx* c, int64_t* n_frames) {
    if (!c || !n_frames) return fail(REPET_ERR_BAD_ARG, "null argument");
    *n_frames = c->last_T;
    return REPET_OK;
}

// the counters of the last run, the copies of every diagnostic counter added up ([8] is a maximum) -- common.h, kStatShards
static int read_stats(repet_ctx* c, unsigned int (&total)[kRefineStats]) {
    std::vector<unsigned int> words(kStatWords);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(words.data(), c->refine_stats.p, kStatWords * sizeof(unsigned int), hipMemcpyDeviceToHost));
    for (int k = 0; k < kRefineStats; ++k) total[k] = words[k];
    for (int sh = 1; sh <= kStatShards; ++sh)
        for (int k = 0; k < kRefineStats; ++k)
            total[k] = (k == 8) ? std::max(total[k], words[sh * kRefineStats + k]) : total[k] + words[sh * kRefineStats + k];
    return REPET_OK;
}

int repet_ctx_last_exact_stats(repet_ctx* c, int64_t out[8]) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    for (int k = 0; k < 8; ++k) out[k] = 0;
    out[5] = c->has_lo ? 1 : 0;
    if (!c->refine_stats.p) return REPET_OK;
    DeviceGuard guard(c->device);
    unsigned int host[kRefineStats] = {};
    RP_TRY(read_stats(c, host));
    out[0] = host[4] + host[12] - host[14]; out[1] = host[6]; out[2] = host[7]; out[3] = host[8]; out[4] = host[9];
    out[6] = host[12]; out[7] = host[14];
    return REPET_OK;
}

#ifdef REPET_EXACT_STAMPS
int repet_debug_exact_phases(repet_ctx* c, int64_t out[6]) {
    unsigned int host[kRefineStats] = {};
    RP_TRY(read_stats(c, host));
    for (int k = 0; k < 6; ++k) out[k] = host[24 + k];
    return REPET_OK;
}
#endif

int repet_ctx_last_refine_stats(repet_ctx* c, int64_t out[4]) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    for (int k = 0; k < 4; ++k) out[k] = 0;
    if (!c->refine_stats.p) return REPET_OK;
    DeviceGuard guard(c->device);
    unsigned int host[kRefineStats] = {};
    RP_TRY(read_stats(c, host));
    for (int k = 0; k < 4; ++k) out[k] = host[k];
    return REPET_OK;
}

}  // extern "C"

// =====================================================================================================
// Streaming online REPET-SIM: the reference's "online" variant needs the whole signal
// (repet.py:712-911); this handle accepts audio in arbitrary chunks and returns each hop of background as
// soon as its frame has been seen, with the same kernels as the offline path, so the concatenated output is
// bit-identical to repet.simonline of the whole signal. Device state: a sliding window of the last B-1
// frames (magnitudes, unit rows, the last masked spectrum for the overlap-add tail) plus the unconsumed
// samples; every push processes all newly complete frames in one batch of launches.
// =====================================================================================================
struct repet_online {
    repet_ctx* ctx = nullptr;       // stream, tables, tile cache, scratch buffers
    repet_params p{};
    int C = 0, W = 0, H = 0, F = 0, FS = 0, B = 0, Hh = 0, LP = 0;
    DevBuf X[2], V[2], Vn[2], pend[2], pend_lo[2], band, outf, out64, staging;
    int cur = 0, pcur = 0;
    // the pending buffers start with `pend_hist` samples of HISTORY (already transformed: the frames of the sliding window,
    // whose float64 spectra the second level of the peak picking may ask for) followed by the pend_count unconsumed ones;
    // pend_lo: the fp32 remainders of float64 pushes, sample for sample
    int64_t pend_hist = 0;
    int64_t rows_cap = 0;           // frame rows per channel plane of the windows (without the 8 pad rows)
    int64_t pend_cap = 0, pend_count = 0;   // samples per channel
    int64_t hist_valid = 0;         // valid history rows, right-aligned at row Hh
    int64_t frames_done = 0, total_in = 0, emitted = 0;
    bool finished = false;
};

namespace {

int online_ensure_windows(repet_online* o, int64_t n_new) {
    repet_ctx* c = o->ctx;
    const int64_t need = round_up(o->Hh + n_new, kTile) + kTile;
    if (need <= o->rows_cap) return REPET_OK;
    const int64_t new_cap = std::max(need, 2 * o->rows_cap);
    HIP_TRY(hipStreamSynchronize(c->stream));
    DevBuf nx, nv, nvn;
    const size_t plane = (size_t)(new_cap + kPadRows) * o->FS;
    HIP_TRY(nx.ensure(plane * o->C * sizeof(float2)));
    HIP_TRY(nv.ensure(plane * o->C * sizeof(float)));
    HIP_TRY(nvn.ensure((size_t)new_cap * o->FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(nx.p, 0, plane * o->C * sizeof(float2), c->stream));
    HIP_TRY(hipMemsetAsync(nv.p, 0, plane * o->C * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(nvn.p, 0, (size_t)new_cap * o->FS * sizeof(float), c->stream));
    if (o->hist_valid > 0) {        // carry the history (rows [Hh - hist_valid, Hh)) into the bigger window
        const int64_t r0 = o->Hh - o->hist_valid;
        const size_t old_plane = (size_t)(o->rows_cap + kPadRows) * o->FS;
        HIP_TRY(hipMemcpyAsync(nvn.as<float>() + r0 * o->FS, o->Vn[o->cur].as<float>() + r0 * o->FS,
                               (size_t)o->hist_valid * o->FS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        for (int ch = 0; ch < o->C; ++ch) {
            HIP_TRY(hipMemcpyAsync(nv.as<float>() + ch * plane + r0 * o->FS, o->V[o->cur].as<float>() + ch * old_plane + r0 * o->FS,
                                   (size_t)o->hist_valid * o->FS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(nx.as<float2>() + ch * plane + r0 * o->FS, o->X[o->cur].as<float2>() + ch * old_plane + r0 * o->FS,
                                   (size_t)o->hist_valid * o->FS * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    o->X[o->cur].release(); o->V[o->cur].release(); o->Vn[o->cur].release();
    o->X[o->cur] = nx; o->V[o->cur] = nv; o->Vn[o->cur] = nvn;
    // the other window is only ever written after being (re)initialised below
    o->X[o->cur ^ 1].release(); o->V[o->cur ^ 1].release(); o->Vn[o->cur ^ 1].release();
    HIP_TRY(o->X[o->cur ^ 1].ensure(plane * o->C * sizeof(float2)));
    HIP_TRY(o->V[o->cur ^ 1].ensure(plane * o->C * sizeof(float)));
    HIP_TRY(o->Vn[o->cur ^ 1].ensure((size_t)new_cap * o->FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(o->X[o->cur ^ 1].p, 0, plane * o->C * sizeof(float2), c->stream));
    HIP_TRY(hipMemsetAsync(o->V[o->cur ^ 1].p, 0, plane * o->C * sizeof(float), c->stream));
    HIP_TRY(hipMemsetAsync(o->Vn[o->cur ^ 1].p, 0, (size_t)new_cap * o->FS * sizeof(float), c->stream));
    o->rows_cap = new_cap;
    for (int k = 0; k < 2; ++k)
        HIP_TRY(launch_fill_pad_rows(o->V[k].as<float>(), (new_cap + kPadRows) * o->FS, o->C, new_cap, o->FS, c->stream));
    return REPET_OK;
}

// Process n_new frames starting at global frame o->frames_done (the samples are at the front of the pending
// buffer; samples past pend_count read as zero) and write `n_emit` output samples per channel, starting at
// the first sample of hop frames_done, to out (float64, interleaved).
int online_process(repet_online* o, int64_t n_new, int64_t n_emit, double* out) {
    repet_ctx* c = o->ctx;
    if (n_new <= 0 && n_emit <= 0) {
        HIP_TRY(hipStreamSynchronize(c->stream));      // the caller's chunk has been copied
        return REPET_OK;
    }
    Tables* tb = nullptr;
    RP_TRY(get_tables(c, o->W, &tb));
    RP_TRY(online_ensure_windows(o, n_new));
    const int64_t plane = (o->rows_cap + kPadRows) * o->FS;      // chan_stride of X and V
    const int64_t r0 = o->Hh - o->hist_valid;                    // first valid window row
    const int64_t Tw = o->hist_valid + n_new;                    // valid rows (history + new), relative to r0
    float2* Xb = o->X[o->cur].as<float2>() + r0 * o->FS;
    float* Vb = o->V[o->cur].as<float>() + r0 * o->FS;
    float* Vnb = o->Vn[o->cur].as<float>() + r0 * o->FS;
    const int64_t first_global = o->frames_done - o->hist_valid; // global frame number of window row r0

    if (n_new > 0) {
        StftArgs a{};
        a.audio = o->pend[o->pcur].as<float>(); a.n_samples = o->pend_count; a.n_channels = o->C; a.sample_offset = o->pend_hist;
        a.window = tb->window.as<float>(); a.twiddle = tb->twiddle.as<float2>();
        a.W = o->W; a.H = o->H; a.T = n_new; a.FS = o->FS; a.centred = 0;
        a.X = Xb + o->hist_valid * o->FS; a.V = Vb + o->hist_valid * o->FS; a.chan_stride = plane;
        a.Vn = Vnb + o->hist_valid * o->FS;
        HIP_TRY(launch_stft(a, c->stream));
        // rows behind the new frames up to the next tile boundary must read as zero for the Gram tiles
        const int64_t Tpad = round_up(Tw, kTile);
        HIP_TRY(hipMemsetAsync(Vnb + Tw * o->FS, 0, (size_t)(Tpad - Tw) * o->FS * sizeof(float), c->stream));

        const int64_t first_active = std::max<int64_t>(o->frames_done, o->B - 1);     // global frame number
        const int64_t n_active = o->frames_done + n_new - first_active;
        const int K = o->p.sim_number, KP = std::max(K, kMinIdxPitch);
        if (n_active > 0) {
            HIP_TRY(o->band.ensure((size_t)Tpad * o->LP * sizeof(float)));
            RP_TRY(run_gram_band(c, Vnb, Tw, o->FS, o->band.as<float>(), o->B, o->LP, true, 1, 0, 0, false, true));
            const int peak_mode = c->band_lookback ? 2 : 1;
            HIP_TRY(c->idx.ensure((size_t)n_active * KP * sizeof(int32_t)));
            HIP_TRY(c->cnt.ensure((size_t)n_active * sizeof(int32_t)));
            PeakRefine rf{};
            RP_TRY(make_refine(c, Vnb, o->FS, o->p.sim_threshold, &rf, n_active, 1, o->B, o->p.sim_distance_frames, Tpad));
            hipError_t e = launch_local_maxima(o->band.as<float>(), n_active, first_active, o->B, o->LP, peak_mode, (float)o->p.sim_threshold,
                                               o->p.sim_distance_frames, K, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), c->stream,
                                               first_global, &rf);
            if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "online: buffer too long for the peak-picking kernel");
            HIP_TRY(e);
            // second level: window row fr is global frame first_global + fr, whose first sample sits hist_valid - fr hops
            // before the pending ones in the buffer (zero beyond what has been pushed, as in the offline run's last frame)
            const Geo go = make_geo(o->W, o->H, Tw, o->C);
            RP_TRY(run_exact_rows(c, tb, go, o->band.as<float>(), first_active, o->B, o->LP, peak_mode, (float)o->p.sim_threshold,
                                  o->p.sim_distance_frames, K, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), first_global, rf, nullptr,
                                  o->pend[o->pcur].as<float>(), o->pend_lo[o->pcur].as<float>(), o->pend_hist + o->pend_count, 0,
                                  o->pend_hist - o->hist_valid * (int64_t)o->H, Tpad, 1));
        }
        MaskArgs m{};
        m.V = Vb; m.chan_stride = plane; m.n_channels = o->C; m.T = Tw; m.F = o->F; m.FS = o->FS; m.X = Xb; m.mask = nullptr;
        m.cutoff = o->p.cutoff_bins; m.pad_row = o->rows_cap - r0; m.n_batch = 1; m.batch_stride = 0; m.frame0 = o->hist_valid;
        const int64_t first_frame = Tw - std::max<int64_t>(n_active, 0);          // warm-up rows before it are zeroed
        const int max_peaks = (int)std::min<int64_t>(K, ceil_div(o->B, o->p.sim_distance_frames + 1));
        HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), first_frame, max_peaks, c->stream,
                                c->side_stream, c->fork_event, c->join_event));
    }
    if (n_emit > 0) {
        HIP_TRY(o->outf.ensure((size_t)n_emit * o->C * sizeof(float)));
        HIP_TRY(o->out64.ensure((size_t)n_emit * o->C * sizeof(double)));
        IstftOlaArgs a{};
        a.Y = Xb; a.chan_stride = plane; a.n_channels = o->C; a.T = Tw; a.FS = o->FS; a.W = o->W;
        a.twiddle = tb->twiddle.as<float2>(); a.trim = o->hist_valid * (int64_t)o->H; a.out = o->outf.as<float>();
        a.n_out = n_emit; a.out_offset = 0; a.scale = (float)(1.0 / tb->cola);
        hipError_t e = launch_istft_ola(a, c->stream);
        if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "too many channels for the fused inverse STFT");
        HIP_TRY(e);
        HIP_TRY(launch_convert_out(o->outf.as<float>(), o->out64.as<double>(), n_emit * o->C, c->stream));
        HIP_TRY(hipMemcpyAsync(out, o->out64.p, (size_t)n_emit * o->C * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (n_new > 0) {
        // slide: the last min(Hh, Tw) rows become the history of the other window; drop the consumed samples
        const int64_t h2 = std::min<int64_t>(o->Hh, Tw);
        const int nxt = o->cur ^ 1;
        const int64_t src = r0 + Tw - h2, dst = o->Hh - h2;
        HIP_TRY(hipMemcpyAsync(o->Vn[nxt].as<float>() + dst * o->FS, o->Vn[o->cur].as<float>() + src * o->FS,
                               (size_t)h2 * o->FS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        for (int ch = 0; ch < o->C; ++ch) {
            HIP_TRY(hipMemcpyAsync(o->V[nxt].as<float>() + ch * plane + dst * o->FS, o->V[o->cur].as<float>() + ch * plane + src * o->FS,
                                   (size_t)h2 * o->FS * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            // only the last masked spectrum is needed again (overlap-add tail of the next hop)
            HIP_TRY(hipMemcpyAsync(o->X[nxt].as<float2>() + ch * plane + (o->Hh - 1) * o->FS,
                                   o->X[o->cur].as<float2>() + ch * plane + (r0 + Tw - 1) * o->FS,
                                   (size_t)o->FS * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
        }
        o->cur = nxt;
        o->hist_valid = h2;
        const int64_t consumed = std::min<int64_t>(n_new * (int64_t)o->H, o->pend_count);
        const int64_t left = o->pend_count - consumed;
        // the samples of the window's frames stay in front of the unconsumed ones (h2 hops of history)
        const int64_t keep = std::min<int64_t>(h2 * (int64_t)o->H, o->pend_hist + consumed);
        const int64_t from = o->pend_hist + consumed - keep;
        if (keep + left > 0) {
            HIP_TRY(hipMemcpyAsync(o->pend[o->pcur ^ 1].p, o->pend[o->pcur].as<float>() + from * o->C,
                                   (size_t)(keep + left) * o->C * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(o->pend_lo[o->pcur ^ 1].p, o->pend_lo[o->pcur].as<float>() + from * o->C,
                                   (size_t)(keep + left) * o->C * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        o->pcur ^= 1;
        o->pend_hist = keep;
        o->pend_count = left;
        o->frames_done += n_new;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    o->emitted += n_emit;
    return REPET_OK;
}

}  // namespace

extern "C" {

int repet_online_open(int device, int32_t n_channels, const repet_params* p, repet_online** out) {
    if (!out) return fail(REPET_ERR_BAD_ARG, "out is null");
    RP_TRY(check_params(p));
    if (n_channels < 1) return fail(REPET_ERR_BAD_ARG, "online: at least one channel");
    if (p->buffer_frames < 2 || p->sim_number < 1) return fail(REPET_ERR_BAD_ARG, "online: bad buffer length or similarity number");
    auto* o = new repet_online();
    int rc = repet_ctx_create(device, &o->ctx);
    if (rc != REPET_OK) { delete o; return rc; }
    o->p = *p; o->C = n_channels; o->W = p->window_length; o->H = p->step_length; o->F = o->W / 2 + 1;
    o->FS = (int)round_up(o->F, kFreqAlign); o->B = p->buffer_frames; o->Hh = o->B - 1; o->LP = (int)round_up(o->B, 64);
    *out = o;
    return REPET_OK;
}

int repet_online_close(repet_online* o) {
    if (!o) return REPET_OK;
    {
        DeviceGuard guard(o->ctx->device);
        (void)hipStreamSynchronize(o->ctx->stream);
        for (int k = 0; k < 2; ++k) { o->X[k].release(); o->V[k].release(); o->Vn[k].release(); o->pend[k].release(); o->pend_lo[k].release(); }
        o->band.release(); o->outf.release(); o->out64.release(); o->staging.release();
    }
    repet_ctx_destroy(o->ctx);
    delete o;
    return REPET_OK;
}

int repet_online_push(repet_online* o, const void* audio, int dtype, int64_t n, double* out, int64_t capacity,
                      int64_t* n_written) {
    if (!o || !n_written || (n > 0 && !audio)) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (o->finished) return fail(REPET_ERR_BAD_ARG, "online: stream already finished");
    if (n < 0 || dtype < REPET_F32 || dtype > REPET_I16) return fail(REPET_ERR_BAD_ARG, "bad size or dtype");
    repet_ctx* c = o->ctx;
    DeviceGuard guard(c->device);
    *n_written = 0;
    const int64_t total = o->total_in + n;
    const int64_t full = total >= o->W ? (total - o->W) / o->H + 1 : 0;          // frames completely covered
    const int64_t n_new = std::max<int64_t>(full - o->frames_done, 0);
    const int64_t n_emit = n_new * (int64_t)o->H;
    if (n_emit > capacity || (n_emit > 0 && !out)) return fail(REPET_ERR_BAD_ARG, "online: output capacity too small (needs n_samples + window_length)");
    // append the new samples to the pending buffer (fp32, interleaved)
    const int64_t need = o->pend_hist + o->pend_count + n;
    if (need > o->pend_cap) {
        const int64_t cap = std::max<int64_t>(need + o->W + (int64_t)o->Hh * o->H, 2 * o->pend_cap);
        DevBuf a, b, al, bl;
        HIP_TRY(a.ensure((size_t)cap * o->C * sizeof(float)));
        HIP_TRY(b.ensure((size_t)cap * o->C * sizeof(float)));
        HIP_TRY(al.ensure((size_t)cap * o->C * sizeof(float)));
        HIP_TRY(bl.ensure((size_t)cap * o->C * sizeof(float)));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const size_t live = (size_t)(o->pend_hist + o->pend_count) * o->C * sizeof(float);
        if (live > 0) {
            HIP_TRY(hipMemcpy(a.p, o->pend[o->pcur].p, live, hipMemcpyDeviceToDevice));
            HIP_TRY(hipMemcpy(al.p, o->pend_lo[o->pcur].p, live, hipMemcpyDeviceToDevice));
        }
        o->pend[0].release(); o->pend[1].release(); o->pend_lo[0].release(); o->pend_lo[1].release();
        o->pend[0] = a; o->pend[1] = b; o->pend_lo[0] = al; o->pend_lo[1] = bl; o->pcur = 0; o->pend_cap = cap;
    }
    if (n > 0) {
        const int64_t at = (o->pend_hist + o->pend_count) * o->C;
        float* dst = o->pend[o->pcur].as<float>() + at;
        float* dst_lo = o->pend_lo[o->pcur].as<float>() + at;
        const size_t esz = dtype == REPET_F64 ? 8 : (dtype == REPET_F32 ? 4 : 2);
        if (dtype == REPET_F32) {
            HIP_TRY(hipMemcpyAsync(dst, audio, (size_t)n * o->C * esz, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemsetAsync(dst_lo, 0, (size_t)n * o->C * sizeof(float), c->stream));
        } else {
            HIP_TRY(o->staging.ensure((size_t)n * o->C * esz));
            HIP_TRY(hipMemcpyAsync(o->staging.p, audio, (size_t)n * o->C * esz, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(launch_convert_in(o->staging.p, dtype, dst, n * o->C, c->stream, dst_lo));
        }
        o->pend_count += n;
        o->total_in = total;
    }
    RP_TRY(online_process(o, n_new, n_emit, out));
    *n_written = n_emit;
    return REPET_OK;
}

int repet_online_finish(repet_online* o, double* out, int64_t capacity, int64_t* n_written) {
    if (!o || !n_written) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (o->finished) return fail(REPET_ERR_BAD_ARG, "online: stream already finished");
    repet_ctx* c = o->ctx;
    DeviceGuard guard(c->device);
    *n_written = 0;
    const int64_t N = o->total_in;
    if (N < (int64_t)(o->B - 2) * o->H + o->W)      // the reference's warm-up needs B-1 whole frames (repet.py:795-810)
        return fail(REPET_ERR_TOO_SHORT, "operands could not be broadcast together (signal shorter than the buffer)");
    const int64_t T = repet_frame_count(N, o->W, o->H, 0);                       // repet.py:781, last frame zero-padded
    const int64_t n_new = std::max<int64_t>(T - o->frames_done, 0);
    const int64_t n_emit = N - o->emitted;                                       // truncate to the samples pushed
    if (n_emit > capacity || (n_emit > 0 && !out)) return fail(REPET_ERR_BAD_ARG, "online: output capacity too small");
    RP_TRY(online_process(o, n_new, n_emit, out));
    *n_written = n_emit;
    o->finished = true;
    return REPET_OK;
}

}  // extern "C"
