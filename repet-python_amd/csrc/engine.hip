// Host orchestration and C ABI (include/repet_hip.h) of the gfx950 REPET engine.
//
// A repet_ctx owns one HIP stream, grow-only device workspaces and the per-window-length tables
// (periodic Hamming window, FFT twiddles). repet_ctx_execute chains the kernels of one variant on
// that stream with no host round trip in between (periods and index lists stay on the device).
#include "engine.h"

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
using namespace repet_eng;

namespace repet_eng {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}




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
int run_gram_full(repet_ctx* c, const float* A, int64_t T, int FS, float* S, int64_t TS, bool unit_rows,
                  bool planes_ready, float* seg, int seg_pitch, bool* seg_written) {
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
int run_gram_band(repet_ctx* c, const float* A, int64_t T, int FS, float* band, int n_lags, int LP, bool unit_rows,
                  int B, int64_t a_stride, int64_t band_stride, bool planes_ready, bool lookback) {
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
// Round 6, measured again with the register STFT kernel (a lane writes the two halves of ITS sixteen components: profiles/
// r06_split_in_stft_ab.txt): single clip, cfg 2: STFT 0.0734 -> 0.0770 ms, Gram stage (which held the split pass) 0.2038 ->
// 0.1927: step 0.851 -> 0.843. So: always. REPET_SPLIT_IN_STFT=0: the separate pass for single clips (A/B).
bool split_in_stft(int B) {
    static const bool single_too = [] { const char* e = getenv("REPET_SPLIT_IN_STFT"); return !(e && e[0] == '0'); }();
    return (B > 1 || single_too) && gram_f16_enabled();
}

// The mask kernels read V, read X and write X: 20 bytes per cell, and the inverse STFT reads X again. With the mask as a
// plane of its own they write 4 bytes and the inverse STFT multiplies while it fetches (8 + 4): 20 instead of 28 bytes per
// cell over the two stages, the same products bit for bit (mul_rounded). The inverse kernel pays for its extra loads
// about what the byte count says, so it depends on the mask kernel whether the sum gains (mask + inverse, ms, same box):
//   extended cfg 3 (period mask, HBM-bound)      0.415 + 0.461 -> 0.249 + 0.498   default: plane
//   simonline cfg 5 (ten similar frames: HBM)    0.763 + 0.544 -> 0.603 + 0.608   default: plane
//   adaptive cfg 4                                0.077 + 0.055 -> 0.060 + 0.068   default: in place (round 6: plane, 0.0775 + 0.0494 -> 0.0529 + 0.0573)
//   sim cfg 2 (selection-bound mask)              0.50 + 0.072 -> 0.50 + 0.091     default: in place (round 6: plane, 0.1045 + 0.0533 -> 0.0855 + 0.0661)
// REPET_MASK_PLANE=0 / 1 / p: never / in every variant (with the repeating-segment model where a variant has one) / the same
// as a plain plane, without the model.
int mask_plane_forced() {
    static const int forced = [] { const char* e = getenv("REPET_MASK_PLANE"); return e ? (e[0] == '0' ? 0 : (e[0] == 'p' ? 2 : 1)) : -1; }();
    return forced;
}
bool mask_plane_wanted(MaskKind kind) {
    // (round 6: `adaptive` too -- mask 0.0775 -> 0.0529, inverse 0.0494 -> 0.0573, step 0.262 -> 0.246 ms at cfg 4; and `sim` on rank
    // codes where the register inverse kernel applies: exec_sim. profiles/r06_mask_plane_ab.txt)
    return mask_plane_forced() >= 0 ? mask_plane_forced() != 0 : (kind == MaskKind::period || kind == MaskKind::sim_float || kind == MaskKind::adaptive);
}


// p_planes: the forward STFT will write the row-scaled f16 planes of the power spectra (prepare_power_planes): their pad
// rows [T, Tpad) of every clip are zeroed by the same housekeeping launch (as a 2-D memset they were 61 us at cfg 3)
int ensure_spectra(repet_ctx* c, const Geo& g, bool want_vn, bool want_p, int B, bool p_planes) {
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
             int B, int64_t batch_sample_stride, bool p_as_planes) {
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
    if (c->nonfinite_passes()) HIP_TRY(launch_infinite_frames_fix(a, c->stream));    // (strict reference mode only: see the kernel)
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
void apply_model(IstftOlaArgs& a, repet_ctx*, const ModelRef* mr) {
    if (!mr) return;
    a.M = nullptr; a.model = mr->model; a.periods = mr->periods;
    a.model_batch_stride = mr->batch_stride; a.model_chan_stride = mr->chan_stride; a.cutoff = mr->cutoff;
}

int run_istft(repet_ctx* c, const Geo& g, const Tables* tb, int64_t trim, int64_t n_out, int64_t out_offset,
              bool weighted, int64_t fade_in, int64_t fade_out, const ModelRef* mr) {
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

int check_params(const repet_params* p) {
    if (!p) return fail(REPET_ERR_BAD_ARG, "params is null");
    if (p->window_length < 64 || p->window_length > 8192 || (p->window_length & (p->window_length - 1)))
        return fail(REPET_ERR_LIMIT, "window length must be a power of two in [64, 8192]");
    if (p->step_length * 2 != p->window_length) return fail(REPET_ERR_BAD_ARG, "step length must be half the window length");
    if (p->period_lo < 0 || p->cutoff_bins < 0 || p->sim_distance_frames < 0) return fail(REPET_ERR_BAD_ARG, "negative parameter");
    if (p->flags & ~(REPET_FLAG_STRICT_REFERENCE | REPET_FLAG_REFUSE_NONFINITE))
        return fail(REPET_ERR_BAD_ARG, "unknown bits in repet_params.flags (an ABI-2 caller must zero reserved0)");
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

}  // namespace repet_eng

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

}  // extern "C"
namespace repet_eng {

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
        // the side stream carries the LONGER of two chains that must both end before the next stage (sim: the column sort beside
        // the peak picking): at the highest priority its workgroups are placed first (peaks + sort 0.275 -> 0.267 ms at cfg 2)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, greatest);
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

}  // namespace repet_eng
extern "C" {

}  // extern "C"
namespace repet_eng {
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
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->norms_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->norms_done, hipEventDisableTiming);
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
}  // namespace repet_eng
extern "C" {

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
                      &c->refine_stats, &c->R, &c->Vs, &c->rank_codes, &c->code_planes, &c->median_codes, &c->tiles_big,
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
    if (c->norms_fork) (void)hipEventDestroy(c->norms_fork);
    if (c->norms_done) (void)hipEventDestroy(c->norms_done);
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
    c->input_not_finite = not_finite;
    c->input_unscanned = false;
    if (not_finite && !c->strict) {
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
    // a float64 host upload of this context may still have its remainder plane on the way into audio_lo (copy stream): what
    // is written below must land after it, not under it
    if (c->ring.lo_in_flight) HIP_TRY(hipStreamWaitEvent(c->stream, c->ring.lo_done, 0));
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
    // device buffers are not scanned: what they hold is computed on, as in repet.py -- and in strict reference mode the passes
    // that make the result the reference's on NaN / infinite samples run whatever the planes hold (nonfinite_passes)
    c->input_not_finite = false;
    c->input_unscanned = true;
    c->ring.lo_in_flight = false;              // (the stream waited for it above and is idle now)
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

int repet_device_host_cpus(int device, int32_t* cpus, int32_t capacity, int32_t* n_cpus) {
    if (!n_cpus || capacity < 0 || (capacity > 0 && !cpus)) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (device < 0 || device >= repet_device_count()) return fail(REPET_ERR_BAD_ARG, "no such device");
    const std::vector<int> near = host_cpus_near_device(device);
    *n_cpus = (int32_t)near.size();
    for (int i = 0; i < capacity && i < (int)near.size(); ++i) cpus[i] = near[i];
    return REPET_OK;
}

int repet_ctx_set_strict_reference(repet_ctx* c, int on) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    c->strict = on != 0;
    return REPET_OK;
}

int repet_ctx_stream(repet_ctx* c, void** hip_stream) {
    if (!c || !hip_stream) return fail(REPET_ERR_BAD_ARG, "null argument");
    *hip_stream = reinterpret_cast<void*>(c->stream);
    return REPET_OK;
}

int repet_ctx_result_view(repet_ctx* c, float** dev_out, int64_t* n_values) {
    if (!c || !dev_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    *dev_out = c->out.as<float>();
    if (n_values) *n_values = c->n_samples * c->n_channels * c->n_clips;
    return REPET_OK;
}

int repet_ctx_input_view(repet_ctx* c, float** dev_audio, float** dev_audio_lo, int64_t* n_values) {
    if (!c || !dev_audio) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    DeviceGuard guard(c->device);
    // the remainder plane of a float64 upload follows the samples on the copy stream: a reader on the context's stream
    // (or behind an event recorded on it) finds both planes complete
    if (c->has_lo && c->ring.lo_in_flight) HIP_TRY(hipStreamWaitEvent(c->stream, c->ring.lo_done, 0));
    *dev_audio = c->audio.as<float>();
    if (dev_audio_lo) *dev_audio_lo = c->has_lo ? c->audio_lo.as<float>() : nullptr;
    if (n_values) *n_values = c->n_samples * c->n_channels * c->n_clips;
    return REPET_OK;
}

int repet_ctx_download_from(repet_ctx* c, const float* dev_src, int64_t n_values, double* out) {
    if (!c || (n_values > 0 && (!dev_src || !out))) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (n_values < 0) return fail(REPET_ERR_BAD_ARG, "negative count");
    if (n_values == 0) return REPET_OK;
    DeviceGuard guard(c->device);
    HIP_TRY(staged_download(c->ring, dev_src, out, (size_t)n_values, c->stream));
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

}  // extern "C"
namespace repet_eng {
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
}  // namespace repet_eng
extern "C" {

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

int repet_ctx_execute_extended_range_async(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg) {
    if (!c) return fail(REPET_ERR_BAD_ARG, "ctx is null");
    RP_TRY(check_params(p));
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    if (n_seg < 0) return fail(REPET_ERR_BAD_ARG, "negative segment count");
    if (c->n_clips > 1) return fail(REPET_ERR_BAD_ARG, "segment ranges apply to a single resident clip, not to a batch context");
    DeviceGuard guard(c->device);
    c->timing = nullptr;
    c->last_algo = REPET_EXTENDED;
    c->last_n_periods = 0;
    return exec_extended(c, p, first, n_seg);
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
    // integer PCM is finite by construction; float payloads are decoded on the device and not scanned (see upload_device_split)
    c->input_not_finite = false;
    c->input_unscanned = w.format != 1;
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
    if (list_bound < 0) {                // the bit-sliced selection: wave instructions per frame and PLANE (mask_bits.hip)
        if (network_size) *network_size = 0;
        if (instructions) *instructions = mask_sim_bits_instructions(-list_bound, 1);
        return REPET_OK;
    }
    int size = 0;
    const int n = median_network_instructions(list_bound, &size);
    if (network_size) *network_size = size;
    if (instructions) *instructions = n;
    return REPET_OK;
}

int repet_release_thread_ctx(void) {
    for (auto& kv : g_thread_ctx.by_device) repet_ctx_destroy(kv.second);
    g_thread_ctx.by_device.clear();
    release_stream_contexts();                     // (the idle contexts of repet_run_stream: process-wide)
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
    RP_TRY(check_params(p));
    c->strict = !(p->flags & REPET_FLAG_REFUSE_NONFINITE);
    RP_TRY(repet_ctx_upload(c, audio, dtype, n, ch));
    // (without a timing request nothing waits between the last kernel and the first copy back: the download is ordered behind
    // the run on the context's stream, and an error of the run surfaces there)
    if (timing) RP_TRY(repet_ctx_execute(c, algo, p, timing));
    else RP_TRY(repet_ctx_execute_async(c, algo, p));
    return repet_ctx_download(c, out);
}

}  // extern "C"
