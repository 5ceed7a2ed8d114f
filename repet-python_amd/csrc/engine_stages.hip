// stage-level exports and the accessors of the last run's integer intermediates (see engine.h for the map of the engine's files)
#include "engine.h"

using namespace repet;
using namespace repet_eng;

extern "C" {

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

int repet_selfsim_records(repet_ctx* c, const float* v, int64_t T, int32_t F, float* s_out, float* max_out, float* second_out,
                          int32_t* at_out) {
    if (!c || !v || !s_out || !max_out || !second_out || !at_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    DeviceGuard guard(c->device);
    const int FS = (int)round_up(F, kFreqAlign);
    const int64_t Tpad = round_up(T, kTile), TS = round_up(T, 64);
    const int seg_pitch = segment_pitch((int)TS), n_seg = (int)ceil_div(T, kSegWidth);
    HIP_TRY(c->tmp_a.ensure((size_t)T * F * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(c->tmp_a.p, v, (size_t)T * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c->Vn.ensure((size_t)Tpad * FS * sizeof(float)));
    HIP_TRY(hipMemsetAsync(c->Vn.p, 0, (size_t)Tpad * FS * sizeof(float), c->stream));
    HIP_TRY(launch_unit_rows(c->tmp_a.as<float>(), c->Vn.as<float>(), T, F, FS, c->stream));
    HIP_TRY(c->S.ensure((size_t)T * TS * sizeof(float)));
    HIP_TRY(c->seg.ensure((size_t)T * 3 * seg_pitch * sizeof(float)));
    // the records as `sim` gets them: from the 256 x 256 kernel's epilogue for clips of 2 048 frames and more, else by a pass over S
    RP_TRY(run_gram_full(c, c->Vn.as<float>(), T, FS, c->S.as<float>(), TS, true, false, c->seg.as<float>(), seg_pitch));
    RP_TRY(d2h_pitched(c, s_out, c->S.as<float>(), TS, T, T));
    for (int plane = 0; plane < 3; ++plane) {
        void* dst = plane == 0 ? (void*)max_out : plane == 1 ? (void*)second_out : (void*)at_out;
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)n_seg * 4, c->seg.as<float>() + (size_t)plane * seg_pitch, (size_t)3 * seg_pitch * 4,
                                 (size_t)n_seg * 4, (size_t)T, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return REPET_OK;
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

int repet_mask_sim_ranked(repet_ctx* c, const float* v, int64_t T, int32_t F, const int32_t* idx, const int32_t* count,
                          int32_t number, int32_t path, float* mask_out, uint32_t* median_codes_out) {
    if (!c || !v || !idx || !count || !mask_out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (path != 1 && path != 2) return fail(REPET_ERR_BAD_ARG, "path: 1 packed network on rank codes, 2 bit-sliced selection");
    const int n_cols = F - 1;
    if (F <= 128 || (n_cols & 127) || !rank_columns_supported(T) || number < 2 || number > 128)
        return fail(REPET_ERR_LIMIT, "rank-domain median: 1024 < n_frames <= 30720, n_freq - 1 a multiple of 128, lists of 2..128 entries");
    if (path == 2 && !mask_sim_bits_supported(T, 1, n_cols, number)) return fail(REPET_ERR_LIMIT, "bit-sliced selection: n_freq - 1 a power of two, at most 2048");
    DeviceGuard guard(c->device);
    MaskArgs m; int FS;
    RP_TRY(stage_mask_common(c, v, T, F, &m, &FS));
    const int64_t rows = T + kPadRows, vs_pitch = round_up(T, 32);
    HIP_TRY(c->R.ensure((size_t)rows * FS * sizeof(unsigned short)));
    c->r_pads_ptr = nullptr;                              // this export lays R out differently
    HIP_TRY(launch_fill_rank_pad_rows(c->R.as<unsigned short>(), rows * FS, 1, T, FS, c->stream));
    HIP_TRY(c->Vs.ensure((size_t)n_cols * vs_pitch * sizeof(float)));
    HIP_TRY(c->rank_codes.ensure((size_t)n_cols * vs_pitch * sizeof(unsigned short)));
    RankArgs a{};
    a.V = c->V.as<float>(); a.chan_stride = rows * FS; a.n_channels = 1; a.T = T; a.FS = FS; a.n_cols = n_cols;
    a.R = path == 2 ? nullptr : c->R.as<unsigned short>(); a.r_chan_stride = rows * FS; a.Vs = c->Vs.as<float>(); a.vs_pitch = vs_pitch;
    a.codes = c->rank_codes.as<unsigned short>();
    if (path == 2) {
        a.n_planes = code_planes_for(T);
        HIP_TRY(c->code_planes.ensure((size_t)T * a.n_planes * 64 * sizeof(unsigned)));
        a.P = c->code_planes.as<unsigned>();
        HIP_TRY(c->median_codes.ensure((size_t)rows * FS * sizeof(unsigned)));
        HIP_TRY(hipMemsetAsync(c->median_codes.p, 0, (size_t)rows * FS * sizeof(unsigned), c->stream));
        m.median_codes = c->median_codes.as<unsigned>();
    }
    HIP_TRY(launch_rank_columns(a, c->stream));
    m.R = a.R; m.r_chan_stride = a.r_chan_stride; m.Vs = a.Vs; m.vs_pitch = vs_pitch; m.n_rank_cols = n_cols;
    m.P = a.P; m.n_planes = a.n_planes;
    const int KP = std::max(number, kMinIdxPitch);
    HIP_TRY(c->idx.ensure((size_t)T * KP * sizeof(int32_t)));
    HIP_TRY(c->cnt.ensure((size_t)T * sizeof(int32_t)));
    HIP_TRY(hipMemsetAsync(c->idx.p, 0, (size_t)T * KP * sizeof(int32_t), c->stream));
    HIP_TRY(hipMemcpy2DAsync(c->idx.p, (size_t)KP * sizeof(int32_t), idx, (size_t)number * sizeof(int32_t),
                             (size_t)number * sizeof(int32_t), T, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->cnt.p, count, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, number, c->stream));
    if (median_codes_out) {
        if (path != 2) return fail(REPET_ERR_BAD_ARG, "median codes exist on path 2 only");
        HIP_TRY(hipMemcpy2DAsync(median_codes_out, (size_t)n_cols * 4, c->median_codes.p, (size_t)FS * 4, (size_t)n_cols * 4, T,
                                 hipMemcpyDeviceToHost, c->stream));
    }
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

int repet_ctx_download_input(repet_ctx* c, float* samples_out, float* remainders_out, int32_t* has_remainders) {
    if (!c || !samples_out || !remainders_out || !has_remainders) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (c->n_channels < 1) return fail(REPET_ERR_BAD_ARG, "no clip uploaded");
    DeviceGuard guard(c->device);
    const size_t bytes = (size_t)c->n_samples * c->n_channels * c->n_clips * sizeof(float);
    if (c->has_lo && c->ring.lo_in_flight) HIP_TRY(hipStreamWaitEvent(c->stream, c->ring.lo_done, 0));
    HIP_TRY(hipMemcpyAsync(samples_out, c->audio.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (c->has_lo) HIP_TRY(hipMemcpyAsync(remainders_out, c->audio_lo.p, bytes, hipMemcpyDeviceToHost, c->stream));
    else std::memset(remainders_out, 0, bytes);
    HIP_TRY(hipStreamSynchronize(c->stream));
    *has_remainders = c->has_lo ? 1 : 0;
    return REPET_OK;
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

int repet_ctx_last_median_codes(repet_ctx* c, uint32_t* out, int64_t n_frames, int32_t n_bins) {
    if (!c || !out) return fail(REPET_ERR_BAD_ARG, "null argument");
    if (c->last_median_path != 2 || !c->median_codes.p) return fail(REPET_ERR_BAD_ARG, "the last run did not take the bit-sliced selection");
    if (n_frames != c->last_T || n_bins < 1 || n_bins > c->last_FS) return fail(REPET_ERR_BAD_ARG, "shape does not match the last run");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int ch = 0; ch < c->n_channels; ++ch)
        HIP_TRY(hipMemcpy2D(out + (size_t)ch * n_frames * n_bins, (size_t)n_bins * 4, c->median_codes.as<unsigned>() + (size_t)ch * c->last_chan_stride,
                            (size_t)c->last_FS * 4, (size_t)n_bins * 4, n_frames, hipMemcpyDeviceToHost));
    return REPET_OK;
}

int repet_ctx_last_median_path(repet_ctx* c, int32_t* path) {
    if (!c || !path) return fail(REPET_ERR_BAD_ARG, "null argument");
    *path = c->last_median_path;
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
