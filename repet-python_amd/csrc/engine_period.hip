// original, extended, adaptive: the period family of the REPET variants (see engine.h for the map of the engine's files)
#include "engine.h"

using namespace repet;
using namespace repet_eng;

namespace repet_eng {

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

int exec_extended(repet_ctx* c, const repet_params* p, int64_t first, int64_t n_seg) {
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
        x->strict = c->strict; x->input_not_finite = c->input_not_finite; x->input_unscanned = c->input_unscanned;
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

}  // namespace repet_eng
