// the streaming handle: repet_online_* (see engine.h for the map of the engine's files)
#include "engine.h"

using namespace repet;
using namespace repet_eng;


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

namespace repet_eng {

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

}  // namespace repet_eng

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

