// sim and simonline, and the plumbing of the peak picking's two refinement levels (see engine.h for the map of the engine's files)
#include "engine.h"

using namespace repet;
using namespace repet_eng;

namespace repet_eng {

// Near-tie refinement of the peak picking (peaks.hip): the tolerance inside which an fp32 similarity is not
// trusted, delta = scale * sqrt(FS) * 2^-24 (an error random walk over the FS products of unit-vector components).
// Measured against float64 on MI355X at FS = 1056 (tools/refine_probe.py --ambiguity):
//   exact-fp32 MFMA chain : rms 3.8e-7, max 5.9e-6  -> scale 4 (7.7e-6); 4x, 8x, 16x give identical index lists
//                           (0, 0, 0, 1 of 8062 rows differ from the float64 oracle; plain fp32: 71, 31, 35, 110),
//                           2x loses one more row
//   f16-split Gram kernel : rms 1.3e-7, max 1.1e-6  -> scale 2 (3.9e-6); 1x, 2x and 4x give identical lists
// The cost grows with delta (cfg 2, f16 Gram: peaks 0.28 / 0.30 / 0.34 ms at 1x / 2x / 4x).
float peak_refine_delta(int FS, bool f16_gram) {
    return (f16_gram ? 2.0f : 4.0f) * sqrtf((float)FS) * 5.9604645e-8f;
}

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
int make_refine(repet_ctx* c, const float* unit_rows, int FS, double threshold, PeakRefine* rf, int64_t rows, int clips,
                int n_cols, int d, int64_t frames) {
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
// REPET_MEDIAN=rank keeps the rank-domain selection on the packed 16-bit network (mask_sim_rank_kernel); default: the
// bit-sliced selection on the same codes (mask_bits.hip) where its layout applies.
static bool bits_median_enabled() {
    static const bool on = [] { const char* e = getenv("REPET_MEDIAN"); return !(e && (e[0] == 'f' || e[0] == 'r')); }();
    return on;
}

// Sort every column of V and fill m's rank fields (bins [0, F-1); the lone Nyquist bin stays on the float kernel).
int run_rank_columns(repet_ctx* c, const Geo& g, MaskArgs* m, hipStream_t stream, bool with_mark, int max_count, int phases) {
    const int n_cols = g.F - 1;
    const int64_t vs_pitch = round_up(g.T, 32);
    // (the bit-sliced selection reads the code planes only: no frame-major codes R then)
    const bool bits = bits_median_enabled() && mask_sim_bits_supported(g.T, g.C, n_cols, max_count);
    if (!bits) HIP_TRY(c->R.ensure((size_t)g.C * g.chan_stride * sizeof(unsigned short)));
    HIP_TRY(c->Vs.ensure((size_t)g.C * n_cols * vs_pitch * sizeof(float)));
    HIP_TRY(c->rank_codes.ensure((size_t)g.C * n_cols * vs_pitch * sizeof(unsigned short)));
    if (!bits && (c->r_pads_ptr != c->R.p || c->r_pads_stride != g.chan_stride || c->r_pads_row != g.Tpad || c->r_pads_channels != g.C ||
                  c->r_pads_fs != g.FS)) {
        HIP_TRY(launch_fill_rank_pad_rows(c->R.as<unsigned short>(), g.chan_stride, g.C, g.Tpad, g.FS, stream));
        c->r_pads_ptr = c->R.p; c->r_pads_stride = g.chan_stride; c->r_pads_row = g.Tpad; c->r_pads_channels = g.C; c->r_pads_fs = g.FS;
    }
    RankArgs a{};
    a.V = c->V.as<float>(); a.chan_stride = g.chan_stride; a.n_channels = g.C; a.T = g.T; a.FS = g.FS; a.n_cols = n_cols;
    a.R = bits ? nullptr : c->R.as<unsigned short>(); a.r_chan_stride = g.chan_stride; a.Vs = c->Vs.as<float>(); a.vs_pitch = vs_pitch;
    a.codes = c->rank_codes.as<unsigned short>();
    a.phases = phases;
    if (bits) {
        a.n_planes = code_planes_for(g.T);
        HIP_TRY(c->code_planes.ensure((size_t)g.T * a.n_planes * 64 * sizeof(unsigned)));
        a.P = c->code_planes.as<unsigned>();
        HIP_TRY(c->median_codes.ensure((size_t)g.C * g.chan_stride * sizeof(unsigned)));
        m->median_codes = c->median_codes.as<unsigned>();
    }
    // (with_mark: the chain runs alone on the main stream and every kernel gets its own timing mark -- bench.py's per-kernel rows)
    struct HookCtx { repet_ctx* c; double cells; double plane_bytes; } hc{c, (double)n_cols * (double)g.T * g.C, bits ? (double)g.T * a.n_planes * 256.0 : 0.0};
    auto hook = [](void* user, int step) {
        HookCtx* h = static_cast<HookCtx*>(user);
        if (step == 0) mark(h->c, "columns_from_rows", (4.0 + 4.0) * h->cells, 0);                    // V read, columns written
        else if (step == 1) mark(h->c, "rank_columns_sort", (4.0 + 4.0 + 2.0) * h->cells, 0);        // column read, written sorted, codes written
        else if (step == 2) mark(h->c, "code_planes", 2.0 * h->cells + h->plane_bytes, 0);
        else mark(h->c, "rows_from_codes", (2.0 + 2.0) * h->cells, 0);
    };
    HIP_TRY(launch_rank_columns(a, stream, with_mark ? +hook : nullptr, &hc));
    m->R = a.R; m->r_chan_stride = a.r_chan_stride; m->Vs = a.Vs; m->vs_pitch = vs_pitch; m->n_rank_cols = n_cols;
    m->P = a.P; m->n_planes = a.n_planes;
    // V read, columns written / read twice / written sorted, codes written column-major, read, written frame-major
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
    // Round 6: with the median on rank codes too the mask leaves the lookup kernel as a PLANE when the register inverse STFT will
    // apply it (W = 2048, mono / stereo). Rounds 3-5 masked X in place there (0.50 + 0.072 -> 0.50 + 0.091 ms with the kernels of
    // round 3); with the lookups a kernel of their own and the inverse merging in pairs the plane wins: lookups 0.1045 -> 0.0855,
    // inverse 0.0533 -> 0.0661, step 0.840 -> 0.829 ms (profiles/r06_mask_plane_ab.txt). REPET_MASK_PLANE=0 / p force either.
    const bool plane_for_ranks = ranks_ahead && mask_plane_forced() < 0 && reg_fft_supported(g.W, g.C, true);
    MaskPlaneScope plane(c, plane_for_ranks || mask_plane_wanted(ranks_ahead ? MaskKind::sim_ranks : MaskKind::sim_float));
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
    // Experiment of round 6 (REPET_RANK_TRANSPOSE=early; measured: stage -12 us, step +13 us, profiles/r06_transpose_early_ab.txt -- off):
    // the transpose that opens the column sort needs V only -- in line IN FRONT OF
    // the Gram kernel (26 us alone) instead of squeezed in beside the first pass of the peak picking (55 us there), so that the sort
    // itself is what the side stream starts with.
    static const bool early_transpose = [] { const char* e = getenv("REPET_RANK_TRANSPOSE"); return e && e[0] == 'e'; }();
    const int early_peaks = (int)std::min<int64_t>(p->sim_number, ceil_div(T, p->sim_distance_frames + 1));
    const bool transposed_early = early_transpose && ranks_ahead;
    if (transposed_early) {
        MaskArgs m0 = mask_args(c, g, p->cutoff_bins);
        RP_TRY(run_rank_columns(c, g, &m0, c->stream, false, early_peaks, 1));
        mark(c, "columns_from_rows", (4.0 + 4.0) * (g.F - 1) * (double)g.T * g.C, 0);
    }
    // The float64 norms of the unit rows (the first pass's float64 similarities divide by them: peaks.h) need the unit rows only:
    // they are computed on the side stream BESIDE the Gram kernel -- enqueued behind its launch, so the Gram's workgroups take the
    // CUs first and the 1 939 small workgroups of this kernel run where its second round leaves CUs idle (496 tiles on 256 CUs).
    // Measured (profiles/r06_peak_norms_ab.txt): the first pass is NOT faster with the table (98.6 -> 100.5 us: its span is the late
    // start of its slow rows plus their sweep, not the arithmetic of their similarities), and the two events that tie the side
    // stream's kernel in cost the Gram stage 13 us and the peak stage 12: 0.854 -> 0.881 ms per step. Off by default;
    // REPET_PEAK_NORMS=1 turns it on (A/B).
    static const bool use_norms = [] { const char* e = getenv("REPET_PEAK_NORMS"); return e && e[0] == '1'; }();
    const bool norms_beside = use_norms && c->side_stream && peak_refine_delta(g.FS, gram_f16_enabled()) > 0.0f && (g.FS & 3) == 0 && g.FS <= 1280;
    if (norms_beside) {
        HIP_TRY(c->unit_norms.ensure((size_t)g.Tpad * sizeof(double)));
        HIP_TRY(hipEventRecord(c->norms_fork, c->stream));               // the unit rows are there
    }
    RP_TRY(run_gram_full(c, c->Vn.as<float>(), T, g.FS, c->S.as<float>(), TS, true, split_in_stft(1), seg, seg_pitch, &seg_written));
    if (norms_beside) {
        HIP_TRY(hipStreamWaitEvent(c->side_stream, c->norms_fork, 0));
        HIP_TRY(launch_unit_row_norms(c->Vn.as<float>(), T, g.FS, c->unit_norms.as<double>(), c->side_stream));
        HIP_TRY(hipEventRecord(c->norms_done, c->side_stream));
    }
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
    if (norms_beside) {
        rf.unit_norms = c->unit_norms.as<double>();
        HIP_TRY(hipStreamWaitEvent(c->stream, c->norms_done, 0));
    }
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
        // REPET_RANK_OVERLAP=0 (measurement switch): the sort BEHIND the peak picking on the main stream, every kernel of both
        // chains with a timing mark of its own -- what bench.py's per-kernel rows under "peaks+rank_columns" come from
        static const bool overlap = [] { const char* e = getenv("REPET_RANK_OVERLAP"); return !(e && e[0] == '0'); }();
        const bool beside = use_rank && overlap;
        // (Measured and dropped: starting the sort behind the first pass of the peak picking, beside its second level --
        // peaks + sort 0.446 against 0.419 ms: the second level's kernels hold a whole register file per wave and do not share
        // a CU with the sort any better than the first pass does.)
        // (Measured and dropped: starting the sort behind the first pass of the peak picking, beside its second level --
        // peaks + sort 0.416 against 0.373 ms: the sort fills the first pass's tail better than it shares the GPU with the
        // one-wave-per-SIMD kernels of the second level.)
        if (beside) {
            HIP_TRY(hipEventRecord(c->fork_event, c->stream));          // V is complete (so is S)
            HIP_TRY(hipStreamWaitEvent(c->side_stream, c->fork_event, 0));
            RP_TRY(run_rank_columns(c, g, &m, c->side_stream, false, max_peaks, transposed_early ? 2 : 0));
            HIP_TRY(hipEventRecord(c->join_event, c->side_stream));
        }
        const size_t scratch = local_maxima_scratch_bytes(T, (int)T, p->sim_distance_frames);
        if (scratch > 0) HIP_TRY(c->peak_scratch.ensure(scratch));
        hipError_t e = launch_local_maxima(c->S.as<float>(), T, 0, (int)T, TS, 0, (float)p->sim_threshold,
                                           p->sim_distance_frames, K, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), c->stream, 0, &rf,
                                           nullptr, scratch > 0 ? c->peak_scratch.p : nullptr, nullptr, seg, seg_pitch);
        if (e == hipErrorInvalidValue) return fail(REPET_ERR_LIMIT, "sim: clip has too many frames for the peak-picking kernel's LDS row");
        HIP_TRY(e);
        if (use_rank && !beside) mark(c, "local_maxima_pass1", 12.0 * T * seg_pitch + 4.0 * K * T, 0);   // records read, lists written (+ the lines of S it asks for)
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
            mark(c, use_rank ? "local_maxima_level2" : "local_maxima", use_rank ? 0.0 : 4.0 * T * T + 4.0 * K * T, 0);
            if (use_rank) RP_TRY(run_rank_columns(c, g, &m, c->stream, true, max_peaks, transposed_early ? 2 : 0));
        }
        HIP_TRY(launch_mask_sim(m, c->idx.as<int32_t>(), KP, c->cnt.as<int32_t>(), 0, max_peaks, c->stream, c->side_stream,
                                c->fork_event, c->join_event, 3, m.P != nullptr));
        c->last_median_path = m.P ? 2 : m.R ? 1 : 0;
        c->last_FS = g.FS; c->last_chan_stride = g.chan_stride;
        if (m.P) {
            // the selection gathers one 256-byte plane row per list entry and plane (+ the frame's own), and leaves a word per cell
            mark(c, "mask_sim_select", 256.0 * m.n_planes * (K + 1.0) * T + 4.0 * (g.F - 1) * T * g.C, 0);
            HIP_TRY(launch_mask_from_codes(m, c->cnt.as<int32_t>(), c->stream));
            // V and the code word read, X masked in place (or the mask written), two table entries for the cells that need them
            mark(c, "mask_sim", (4.0 + 4.0 + 8.0 + (c->mask_plane ? 4.0 : 16.0)) * (g.F - 1) * T * g.C, 0);
        } else
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
    if (c->nonfinite_passes() && B > 1) {
        // strict reference mode: repet.py never writes the warm-up frames (repet.py:834: its background stays 0 there); the engine
        // gives them the mask 0, and 0 x NaN would be NaN -- their spectra (which only the inverse STFT still reads) are cleared
        for (int b = 0; b < nb; ++b)
            for (int ch = 0; ch < g.C; ++ch)
                HIP_TRY(hipMemsetAsync(c->X.as<float2>() + b * spec_stride + ch * g.chan_stride, 0,
                                       (size_t)std::min<int64_t>(B - 1, T) * g.FS * sizeof(float2), c->stream));
    }
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

}  // namespace repet_eng
