"""BASELINE.json configs as they are stated, and the adversarial inputs SURVEY.md names, on the MI355X through the
C ABI. (The single-clip full-size checks of cfg 2-5 against the reference's goldens live in test_gpu_variants.py.)"""
import numpy as np
import pytest

import repet
from helpers import list_difference_gaps, golden_input, load_golden, periodic_clip, rms_err
from oracle import repet_oracle as orc
from repet_synth import synth

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-4


def _lists(idx, cnt, rows):
    return [set(idx[r, :cnt[r]].tolist()) for r in range(rows)]


@pytest.mark.slow
def test_cfg5_batch_of_64_clips_as_baseline_states_it():
    """BASELINE.json configs[4]: simonline on a batch of 64 synthetic 30-s 44.1 kHz stereo clips (seeds 0..63),
    resident together (repet_ctx_upload_batch: nb = 64, T = 1 291, W = 2048; one launch per stage over all clips).
    Every clip bit-identical to its own single-clip run; 8 of them against the float64 oracle (repet.py:712-911) at
    the plain 1e-4 bar with equal similar-frame lists; clip 0 against the reference's own samples."""
    fs, n_clips = 44100, 64
    clips = np.stack([synth(30, fs, 2, s) for s in range(n_clips)])
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload_batch(clips)
    ctx.execute("simonline", p)
    got = ctx.download()
    t = ctx.last_frame_count()
    rows = t - p.buffer_frames + 1
    assert (t, rows, p.buffer_frames) == (1291, 861, 431)
    idx, cnt = ctx.last_sim_indices(rows * n_clips, p.sim_number)
    stats = ctx.last_refine_stats()
    assert got.shape == clips.shape and np.all(np.isfinite(got))
    assert stats["flat_rows"] == 0
    for k in range(n_clips):                                      # the batch is 64 independent clips
        ctx.upload(clips[k])
        ctx.execute("simonline", p)
        assert np.array_equal(ctx.download(), got[k]), k
        i1, c1 = ctx.last_sim_indices(rows, p.sim_number)
        assert np.array_equal(c1, cnt[k * rows:(k + 1) * rows]) and np.array_equal(i1, idx[k * rows:(k + 1) * rows]), k
    ctx.close()
    assert np.all(got[:, :(p.buffer_frames - 1) * p.step_length] == 0)     # first 10 s exactly zero (repet.py:834)
    for k in range(0, n_clips, 9):                                # 0, 9, ..., 63: eight clips against the oracle
        tr = orc.Trace()
        want = orc.simonline(clips[k], fs, None, tr)
        theirs = tr.items["similarity_indices"]
        ours = _lists(idx[k * rows:(k + 1) * rows], cnt[k * rows:(k + 1) * rows], rows)
        differ = sum(a != set(np.asarray(b).tolist()) for a, b in zip(ours, theirs))
        assert differ == 0, (k, differ)
        assert rms_err(got[k], want) <= 2e-5, k                   # bar 1e-4; equal lists leave fp32 arithmetic only
    g = load_golden("cfg5_simonline")                             # the reference itself on clip 0
    stride = int(g["sample_stride"])
    assert rms_err(got[0][::stride], g["simonline.samples"]) <= RMS_TOL
    assert np.array_equal(cnt[:rows], g["simonline.counts"])


# ---- exactly periodic clips: exact ties in the similarity matrix (hard part 1) ----------------------
def _run_with_lists(algo, x, fs):
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, None, tr)
    theirs = [np.asarray(v) for v in tr.items["similarity_indices"]]
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    stats = ctx.last_refine_stats()
    ctx.close()
    return got, want, [idx[r, :cnt[r]] for r in range(len(theirs))], theirs, stats, p


@pytest.mark.parametrize("algo", ["sim", "simonline"])
@pytest.mark.parametrize("fs,period_hops,seconds", [(8000, 40, 24.0), (44100, 64, 30.0)])
def test_exactly_periodic_clip_with_the_period_longer_than_the_window(algo, fs, period_hops, seconds):
    """Period = k hops with k > similarity distance d: every frame has exact copies k, 2k, ... frames away, so each
    row of the similarity matrix holds EXACT ties between peaks of different +-d windows. Each tied peak is still a
    strict maximum of its own window, so the lists are well defined up to WHICH of the identical frames the top-K
    cut keeps -- and identical frames have identical magnitudes, so the median (repet.py:1535) does not care.
    Bar: plain RMS <= 1e-4, list lengths equal, lists equal as multisets of frame classes (index mod k)."""
    x = periodic_clip(fs, period_hops, seconds, 2)
    got, want, ours, theirs, stats, p = _run_with_lists(algo, x, fs)
    assert period_hops > p.sim_distance_frames
    assert not np.isnan(want).any() and not np.isnan(got).any()
    assert [len(a) for a in ours] == [len(b) for b in theirs]
    off_class = sum(sorted((a % period_hops).tolist()) != sorted((b % period_hops).tolist()) for a, b in zip(ours, theirs))
    assert off_class <= PERIODIC_OFF_CLASS.get((algo, fs), 0), (off_class, stats)
    assert rms_err(got, want) <= RMS_TOL, stats
    print(f"periodic {algo} fs {fs}: refine stats {stats}")


# rows whose list differs from the oracle's even modulo the frame class (measured; edge frames whose zero padding
# breaks the periodicity can be near-tied with interior ones)
PERIODIC_OFF_CLASS = {}


@pytest.mark.parametrize("algo", ["sim", "simonline"])
@pytest.mark.parametrize("fs,period_hops,seconds", [(8000, 12, 24.0), (44100, 20, 30.0)])
def test_exactly_periodic_clip_with_ties_inside_the_window(algo, fs, period_hops, seconds):
    """Period = k hops with k <= d: the exact copies of a frame sit INSIDE its own +-d window. The reference keeps an
    element only if it is strictly above everything else in the window (repet.py:1318-1326), so in exact arithmetic
    every tied peak is rejected, the list is empty and np.median of nothing gives NaN (repet.py:1535). In the
    reference's float64 the tie is decided by the last bit of its BLAS dot products, i.e. by rounding noise: some
    rows keep one of the copies, most keep none.
    The engine's documented policy is the exact-arithmetic one: tied elements are never local maxima. Every row is a
    flat row for the first pass and is decided again from float64 spectra (peaks_exact.hip); the copies of a frame are
    bit-identical samples, so their float64 unit rows are bit-identical and the float64 dot products -- the same sums in
    the same order -- tie exactly.
    So: wherever the oracle says NaN the engine says NaN; the engine may say NaN where the oracle's rounding noise
    kept a copy; where both are finite (edge frames) they agree to 1e-4."""
    x = periodic_clip(fs, period_hops, seconds, 2)
    got, want, ours, theirs, stats, p = _run_with_lists(algo, x, fs)
    assert period_hops <= p.sim_distance_frames
    assert np.isnan(want).any()
    assert not np.any(np.isnan(want) & ~np.isnan(got))             # the engine never invents a peak out of a tie
    both = ~np.isnan(got) & ~np.isnan(want)
    if both.any():
        assert rms_err(got[both], want[both]) <= RMS_TOL
    # the rows the engine fills are rows the oracle fills too, with the same frames (edge frames, whose zero padding
    # breaks the periodicity); the rows only the ORACLE fills are the ties its rounding noise happened to break
    for a, b in zip(ours, theirs):
        assert set(a.tolist()) <= set(b.tolist())
    assert sum(len(a) > 0 for a in ours) <= sum(len(b) > 0 for b in theirs) <= 0.2 * len(theirs)
    # more near-ties per row than the first refinement takes on (kAmbCap, peaks.hip): flat rows, decided again from
    # float64 spectra -- where identical samples give bit-identical similarities, i.e. the same exact ties
    assert stats["flat_rows"] in (0, len(theirs))
    print(f"periodic-inside {algo} fs {fs}: oracle nonempty {sum(len(b) > 0 for b in theirs)} engine nonempty "
          f"{sum(len(a) > 0 for a in ours)} of {len(theirs)} rows, refine stats {stats}")


@pytest.mark.parametrize("algo", ["sim", "simonline"])
@pytest.mark.parametrize("fs,k,seconds", [(8000, 12, 24.0), (44100, 20, 30.0)])
def test_jittered_ties_inside_the_window_are_decided_in_float64(algo, fs, k, seconds):
    """Same clips as above (copies of a frame INSIDE its own +-d window) plus 1e-7 white noise: the copies now differ by
    ~1e-13 in similarity. The float64 reference resolves that and keeps one winner per window (repet.py:1318-1326 on
    complex128 spectra, :149, :1223). fp32 spectra cannot: every row is a flat row for the first pass (round 2 documented
    this as a limit: the engine emitted NaN frames where the reference is finite). The second level (peaks_exact.hip) decides
    such rows from float64 spectra of the 48-bit samples: NaN positions EQUAL to the oracle's (none), plain RMS <= 1e-4,
    no empty list, and every difference between a list and the oracle's NAMED as a tie inside the oracle's own dot-product
    rounding."""
    x = periodic_clip(fs, k, seconds, 2, jitter=1e-7)
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, orc.Params(), tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    stats, exact = ctx.last_refine_stats(), ctx.last_exact_stats()
    ctx.close()
    ours = [idx[r, :cnt[r]] for r in range(len(theirs))]
    assert k <= p.sim_distance_frames
    assert np.array_equal(np.isnan(got), np.isnan(want)) and not np.isnan(want).any()
    assert rms_err(got, want) <= RMS_TOL
    assert all(len(a) > 0 for a in ours) and exact["rows_exact"] == len(theirs) and exact["input_has_remainders"]
    # where a list differs from the oracle's, the frame in question sits on a tie of the ORACLE's own float64 matrix: the two
    # values are apart by less than the rounding of a 1 025-term float64 dot product (sqrt(F) 2^-53 = 3.6e-15; measured: up
    # to 2.4e-15, half of them exactly 0 or 1 ulp), i.e. the reference's BLAS decided them by its summation order
    differ, named = list_difference_gaps(algo, tr, ours, p)
    assert all(gap <= 1e-14 for _, _, gap in named), sorted(g for _, _, g in named)[-5:]
    print(f"jittered ties inside the window, {algo} fs {fs}: lists of {np.mean([len(a) for a in ours]):.2f} frames on average; "
          f"{differ} of {len(theirs)} rows differ from the oracle's, all on ties of its own matrix (largest gap "
          f"{max([g for _, _, g in named] or [0.0]):.1e}); refine stats {stats}, second level {exact}")


@pytest.mark.parametrize("algo", ["sim", "simonline"])
def test_periodic_clip_with_jitter_below_fp32_resolution(algo):
    """The same tiling plus 1e-7 white noise: the ties become similarity differences of ~1e-9, which float64
    resolves and fp32 spectra cannot. Lists may then differ from the oracle's -- but only between frames that are
    equal to ~1e-7, so the plain bar on the output must still hold."""
    fs, k = 8000, 40
    x = periodic_clip(fs, k, 24.0, 2, jitter=1e-7)
    got, want, ours, theirs, stats, p = _run_with_lists(algo, x, fs)
    assert not np.isnan(want).any() and not np.isnan(got).any()
    assert rms_err(got, want) <= RMS_TOL, stats
    differ = sum(set(a.tolist()) != set(b.tolist()) for a, b in zip(ours, theirs))
    print(f"periodic+jitter {algo}: {differ} of {len(theirs)} lists differ, refine stats {stats}")
