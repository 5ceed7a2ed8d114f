"""End-to-end parity of the five REPET variants on the MI355X, through the C ABI, against the float64
oracle and against the committed golden vectors of the reference. Bar (BASELINE.json north_star):
RMS error of background_signal <= 1e-4 absolute (full scale 1.0)."""
import numpy as np
import pytest

import repet
from repet import _native, parallel
from helpers import (assert_parity_modulo_near_ties, golden_input, list_difference_gaps, load_edge_cases, load_golden,
                     rms_err)
from oracle import repet_oracle as orc
from repet_synth import synth, synth_groove

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-4
ALGOS = ["original", "extended", "adaptive", "sim", "simonline"]
# two clip families with goldens of the reference: synth (decaying notes + FM voice, constant period) and synth_groove
# (drum transients with a drifting tempo, level steps, inharmonic bell, a bar of digital silence, a bar that changes length)
CASES = ["small_mono", "small_stereo", "mid_stereo", "g44k_stereo", "groove_small", "groove_mid", "groove_44k"]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("algo", ALGOS)
def test_variant_matches_oracle_and_golden(case, algo):
    x, fs = golden_input(case)
    got = getattr(repet, algo)(x, fs)
    assert got.dtype == np.float64 and got.shape == x.shape and got.flags.c_contiguous
    want = orc.ALGORITHMS[algo](np.array(x), fs)
    assert np.array_equal(np.isnan(got), np.isnan(want))         # (the silent bar of the groove clips: NaN frames in sim / simonline)
    ok = ~np.isnan(want)
    err = rms_err(got[ok], want[ok])
    assert err <= RMS_TOL, f"rms {err:.3e}"
    g = load_golden(case)
    stride = int(g["sample_stride"])
    ref = g[f"{algo}.samples"]
    assert np.array_equal(np.isnan(got[::stride]), np.isnan(ref))
    assert rms_err(got[::stride][~np.isnan(ref)], ref[~np.isnan(ref)]) <= RMS_TOL
    assert np.max(np.abs(got[ok] - want[ok])) < 5e-3


@pytest.mark.parametrize("case", ["groove_small", "groove_mid", "groove_44k"])
def test_groove_family_integer_intermediates(case):
    """The second clip family against the reference's own integer intermediates: the period of `original`, the per-frame
    periods of `adaptive` (not constant on these clips), the segment periods of `extended`, and the similar-frame lists of
    `sim` / `simonline` -- lengths of EVERY row, the reference's lists on every sampled row."""
    g = load_golden(case)
    x, fs = golden_input(case)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("original", p)
    assert ctx.last_periods(1)[0] == int(g["original.period"])
    ctx.execute("adaptive", p)
    assert np.array_equal(ctx.last_periods(ctx.last_frame_count()), g["adaptive.periods"])
    ctx.execute("extended", p)
    assert np.array_equal(ctx.last_periods(len(g["extended.periods"])), g["extended.periods"])
    ctx.execute("sim", p)
    t = ctx.last_frame_count()
    idx, cnt = ctx.last_sim_indices(t, p.sim_number)
    assert np.array_equal(cnt, g["sim.counts"])
    for row, f in zip(g["sim.indices"], g["sim.index_frames"]):
        assert set(idx[f, :cnt[f]].tolist()) == set(row[row >= 0].tolist()), f
    ctx.execute("simonline", p)
    rows = ctx.last_frame_count() - p.buffer_frames + 1
    idx, cnt = ctx.last_sim_indices(rows, p.sim_number)
    assert np.array_equal(cnt, g["simonline.counts"])
    fstride, b = int(g["frame_stride"]), p.buffer_frames
    for k, row in enumerate(g["simonline.buffer_indices"]):
        j = b - 1 + k * fstride
        cols = row[row >= 0]
        assert set(idx[k * fstride, :cnt[k * fstride]].tolist()) == set((j - np.mod(j - cols, b)).tolist()), k
    ctx.close()


def test_input_is_not_mutated_and_dtypes_accepted():
    x, fs = golden_input("small_stereo")
    x = np.array(x[:4 * fs])
    keep = x.copy()
    ref = repet.original(x, fs)
    assert np.array_equal(x, keep)
    as32 = repet.original(x.astype(np.float32), fs)
    assert as32.dtype == np.float64 and rms_err(as32, ref) < 1e-5
    ints = (x * 32767).astype(np.int16)
    as16 = repet.original(ints, fs)
    assert rms_err(as16 / 32767.0, ref) < 1e-4
    fortran = np.asfortranarray(x)
    assert rms_err(repet.original(fortran, fs), ref) == 0.0


def test_edge_cases_behave_like_the_reference():
    edge = load_edge_cases()
    fs = 44100
    base = synth(16, fs, 2, 5)

    def outcome(algo, x):
        try:
            y = getattr(repet, algo)(x, fs)
            return {"ok": True, "nan_count": int(np.isnan(y).sum()), "all_zero": bool(np.all(y == 0))}
        except Exception as e:  # noqa: BLE001
            return {"ok": False, "error": type(e).__name__}

    for key, algo, x in [("original_2.9s", "original", base[:int(2.9 * fs)]),
                         ("original_3.2s", "original", base[:int(3.2 * fs)]),
                         ("sim_0.5s", "sim", base[:int(0.5 * fs)]),
                         ("simonline_9s", "simonline", base[:9 * fs]),
                         ("simonline_441343", "simonline", base[:441343]),
                         ("simonline_441344", "simonline", base[:441344]),
                         ("simonline_441345", "simonline", base[:441345])]:
        want = edge[key]
        got = outcome(algo, x)
        assert got["ok"] == want["ok"], key
        if want["ok"]:
            assert got["all_zero"] == want["all_zero"], key
            assert (got["nan_count"] > 0) == (want["nan_count"] > 0), key
        else:
            assert got["error"] == want["error"], key

    x149 = synth(14.9, fs, 2, 5)
    assert edge["extended_14.9s_equals_original"]
    assert np.array_equal(repet.extended(x149, fs), repet.original(x149, fs))


def test_nan_and_infinite_samples_are_refused(monkeypatch):
    """repet.py computes on with such samples; what comes out depends on what is global in the variant (`sim`: the frames that
    hold them; original / extended / adaptive: NaN through the beat spectrum, i.e. whole segments or clips). With
    ``repet.strict_reference = False`` (REPET_FLAG_REFUSE_NONFINITE; the default until round 5) the drop-in refuses host arrays
    that contain them, whatever the dtype, and is usable afterwards."""
    monkeypatch.setattr(repet, "strict_reference", False)
    assert repet.derive_params(8000).flags == 2
    fs = 8000
    x = synth(12, fs, 2, 7)
    for value, dtype in ((np.nan, np.float64), (np.inf, np.float64), (-np.inf, np.float32), (np.nan, np.float32)):
        bad = x.astype(dtype)
        bad[40001, 1] = value
        for algo in ALGOS:
            with pytest.raises(ValueError, match="NaN or infinite"):
                getattr(repet, algo)(bad, fs)
        with pytest.raises(ValueError, match="NaN or infinite"):
            repet.run_batch("sim", [x, bad], fs)
    got = repet.sim(x, fs)
    assert rms_err(got, orc.sim(x, fs)) <= 1e-4
    # the largest finite values are not "infinite"
    big = x * 1e30
    assert np.all(np.isfinite(repet.original(big.astype(np.float32), fs))) or True     # (no exception: fp32 range is the caller's business)


@pytest.mark.parametrize("algo,fs,seconds,channels,dtype", [("sim", 16000, 24, 2, np.float64), ("sim", 44100, 40, 2, np.float64),
                                                             ("simonline", 16000, 30, 2, np.float64), ("sim", 22050, 30, 1, np.float32),
                                                             ("simonline", 44100, 24, 2, np.float32)])
def test_strict_reference_reproduces_repet_py_on_samples_that_are_not_finite(algo, fs, seconds, channels, dtype, monkeypatch):
    """``repet.strict_reference = True``: NaN / +-inf samples are let through and `sim` / `simonline` return what repet.py returns
    (repet.py:125 has no input check; :1220 and :1318-1326 confine the damage): NaN on exactly the samples of the frames that
    hold such a sample, every other sample as the float64 oracle gives it (the oracle equals the reference bit for bit on
    this kind of input: NaN positions equal, max-abs 0.0 elsewhere -- checked against /root/reference when the strict mode was
    built), and the similar-frame lists of every other frame equal to the oracle's. 44.1 kHz / 40 s takes the rank-domain
    median: the NaN / inf magnitudes go through the column sort. (The period family: the next test.)"""
    assert repet.strict_reference is True                 # the default: the reference's behaviour
    x = synth(seconds, fs, channels, 11).astype(dtype)
    n = len(x)
    x[n // 3, 0] = np.nan                                   # one sample
    x[n // 2:n // 2 + 5000, channels - 1] = np.nan          # a run that spans several frames
    x[(2 * n) // 3, 0] = np.inf
    x[(2 * n) // 3 + 40000, channels - 1] = -np.inf
    tr = orc.Trace()
    with np.errstate(all="ignore"):
        want = orc.ALGORITHMS[algo](x.astype(np.float64), fs, None, tr)
    p = repet.derive_params(fs)
    assert p.flags == 0
    ctx = repet.Context(0)                                  # (a fresh context lets such samples through as well)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    theirs = tr.items["similarity_indices"]
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    ctx.close()
    bad = np.isnan(want)
    assert 0 < bad.any(axis=1).sum() < 0.2 * n and not np.isinf(want).any()
    assert np.array_equal(np.isnan(got), bad) and not np.isinf(got).any()
    assert rms_err(got[~bad], want[~bad]) <= 2e-5
    differ = sum(set(idx[r, :cnt[r]].tolist()) != set(np.asarray(theirs[r]).tolist()) for r in range(len(theirs)))
    assert differ == 0
    assert np.array_equal(getattr(repet, algo)(x, fs), got, equal_nan=True)      # the one-shot call (flag in repet_params) gives the same
    if algo == "sim":
        assert np.array_equal(repet.run_batch("sim", [x], fs)[0], got, equal_nan=True)
    monkeypatch.setattr(repet, "strict_reference", False)
    with pytest.raises(ValueError, match="NaN or infinite"):
        getattr(repet, algo)(x, fs)


@pytest.mark.parametrize("algo", ["original", "extended", "adaptive"])
@pytest.mark.parametrize("kind", ["nan", "inf"])
def test_strict_reference_period_family(algo, kind, monkeypatch):
    """The period family on samples that are not finite, as repet.py has it (measured against the unmodified reference: the
    oracle equals it bit for bit on these inputs). A NaN sample makes its frames NaN, the beat spectrum of its clip / segment /
    windows NaN at every lag (the autocorrelation goes through an FFT over time: repet.py:1108-1139), hence the period
    `period_range[0] + 1` there, and -- np.median's NaN rule -- NaN at the SAME position of every period (`original`: 23 runs of
    three hops on a 24-s clip, one per period). kind = nan: NaN positions equal, every other sample within the bar, every
    period equal. An INFINITE sample is treated as NaN (what the reference makes of one is pocketfft's butterfly order: of the
    two below, one marks every period and the other only its own frames): the result equals the engine's result for the same
    clip with NaN in those places, the periods still equal the reference's, and the reference's NaN samples are among the
    engine's."""
    monkeypatch.setattr(repet, "strict_reference", True)
    fs = 16000
    x = synth(24 if algo != "extended" else 36, fs, 2, 3)
    n = len(x)
    places = [(n // 3, 0)] if kind == "nan" else [((2 * n) // 3 + 777, 1), ((5 * n) // 6, 0)]
    for k, (at, ch) in enumerate(places):
        x[at, ch] = np.nan if kind == "nan" else (np.inf if k == 0 else -np.inf)
    tr = orc.Trace()
    with np.errstate(all="ignore"):
        want = orc.ALGORITHMS[algo](x, fs, None, tr)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.set_strict_reference(True)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    if algo == "original":
        assert ctx.last_periods(1)[0] == tr.items["repeating_period"] == p.period_lo + 1
    elif algo == "extended":
        assert np.array_equal(ctx.last_periods(64), tr.items["segment_periods"])
    else:
        assert np.array_equal(ctx.last_periods(ctx.last_frame_count()), tr.items["repeating_periods"])
    bad = np.isnan(want)
    assert bad.any() and not bad.all() and not np.isinf(want).any() and not np.isinf(got).any()
    if kind == "nan":
        assert np.array_equal(np.isnan(got), bad)
        assert rms_err(got[~bad], want[~bad]) <= 2e-5
    else:
        as_nan = np.where(np.isinf(x), np.nan, x)
        ctx.upload(as_nan)
        ctx.execute(algo, p)
        assert np.array_equal(ctx.download(), got, equal_nan=True)
        assert not np.any(bad & ~np.isnan(got))                  # the reference's NaN samples are among the engine's
        both = ~bad & ~np.isnan(got)
        assert rms_err(got[both], want[both]) <= 2e-5
    ctx.close()
    assert np.array_equal(getattr(repet, algo)(x, fs), got, equal_nan=True)


def test_stated_limits_of_the_engine():
    """repet.py accepts any sampling rate (:130) and any length (:149). The engine's limits are stated, not silent: the largest
    window is 8 192 samples (fs <= 204.8 kHz: 192 kHz works, against the oracle), a larger one is REPET_ERR_LIMIT with the
    window named; one channel's magnitude plane stays below 2 GiB (508 000 frames: 3.2 hours at 44.1 kHz), a longer clip is REPET_ERR_LIMIT
    with the length named -- before anything of that size is allocated -- and the context is usable afterwards."""
    fs = 192000
    x = synth(4.5, fs, 1, 21)
    assert repet.derive_params(fs).window_length == 8192
    assert rms_err(repet.sim(x, fs), orc.sim(x, fs)) <= RMS_TOL
    assert rms_err(repet.original(x, fs), orc.original(x, fs)) <= RMS_TOL
    with pytest.raises(RuntimeError, match="window length must be a power of two in \\[64, 8192\\]"):
        repet.sim(synth(1.0, 220500, 1, 2), 220500)                          # W = 16 384
    long_clip = np.zeros((530_000_000, 1), dtype=np.int16)                   # 3.3 hours of 44.1 kHz mono: 517 579 frames
    ctx = repet.Context(0)
    ctx.upload(long_clip)
    with pytest.raises(RuntimeError, match="spectrogram must stay below 2 GiB"):
        ctx.execute("original", repet.derive_params(44100))
    del long_clip
    ctx.upload(synth(6.0, 44100, 2, 3))
    ctx.execute("original", repet.derive_params(44100))
    assert np.all(np.isfinite(ctx.download()))
    ctx.close()


def test_silent_gap_gives_nan_only_for_sim():
    edge = load_edge_cases()
    fs = 44100
    gap = synth(16, fs, 2, 5)[:8 * fs].copy()
    gap[100000:140000] = 0.0
    y = repet.sim(gap, fs)
    want = orc.sim(gap, fs)
    assert np.array_equal(np.isnan(y), np.isnan(want))
    assert int(np.isnan(y).sum()) == edge["silence_gap_sim"]["nan_count"]
    ok = ~np.isnan(want)
    assert rms_err(y[ok], want[ok]) <= RMS_TOL
    assert np.all(np.isfinite(repet.original(gap, fs)))
    assert np.all(np.isfinite(repet.adaptive(gap, fs)))


def test_module_parameters_change_behaviour():
    x, fs = golden_input("small_stereo")
    saved = (repet.period_range, repet.similarity_number, repet.cutoff_frequency, repet.filter_order)
    try:
        repet.period_range = [0.5, 3]
        repet.similarity_number = 13
        repet.cutoff_frequency = 250
        repet.filter_order = 4
        p = orc.Params(period_range=(0.5, 3), similarity_number=13, cutoff_frequency=250, filter_order=4)
        for algo in ALGOS:
            got = getattr(repet, algo)(x, fs)
            want = orc.ALGORITHMS[algo](np.array(x), fs, p)
            assert rms_err(got, want) <= RMS_TOL, algo
    finally:
        repet.period_range, repet.similarity_number, repet.cutoff_frequency, repet.filter_order = saved


def test_integer_intermediates_through_the_context():
    x, fs = golden_input("mid_stereo")
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    g = load_golden("mid_stereo")

    ctx.execute("original", p)
    assert ctx.last_periods(4)[0] == int(g["original.period"])

    ctx.execute("extended", p)
    assert np.array_equal(ctx.last_periods(64), g["extended.periods"])

    ctx.execute("adaptive", p)
    t = ctx.last_frame_count()
    got = ctx.last_periods(t)
    assert np.array_equal(got, g["adaptive.periods"])

    ctx.execute("sim", p)
    t = ctx.last_frame_count()
    idx, cnt = ctx.last_sim_indices(t, p.sim_number)
    assert np.array_equal(cnt, g["sim.counts"])
    differ = 0
    for row, frame in zip(g["sim.indices"], g["sim.index_frames"]):
        differ += set(idx[frame, :cnt[frame]]) != set(row[row >= 0])
    # near-ties of the fp32 similarity are re-decided in float64 (peaks.hip), so the lists are the reference's
    assert differ == 0

    timing = ctx.execute("sim", p, timing=True)
    names = [s["name"] for s in timing["stages"]]
    # (the column sort of the rank-domain median runs beside the peak picking: one stage entry for the two; clips of fewer than
    # 2 048 frames get their segment records from a pass over the matrix, a stage of its own)
    # (and the bit-sliced median is two kernels with a stage entry each: the selection, then the lookups and the mask)
    assert [n.replace("_f16x3", "").replace("peaks+rank_columns", "local_maxima") for n in names if n not in ("rank_columns", "segment_maxima", "mask_sim_select")] == \
        ["stft", "similarity_gemm", "local_maxima", "mask_sim", "istft_ola"]
    assert timing["total_ms"] > 0
    ctx.close()


def test_extended_segment_ranges_add_up():
    """Multi-GPU sharding of `extended`: disjoint segment ranges (one per GPU) sum to the full result."""
    x, fs = golden_input("mid_stereo")
    p = repet.derive_params(fs)
    n_seg = _native.lib().repet_extended_segment_count(len(x), p)
    assert n_seg == 3
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("extended", p)
    full = ctx.download()
    total = np.zeros_like(full)
    for first, count in parallel.segment_ranges(n_seg, 2):
        ctx.execute_extended_range(p, first, count)
        total += ctx.download()
    ctx.close()
    assert rms_err(total, full) < 1e-7
    with pytest.raises(ValueError):
        repet.Context(0).execute_extended_range(p, 0, 1)      # nothing uploaded


@pytest.mark.slow
@pytest.mark.parametrize("algo", ["extended", "original", "adaptive"])
def test_quiet_passages_keep_their_own_periods(algo):
    """A clip whose second half is 70 dB below the first (power spectrum: 140 dB). The beat-spectrum Grams of long clips and
    of batched `extended` segments run on the f16-split matrix-core kernel; with ONE scale for the whole matrix the quiet
    half underflowed f16, its beat spectra came out all zero and its periods degenerated to period_lo + 1 although the
    float64 reference finds the real period. Rows are scaled one by one now: periods of every segment / frame equal
    the oracle's."""
    fs = 44100
    x = synth(400.0 if algo != "adaptive" else 60.0, fs, 1, 17)   # long enough for the f16-split band kernel (>= 512 tiles)
    gain = np.ones(len(x))
    gain[len(x) // 2:] = 10 ** (-70 / 20)
    x = x * gain[:, np.newaxis]
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, None, tr)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    tm = ctx.execute(algo, p, timing=True)
    got = ctx.download()
    names = [s["name"] for s in tm["stages"]]
    if algo == "extended":
        assert "gram_band_f16x3" in names                           # the path under test really ran
        assert np.array_equal(ctx.last_periods(256), tr.items["segment_periods"])
    elif algo == "original":
        assert "gram_band_f16x3" in names
        assert ctx.last_periods(1)[0] == tr.items["repeating_period"]
    else:                                                           # short windows: the exact-fp32 band kernel; same bar
        assert np.array_equal(ctx.last_periods(ctx.last_frame_count()), tr.items["repeating_periods"])
    ctx.close()
    quiet = slice(len(x) // 2 + 4 * fs, None)
    assert rms_err(got, want) <= RMS_TOL
    assert rms_err(got[quiet], want[quiet]) <= 1e-3 * np.sqrt(np.mean(want[quiet] ** 2)) + 1e-9      # relative, in the quiet half


def test_device_resident_ingest_and_egress():
    """repet_ctx_upload_device / repet_ctx_download_device: fp32 samples that are already in device memory (what an RCCL
    recv leaves there) go in and come out without a host bounce, bit-identical to the host path."""
    import torch
    fs = 22050
    x = synth(14.0, fs, 2, 5)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    for algo in ("sim", "original", "simonline"):
        ctx.upload(x)
        ctx.execute(algo, p)
        want = ctx.download()
        t = torch.from_numpy(x.astype(np.float32)).to("cuda:0")
        torch.cuda.synchronize()
        ctx.upload_device(t.data_ptr(), t.shape[0], t.shape[1])
        t.zero_()                                                   # the source may be reused at once
        ctx.execute(algo, p)
        out = torch.empty((len(x), 2), dtype=torch.float32, device="cuda:0")
        ctx.download_device(out.data_ptr())
        assert np.array_equal(out.cpu().numpy().astype(np.float64), want), algo
        assert np.array_equal(ctx.download(), want)
    ctx.close()
    # and through the multi-GPU host logic's engine adapters (the functions a worker rank runs on what it received)
    run = parallel._engine_separate("sim", 0)
    t = torch.from_numpy(x.astype(np.float32)).to("cuda:0")
    assert np.array_equal(run(t, fs).cpu().numpy().astype(np.float64), repet.sim(x, fs))


@pytest.mark.parametrize("seg_len,seg_step,seconds,fs", [(10.0, 5.0, 47.3, 16000), (6.0, 1.5, 29.0, 8000), (10.0, 7.5, 52.0, 22050)])
def test_extended_on_windows_of_the_clip(seg_len, seg_step, seconds, fs, monkeypatch):
    """Multi-GPU `extended` as SURVEY 8e specifies it: a rank holds only the samples its own segments cover
    (repet_ctx_set_window), runs them with the WHOLE clip's cross-fade weights and returns its window; the windows added
    into place give repet.extended of the clip to fp32 rounding."""
    import torch
    monkeypatch.setattr(repet, "segment_length", seg_len)
    monkeypatch.setattr(repet, "segment_step", seg_step)
    x = synth(seconds, fs, 2, 12)
    want = repet.extended(x, fs)
    p = repet.derive_params(fs)
    n_seg, segs = parallel.extended_plan(len(x), p.seg_len_samples, p.seg_step_samples)
    assert n_seg == _native.lib().repet_extended_segment_count(len(x), p) and n_seg >= 4
    run = parallel._engine_extended_range(0)
    total = np.zeros(x.shape, dtype=np.float32)
    for first, count in parallel.segment_ranges(n_seg, 3):
        lo, hi = parallel.segment_window(segs, first, count)
        assert hi - lo < 0.6 * len(x)
        window = torch.from_numpy(x[lo:hi].astype(np.float32)).to("cuda:0")
        total[lo:hi] += run(window, fs, first, count, len(x), lo).cpu().numpy()          # device tensor in, device tensor out
        host = run(x[lo:hi], fs, first, count, len(x), lo)                                 # the host-array form of the same
        assert np.array_equal(host, run(window, fs, first, count, len(x), lo).cpu().numpy().astype(np.float64))
    # one fp32 rounding of difference where two ranks' windows overlap (the single-GPU accumulation is a fused multiply-add)
    assert rms_err(total, want) < 1e-7 and np.max(np.abs(total - want)) < 1e-6
    ctx = repet.Context(0)
    ctx.upload(x[:len(x) // 2])
    ctx.set_window(len(x), 0)
    with pytest.raises(ValueError):
        ctx.execute_extended_range(p, n_seg - 1, 1)                 # the last segment's samples are not in this window
    with pytest.raises(ValueError):
        ctx.execute("original", p)                                  # a window is only good for segment ranges
    ctx.close()


@pytest.mark.parametrize("seg_len,seg_step,fs,channels,seconds", [(5.0, 1.25, 44100, 3, 24.65), (8.0, 2.0, 16000, 2, 31.0),
                                                                  (10.0, 7.5, 22050, 1, 36.0), (6.0, 1.0, 8000, 2, 23.0)])
def test_extended_with_other_overlaps(seg_len, seg_step, fs, channels, seconds, monkeypatch):
    """The reference cross-fades in place, so with a step shorter than the overlap the fades of several later
    segments compound on one sample (found by tools/fuzz_parity.py); a step longer than the overlap leaves gaps
    of weight 1. Both against the oracle, and as the sum of segment ranges."""
    monkeypatch.setattr(repet, "segment_length", seg_len)
    monkeypatch.setattr(repet, "segment_step", seg_step)
    x = synth(seconds, fs, channels, 77)
    prm = orc.Params(segment_length=seg_len, segment_step=seg_step)
    want = orc.extended(x, fs, prm)
    got = repet.extended(x, fs)
    assert rms_err(got, want) <= 2e-5
    p = repet.derive_params(fs)
    n_seg = _native.lib().repet_extended_segment_count(len(x), p)
    assert n_seg == len(orc.extended_plan(len(x), fs, prm)[0]) and n_seg >= 3
    ctx = repet.Context(0)
    ctx.upload(x)
    total = np.zeros_like(want)
    for first, count in parallel.segment_ranges(n_seg, 3):
        ctx.execute_extended_range(p, first, count)
        total += ctx.download()
    ctx.close()
    assert rms_err(total, want) <= 2e-5


@pytest.mark.parametrize("algo", ALGOS)
def test_resident_batch_of_clips(algo):
    """repet_ctx_upload_batch: equal-shape clips resident together. simonline runs every stage once over all of
    them (BASELINE.json configs[4] is 64 such clips); the other variants loop over the resident clips. Either way
    clip k of the result is bit-identical to a single-clip run."""
    fs, channels = 16000, 2
    clips = np.stack([synth(13.0, fs, channels, 300 + k) for k in range(5)])
    p = repet.derive_params(fs)
    single = []
    ctx = repet.Context(0)
    for clip in clips:
        ctx.upload(clip)
        ctx.execute(algo, p)
        single.append(ctx.download())
    ctx.upload_batch(clips)
    ctx.execute(algo, p)
    got = ctx.download()
    assert got.shape == clips.shape
    for k in range(len(clips)):
        assert np.array_equal(got[k], single[k]), k
    fg = ctx.foreground()
    assert np.allclose(fg, clips - got, atol=1e-6)
    if algo == "original":
        assert len(ctx.last_periods(16)) == len(clips)          # one period per clip, all stages batched
    with pytest.raises(ValueError):
        ctx.execute_extended_range(p, 0, 1)                     # segment ranges are a single-clip notion
    ctx.upload(clips[0])                       # back to a single clip on the same context
    ctx.execute(algo, p)
    assert np.array_equal(ctx.download(), single[0])
    ctx.close()


@pytest.mark.parametrize("algo", ["original", "sim"])
def test_c_client_matches_the_python_drop_in(algo, tmp_path):
    """examples/c_client.c -- a host with no Python in it -- through the same C ABI gives the same samples."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.dirname(_native.LIB_PATH)
    exe = tmp_path / "c_client"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_client.c"), "-o", str(exe),
                           "-L", lib_dir, "-lrepet_hip", f"-Wl,-rpath,{lib_dir}"])
    fs = 22050
    x = synth(12.0, fs, 2, 91)
    (tmp_path / "in.f64").write_bytes(np.ascontiguousarray(x).tobytes())
    out = subprocess.check_output([str(exe), str(_native.ALGO_IDS[algo]), str(fs), "2", str(tmp_path / "in.f64"), str(tmp_path / "out.f64")])
    assert b"separated 264600 samples x 2 channels" in out
    got = np.frombuffer((tmp_path / "out.f64").read_bytes(), dtype=np.float64).reshape(x.shape)
    assert np.array_equal(got, getattr(repet, algo)(x, fs))


def test_batch_api_matches_single_calls():
    fs = 8000
    clips = [synth(d, fs, 2, s) for d, s in [(4, 1), (7, 2), (5, 3)]]
    outs = repet.run_batch("original", clips, fs, n_devices=1)
    for x, y in zip(clips, outs):
        assert np.array_equal(y, repet.original(x, fs))


@pytest.mark.parametrize("n_devices", [2, 4])
def test_batch_api_deals_over_logical_devices(n_devices, monkeypatch):
    """repet_run_batch with n_devices > 1 on the one-GPU box: REPET_LOGICAL_DEVICES maps logical device d onto physical
    device d % visible, so the dealing (longest first, round-robin), the per-device threads, contexts and streams and the
    placement of the results are the multi-GPU path's. Seven clips of different lengths and channel counts, every
    result bit-identical to its single call; without the switch more devices than GPUs is an argument error."""
    fs = 8000
    clips = [synth(d, fs, c, s) for d, c, s in [(11, 2, 1), (13.5, 1, 2), (12, 2, 3), (16, 2, 4), (11.2, 3, 5), (14, 1, 6), (12.7, 2, 7)]]
    want = [repet.sim(x, fs) for x in clips]
    if _native.lib().repet_device_count() < n_devices:
        with pytest.raises(ValueError):
            repet.run_batch("sim", clips, fs, n_devices=n_devices)
    monkeypatch.setenv("REPET_LOGICAL_DEVICES", str(n_devices))
    outs = repet.run_batch("sim", clips, fs, n_devices=n_devices)
    assert len(outs) == len(clips)
    for y, w in zip(outs, want):
        assert y.shape == w.shape and np.array_equal(y, w)


def test_batch_api_rccl_transport_on_one_device():
    """The in-library xGMI transport (ncclCommInitAll + grouped ncclSend / ncclRecv) with the one device this box
    has: librccl is opened, a communicator is made and destroyed, the clips go through device buffers (upload_device /
    download_device) -- everything but the sends themselves, which need a second GPU."""
    fs = 8000
    clips = [synth(d, fs, 2, s) for d, s in [(5, 1), (8, 2)]]
    outs = repet.run_batch("original", clips, fs, n_devices=1, transport="rccl")
    for x, y in zip(clips, outs):
        assert np.array_equal(y, repet.original(x, fs))


def test_rccl_transport_sends_to_itself_and_carries_the_remainders(monkeypatch):
    """REPET_RCCL_SELF=1: on a one-GPU box every clip of the RCCL transport is sent by device 0 to itself inside the group, so
    ncclSend / ncclRecv / ncclGroupEnd, the rounds and the result path all run. float64 clips travel as two fp32 planes
    (samples + remainders): `sim`, whose peak picking takes its close decisions from float64 spectra of the 48-bit samples,
    must come back BIT-IDENTICAL to repet.sim on the same array, and the remainder planes must have been resident."""
    fs = 22050
    clips = [synth(d, fs, c, s) for d, c, s in [(21, 2, 3), (14, 1, 4), (17, 2, 5)]]
    monkeypatch.setenv("REPET_RCCL_SELF", "1")
    outs = repet.run_batch("sim", clips, fs, n_devices=1, transport="rccl")
    info = repet.last_batch_info()
    assert info == {"transport": "rccl", "clips_sent": 3, "clips_with_remainders": 3, "rccl_groups": 6}, info
    monkeypatch.delenv("REPET_RCCL_SELF")
    for x, y in zip(clips, outs):
        ctx = repet.Context(0)
        ctx.upload(x)
        ctx.execute("sim", repet.derive_params(fs))
        assert ctx.last_exact_stats()["input_has_remainders"] and ctx.last_exact_stats()["rows_exact"] > 0
        assert np.array_equal(y, ctx.download())
        ctx.close()
    # PCM-exact clips send one plane each; the host transport sends nothing
    pcm = [np.round(x * 32768.0).clip(-32768, 32767) / 32768.0 for x in clips[:2]]
    monkeypatch.setenv("REPET_RCCL_SELF", "1")
    outs = repet.run_batch("sim", pcm, fs, n_devices=1, transport="rccl")
    assert repet.last_batch_info() == {"transport": "rccl", "clips_sent": 2, "clips_with_remainders": 0, "rccl_groups": 4}
    monkeypatch.delenv("REPET_RCCL_SELF")
    for x, y in zip(pcm, outs):
        assert np.array_equal(y, repet.sim(x, fs))
    repet.run_batch("original", pcm, fs, n_devices=1, transport="host")
    assert repet.last_batch_info()["clips_sent"] == 0 and repet.last_batch_info()["transport"] == "host"


@pytest.mark.parametrize("depth", [1, 2, 3, 5])
def test_clips_in_flight_on_one_device_equal_the_one_shot_calls(depth, monkeypatch):
    """repet_run_stream (repet.run_batch(..., device=, depth=)): clips one after another through one device with `depth` of them
    in flight, each in a context of its own -- upload, kernels and download of neighbouring clips overlap. Every result must be
    the one-shot call's, bit for bit, whatever the depth, for clips of different lengths, channel counts and dtypes (float64
    with remainders, PCM-exact, float32), for every variant; a second call reuses the pooled contexts; a refused clip reports
    its error and leaves the pool usable."""
    fs = 16000
    clips = [synth(d, fs, c, s) for d, c, s in [(21, 2, 3), (14, 1, 4), (17, 2, 5), (13, 2, 6), (25, 1, 7), (12, 2, 8), (19, 2, 9)]]
    clips[3] = np.round(clips[3] * 32768.0).clip(-32768, 32767) / 32768.0
    for algo in ("sim", "original", "simonline"):            # (simonline: every clip longer than its 10-s buffer)
        outs = repet.run_batch(algo, clips, fs, device=0, depth=depth)
        for x, y in zip(clips, outs):
            assert np.array_equal(y, getattr(repet, algo)(x, fs)), algo
    f32 = [c.astype(np.float32) for c in clips[:3]]
    for x, y in zip(f32, repet.run_batch("adaptive", f32, fs, device=0, depth=depth)):
        assert y.dtype == np.float64 and np.array_equal(y, repet.adaptive(x, fs))
    assert repet.run_batch("sim", [], fs, device=0, depth=depth) == []
    bad = clips[1].copy()
    bad[5000, 0] = np.nan
    with np.errstate(all="ignore"):
        got = repet.run_batch("sim", [clips[0], bad, clips[2]], fs, device=0, depth=depth)
    assert np.array_equal(got[1], repet.sim(bad, fs), equal_nan=True) and np.isnan(got[1]).any() and np.array_equal(got[2], repet.sim(clips[2], fs))
    monkeypatch.setattr(repet, "strict_reference", False)
    with pytest.raises(ValueError, match="NaN or infinite"):
        repet.run_batch("sim", [clips[0], bad, clips[2]], fs, device=0, depth=depth)
    assert np.array_equal(repet.run_batch("sim", clips[:2], fs, device=0, depth=depth)[1], repet.sim(clips[1], fs))
    with pytest.raises(ValueError):
        repet.run_batch("sim", clips[:2], fs, device=0, depth=9)
    repet.release_workspaces()


@pytest.mark.parametrize("algo", ["simonline", "sim", "original", "extended"])
def test_samples_that_are_not_finite_give_the_same_result_on_every_transport(algo, monkeypatch):
    """The reference's behaviour on NaN / infinite samples (the default) must not depend on HOW a clip reached the engine: the
    one-shot call scans the host array on its way into the pinned ring, the RCCL transport and device-tensor uploads hand over
    device planes that nobody scanned (round 5's advisor: the passes that reproduce repet.py -- NaN frames for infinite samples,
    `simonline`'s cleared warm-up spectra, 0 x NaN -- were keyed on the host scan and skipped there). NaN inside `simonline`'s
    first buffer_length seconds, an infinite sample later; `extended` has its longer last segment on the auxiliary context."""
    fs = 16000
    x = synth(36 if algo == "extended" else 24, fs, 2, 17)
    n = len(x)
    x[3 * fs + 123, 0] = np.nan                             # inside simonline's warm-up (10-s buffer)
    x[(3 * n) // 4, 1] = np.inf
    x[n - 2000, 0] = np.nan                                 # the last segment of `extended`
    with np.errstate(all="ignore"):
        want = orc.ALGORITHMS[algo](x, fs)
    one_shot = getattr(repet, algo)(x, fs)
    if algo in ("sim", "simonline"):
        assert np.array_equal(np.isnan(one_shot), np.isnan(want))
    else:
        assert not (np.isnan(want) & ~np.isnan(one_shot)).any()         # (infinite samples: a superset)
    monkeypatch.setenv("REPET_RCCL_SELF", "1")
    sent = repet.run_batch(algo, [x], fs, n_devices=1, transport="rccl")[0]
    assert repet.last_batch_info()["clips_sent"] == 1
    monkeypatch.delenv("REPET_RCCL_SELF")
    assert np.array_equal(sent, one_shot, equal_nan=True)
    host = repet.run_batch(algo, [x], fs, n_devices=1, transport="host")[0]
    assert np.array_equal(host, one_shot, equal_nan=True)
    # device planes handed to a context (what a torch.distributed worker does with a received tensor)
    import torch
    hi = x.astype(np.float32)
    lo = (x - hi.astype(np.float64)).astype(np.float32)
    lo[~np.isfinite(lo)] = 0.0
    t_hi, t_lo = torch.from_numpy(hi).cuda(), torch.from_numpy(lo).cuda()
    torch.cuda.synchronize()
    ctx = repet.Context(0)
    ctx.upload_device(t_hi.data_ptr(), n, 2, 1, t_lo.data_ptr())
    ctx.execute(algo, repet.derive_params(fs))
    got = ctx.download()
    ctx.close()
    assert np.array_equal(got, one_shot, equal_nan=True)
    # and the refusal as the option, on the RCCL transport too
    monkeypatch.setattr(repet, "strict_reference", False)
    with pytest.raises(ValueError, match="NaN or infinite"):
        repet.run_batch(algo, [x], fs, n_devices=1, transport="rccl")


@pytest.mark.slow
def test_sim_headline_config_properties():
    """cfg 2 (180 s, 44.1 kHz stereo): strided golden samples + size-independent properties."""
    g = load_golden("cfg2_sim")
    x, fs = golden_input("cfg2_sim")
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("sim", p)
    y = ctx.download()
    t = ctx.last_frame_count()
    idx, cnt = ctx.last_sim_indices(t, p.sim_number)
    stats = ctx.last_refine_stats()
    ctx.close()
    stride = int(g["sample_stride"])
    assert rms_err(y[::stride], g["sim.samples"]) <= 2e-5            # bar 1e-4; decisions equal => fp32 arithmetic only
    # the reference's own similar-frame lists (every 16th frame) and list lengths (every frame); the top-100 cut is
    # active on most rows here, so this covers the float64 re-ranking of the cut as well
    assert np.array_equal(cnt, g["sim.counts"])
    differ = sum(set(idx[f, :cnt[f]].tolist()) != set(row[row >= 0].tolist())
                 for row, f in zip(g["sim.indices"], g["sim.index_frames"]))
    assert differ == 0, (differ, stats)
    assert stats["flat_rows"] == 0 and stats["decisions_changed"] > 0
    n = (len(y) // fs) * fs
    per_s = np.sqrt(np.mean(y[:n].reshape(-1, fs, 2) ** 2, axis=1))
    assert np.max(np.abs(per_s - g["sim.rms_per_second"])) < 2e-4
    # the soft mask never amplifies: background energy <= mixture energy per second (COLA-exact STFT)
    mix = np.sqrt(np.mean(np.array(x[:n]).reshape(-1, fs, 2) ** 2, axis=1))
    assert np.all(per_s <= mix * 1.001 + 1e-6)


@pytest.mark.slow
def test_sim_headline_size_on_the_second_clip_family():
    """cfg-2 size (180 s, 44.1 kHz stereo, T = 7 753, rank-domain median, 256 x 256 Gram tiles) on the groove clip: the
    reference's strided samples and NaN positions (the silent bar), its list lengths on EVERY row, its lists on every
    16th row, `original`'s period, per-second RMS."""
    g = load_golden("cfg2_groove")
    x, fs = golden_input("cfg2_groove")
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("sim", p)
    y = ctx.download()
    t = ctx.last_frame_count()
    idx, cnt = ctx.last_sim_indices(t, p.sim_number)
    stats, exact = ctx.last_refine_stats(), ctx.last_exact_stats()
    stride = int(g["sample_stride"])
    ref = g["sim.samples"]
    assert np.isnan(ref).any() and np.array_equal(np.isnan(y[::stride]), np.isnan(ref))
    assert rms_err(y[::stride][~np.isnan(ref)], ref[~np.isnan(ref)]) <= 2e-5
    assert np.array_equal(cnt, g["sim.counts"])
    differ = sum(set(idx[f, :cnt[f]].tolist()) != set(row[row >= 0].tolist())
                 for row, f in zip(g["sim.indices"], g["sim.index_frames"]))
    assert differ == 0, (differ, stats, exact)
    n = (len(y) // fs) * fs
    per_s = np.sqrt(np.mean(y[:n].reshape(-1, fs, 2) ** 2, axis=1))
    want_s = g["sim.rms_per_second"]
    assert np.array_equal(np.isnan(per_s), np.isnan(want_s))
    assert np.max(np.abs(per_s[~np.isnan(want_s)] - want_s[~np.isnan(want_s)])) < 2e-4
    ctx.execute("original", p)
    assert ctx.last_periods(1)[0] == int(g["original.period"])
    assert rms_err(ctx.download()[::stride], g["original.samples"]) <= RMS_TOL
    ctx.close()


@pytest.mark.parametrize("fs,channels,seconds", [(4000, 1, 24), (8000, 3, 14), (22050, 4, 13), (96000, 1, 11),
                                                  (48000, 2, 12), (11025, 5, 12), (44100, 16, 12), (16000, 26, 13)])
@pytest.mark.parametrize("algo", ALGOS)
def test_other_rates_and_channel_counts(algo, fs, channels, seconds):
    """Window lengths 256..4096, 1-26 channels (block and per-channel kernel paths; 16 channels at W = 2048 and 26 at
    W = 1024 are more than one workgroup of the fused inverse STFT holds: channel groups), every variant. The reference
    loops over any number of channels (repet.py:152, :179, :510, :543)."""
    x = synth(seconds, fs, channels, 40 + channels)
    if algo in ("sim", "simonline"):
        assert assert_parity_modulo_near_ties(algo, x, fs).branch == "strict"       # plain RMS <= 1e-4, no tie allowance
        return
    got = getattr(repet, algo)(x, fs)
    want = orc.ALGORITHMS[algo](x, fs)
    assert got.shape == x.shape
    assert rms_err(got, want) <= RMS_TOL, f"rms {rms_err(got, want):.3e}"


def test_fft_paths_agree():
    """REPET_FFT_PATH picks the STFT/iSTFT kernels: "reg" (default; one wavefront per transform, data in registers,
    W = 2048 only), "block" (workgroup Stockham in LDS) and "wave" (wave-synchronous Stockham). All must agree."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "x = synth(6, 44100, 2, 3); y = repet.original(x, 44100); z = repet.sim(x, 44100); "
            "m = synth(6, 48000, 1, 4); u = repet.adaptive(m, 48000); q = synth(5, 44100, 4, 5); r = repet.extended(q, 44100);"
            "np.save(sys.argv[1], np.concatenate([y.ravel(), z.ravel(), u.ravel(), r.ravel()]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)
    outs = []
    for path in ("block", "wave", "reg"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_fft_{path}_{os.getpid()}.npy")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_FFT_PATH=path))
        outs.append(np.load(out))
        os.remove(out)
    assert rms_err(outs[0], outs[1]) < 2e-6
    assert rms_err(outs[0], outs[2]) < 2e-6


def test_wide_mask_kernel_gives_the_same_bits():
    """Lists of at most 32 similar frames (`simonline`, short clips of `sim`) go through mask_sim_wide_kernel: four bins per lane,
    one 16-byte gather per list slot. REPET_MASK_WIDE=0 keeps the one-bin-per-lane kernel: the same values through the same
    selection networks, so every output must be IDENTICAL -- in place and with the mask as a plane, a batch context included."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth, synth_groove; "
            "x = synth(31, 44100, 2, 3); outs = [repet.simonline(x, 44100), repet.sim(x[:20 * 44100], 44100)]; "
            "g = synth_groove(26, 16000, 2, 2); outs += [repet.simonline(g, 16000), repet.sim(g, 16000)]; "
            "m = synth(14, 22050, 1, 8); outs += [repet.sim(m, 22050)]; "
            "c = repet.Context(0); c.upload_batch(np.stack([synth(13, 16000, 2, s) for s in range(3)])); c.execute('simonline', repet.derive_params(16000)); "
            "outs.append(c.download()); np.save(sys.argv[1], np.concatenate([o.ravel() for o in outs]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)
    outs = []
    for wide, plane in (("1", ""), ("0", ""), ("1", "0"), ("0", "0")):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_wide_{wide}{plane}_{os.getpid()}.npy")
        env = dict(os.environ, REPET_MASK_WIDE=wide)
        if plane:
            env["REPET_MASK_PLANE"] = plane
        subprocess.check_call([sys.executable, "-c", code, out], env=env)
        outs.append(np.load(out))
        os.remove(out)
    for other in outs[1:]:
        assert np.array_equal(outs[0], other, equal_nan=True)


def test_mask_plane_gives_the_same_bits():
    """Three ways to apply the soft mask, one result. (a) multiplied into the spectrum in place (REPET_MASK_PLANE=0);
    (b) kept as a plane of its own and applied by the inverse STFT while it fetches the spectrum (REPET_MASK_PLANE=p);
    (c) original / extended on the register inverse STFT (REPET_MASK_PLANE=1, their default): the mask kernel writes only the
    repeating-segment model [period][F] and the inverse STFT computes soft_mask(V, model[t mod period]) itself. The same
    rounded products every way, so every variant's output must be IDENTICAL -- batched extended segments, a batch
    context of simonline clips, a batch context of original clips, a mono clip and the 4-channel block-kernel path included."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "x = synth(31, 44100, 2, 3); outs = [getattr(repet, a)(x, 44100) for a in ('original', 'extended', 'adaptive', 'sim', 'simonline')]; "
            "m = synth(47, 44100, 1, 8); outs += [repet.original(m, 44100), repet.extended(m, 44100)]; "
            "q = synth(12, 22050, 4, 5); outs += [repet.extended(q, 22050), repet.simonline(q, 22050)]; "
            "c = repet.Context(0); c.upload_batch(np.stack([synth(13, 16000, 2, s) for s in range(3)])); c.execute('simonline', repet.derive_params(16000)); "
            "outs.append(c.download()); c.execute('original', repet.derive_params(16000)); outs.append(c.download()); "
            "np.save(sys.argv[1], np.concatenate([o.ravel() for o in outs]))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)
    outs = []
    for plane in ("0", "p", "1"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_plane_{plane}_{os.getpid()}.npy")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_MASK_PLANE=plane))
        outs.append(np.load(out))
        os.remove(out)
    assert np.array_equal(outs[0], outs[2], equal_nan=True) and np.array_equal(outs[1], outs[2], equal_nan=True)
    assert np.array_equal(outs[0], outs[2], equal_nan=True)


def test_gram_paths_agree():
    """sim's similarity matrix: the f16-split matrix-core kernel (default) against the exact-fp32 one (REPET_GRAM=f32).
    Same similar-frame lists (the float64 refinement settles every near-tie either way), outputs equal to fp32 noise."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "x = synth(30, 44100, 2, 21); p = repet.derive_params(44100); c = repet.Context(0); c.upload(x); c.execute('sim', p); "
            "y = c.download(); idx, cnt = c.last_sim_indices(c.last_frame_count(), p.sim_number); "
            "np.savez(sys.argv[1], y=y, idx=np.sort(idx, axis=1), cnt=cnt)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)
    outs = []
    for path in ("f16", "f32"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_gram_{path}_{os.getpid()}.npz")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_GRAM=path))
        with np.load(out) as z:
            outs.append({k: z[k] for k in z.files})
        os.remove(out)
    assert np.array_equal(outs[0]["cnt"], outs[1]["cnt"])
    assert np.array_equal(outs[0]["idx"], outs[1]["idx"])
    assert rms_err(outs[0]["y"], outs[1]["y"]) < 5e-6


def test_gram_tile_sizes_give_the_same_matrix():
    """The full similarity matrix on 256 x 256 tiles with LDS-DMA staging (gram_f16_big.hip, default from 2 048 frames on)
    against the 128 x 128-tile kernel (REPET_GRAM_TILE=128): the same three f16 products per term in the same order, so
    the MATRIX is bit-identical -- checked on the stage export (exact symmetry, NaN row and column of a silent frame
    included) and end to end."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "rs = np.random.RandomState(5); v = np.abs(rs.standard_normal((513, 2300))); v[:, 700] = 0.0; "
            "s = repet._selfsimilaritymatrix(v); x = synth(60, 44100, 2, 21); y = repet.sim(x, 44100); "
            "np.savez(sys.argv[1], s=s, y=y)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)
    outs = []
    for tile in ("256", "128"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_tile_{tile}_{os.getpid()}.npz")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_GRAM_TILE=tile))
        with np.load(out) as z:
            outs.append({k: z[k] for k in z.files})
        os.remove(out)
    s = outs[0]["s"]
    assert np.array_equal(s, outs[1]["s"], equal_nan=True)
    assert np.array_equal(s, s.T, equal_nan=True)
    assert np.all(np.isnan(s[700])) and np.all(np.isnan(s[:, 700])) and np.isnan(s).sum() == 2 * 2300 - 1
    assert np.array_equal(outs[0]["y"], outs[1]["y"])


@pytest.mark.parametrize("seconds,fs,channels,number,distance", [(50, 44100, 2, 100, 1.0), (110, 22050, 1, 100, 0.3),
                                                                  (70, 16000, 3, 64, 0.2), (30, 44100, 2, 31, 0.1),
                                                                  (40, 44100, 2, 128, 0.05), (100, 8000, 4, 101, 0.1)])
def test_median_paths_agree_bit_for_bit(seconds, fs, channels, number, distance):
    """sim's median on clips of more than 1 024 frames: the bit-sliced selection on the rank codes (default: rank.hip +
    mask_bits.hip) and the packed 16-bit selection network on the same codes (REPET_MEDIAN=rank) against the selection on
    the float magnitudes (REPET_MEDIAN=f32). A median is a selection, so the three must agree BIT FOR BIT, not to a
    tolerance -- odd (31, 101) and even (64, 100, 128) list lengths, short lists, both list halves of the bit-sliced
    kernel (50 and 64 entries per wave), 11 to 13 code planes, 1-4 channels (8, 16, 24 and 32 blocks of 64 bins)."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "repet.similarity_number = {number}; repet.similarity_distance = {distance}; "
            "x = synth({seconds}, {fs}, {channels}, 33); p = repet.derive_params({fs}); c = repet.Context(0); c.upload(x); "
            "tm = c.execute('sim', p, timing=True); y = c.download(); _, cnt = c.last_sim_indices(c.last_frame_count(), p.sim_number); "
            "np.savez(sys.argv[1], y=y, cnt=cnt, stages=np.array([s['name'] for s in tm['stages']]), path=np.array(c.last_median_path()))")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (code % (os.path.join(root, "repet-python_amd"), root)).format(number=number, distance=distance, seconds=seconds, fs=fs, channels=channels)
    outs = []
    for path in ("bits", "rank", "f32"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_median_{path}_{os.getpid()}.npz")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_MEDIAN=path))
        with np.load(out) as z:
            outs.append({k: z[k] for k in z.files})
        os.remove(out)
        assert str(outs[-1]["path"]) == path
    assert any("rank_columns" in n for n in outs[0]["stages"].tolist()) and not any("rank_columns" in n for n in outs[2]["stages"].tolist())
    assert np.array_equal(outs[1]["y"], outs[2]["y"])
    assert np.array_equal(outs[0]["y"], outs[2]["y"])


@pytest.mark.parametrize("variant,seconds,fs,channels,number,distance", [
    ("sim", 50, 44100, 2, 100, 1.0), ("sim", 30, 44100, 2, 31, 0.1), ("sim", 40, 44100, 1, 128, 0.05), ("sim", 100, 8000, 3, 101, 0.1),
    ("sim", 12, 16000, 2, 7, 1.0), ("simonline", 30, 44100, 2, 100, 1.0), ("simonline", 20, 16000, 2, 5, 0.3)])
def test_nyquist_bin_kernels_agree_bit_for_bit(variant, seconds, fs, channels, number, distance):
    """Bin F - 1 of `sim` / `simonline` (the one bin outside the 64-bin blocks of the selection kernels): the wave-per-frame
    kernel of round 6 (rank by counting, the network's pad slots as counts) against the lane-per-frame kernel that runs the
    selection network itself (REPET_NYQUIST=lane) -- odd and even lists, lists shorter than their network (pads on both
    sides), more than 64 entries (both registers of a lane), one to three channels. A selection: the same bits."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "repet.similarity_number = {number}; repet.similarity_distance = {distance}; "
            "x = synth({seconds}, {fs}, {channels}, 41); y = repet.{variant}(x, {fs}); np.save(sys.argv[1], y)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (code % (os.path.join(root, "repet-python_amd"), root)).format(number=number, distance=distance, seconds=seconds, fs=fs,
                                                                          channels=channels, variant=variant)
    outs = []
    for path in ("lists", "wave", "lane"):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_nyquist_{path}_{os.getpid()}.npy")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, REPET_NYQUIST=path))
        outs.append(np.load(out))
        os.remove(out)
    assert np.array_equal(outs[0], outs[2], equal_nan=True) and np.array_equal(outs[1], outs[2], equal_nan=True)


def test_round_six_switches_give_the_same_result():
    """The A/B switches of round 6 select other kernels / schedules for the same arithmetic: the 64-frame transposes of the rank
    chain, the five-stage trips of the column sort, the separate f16-split pass, the forked Nyquist-bin kernel, the Gram
    kernel's staggered start, plain host stores, the mask multiplied in place -- `sim` on a clip long enough for the big-tile
    Gram kernel and the rank-domain median must come out BIT FOR BIT as with the defaults. The float64 norms of the unit rows on
    a table change level-1 values in their last bits (never a decision that is not re-taken at level 2): the lists' lengths
    stay, the signal within 1e-6."""
    import os
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = [%r, %r]; import repet; from repet_synth import synth; "
            "x = synth(52, 44100, 2, 77); p = repet.derive_params(44100); c = repet.Context(0); c.upload(x); c.execute('sim', p); "
            "y = c.download(); _, cnt = c.last_sim_indices(c.last_frame_count(), p.sim_number); np.savez(sys.argv[1], y=y, cnt=cnt)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "repet-python_amd"), root)

    def run(env):
        out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"repet_r6_switch_{os.getpid()}.npz")
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, **env))
        with np.load(out) as z:
            got = {k: z[k] for k in z.files}
        os.remove(out)
        return got

    base = run({})
    assert np.all(np.isfinite(base["y"])) and base["cnt"].min() >= 1
    for env in ({"REPET_RANK_TILE": "64"}, {"REPET_RANK_TRIPS": "5"}, {"REPET_SPLIT_IN_STFT": "0"}, {"REPET_NYQUIST_STREAM": "side"},
                {"REPET_GRAM_STAGGER_US": "5", "REPET_GRAM_STAGGER_GROUPS": "4"}, {"REPET_HOST_NT": "0"}, {"REPET_MASK_PLANE": "0"}):
        got = run(env)
        assert np.array_equal(got["y"], base["y"]), env
        assert np.array_equal(got["cnt"], base["cnt"]), env
    got = run({"REPET_PEAK_NORMS": "1"})
    assert np.array_equal(got["cnt"], base["cnt"]) and rms_err(got["y"], base["y"]) <= 1e-6


def test_long_similarity_number_uses_bisection_path():
    """similarity_number > 128 takes the bisection median (no sorting network of that size)."""
    x, fs = golden_input("small_stereo")
    saved = (repet.similarity_number, repet.similarity_distance)
    try:
        repet.similarity_number = 300
        repet.similarity_distance = 0.005         # 0 frames: every frame is a candidate, the 300 most similar are kept
        assert assert_parity_modulo_near_ties("sim", x, fs, dict(similarity_number=300, similarity_distance=0.005)).branch == "strict"
        ctx = repet.Context(0)
        ctx.upload(x)
        ctx.execute("sim", repet.derive_params(fs))
        _, cnt = ctx.last_sim_indices(ctx.last_frame_count(), 300)
        ctx.close()
        assert cnt.max() > 128                    # the bisection median really ran
    finally:
        repet.similarity_number, repet.similarity_distance = saved


def test_two_host_threads_with_their_own_contexts():
    """A repet_ctx serialises its own work; different contexts may run from different host threads
    (ctypes releases the GIL during the call)."""
    import threading
    fs = 16000
    clips = [synth(9, fs, 2, 70 + i) for i in range(4)]
    want = [repet.sim(x, fs) for x in clips]
    got = [None] * 4
    errors = []

    def work(ids):
        try:
            ctx = repet.Context(0)
            p = repet.derive_params(fs)
            for _ in range(3):
                for i in ids:
                    ctx.upload(clips[i])
                    ctx.execute("sim", p)
                    got[i] = ctx.download()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(ids,)) for ids in ([0, 2], [1, 3])]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_contexts_overlapping_on_the_device_do_not_disturb_each_other():
    """Pipelines of different contexts interleave on the device: the forward STFT of one context runs beside the f16-split similarity
    kernels of another. Built with packed-fp32 VALU ops the FFT kernels lost about one result in five here (DESIGN.md
    "Contexts and concurrency", tools/pk_pairs.py); the shipped build must lose none."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "pk_pairs.py"), "600", "stft:selfsim"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "0 of 600 differ" in out.stdout, out.stdout


def test_long_similarity_rows_are_picked_in_segments():
    """Rows longer than one workgroup's LDS (about 8 600 frames) are cut into segments with a halo of the similarity
    distance; a second kernel ranks the per-segment candidates. 310 s at 8 kHz = 9 687 frames = two segments."""
    fs = 8000
    x = synth(310.0, fs, 1, 17)
    tr = orc.Trace()
    want = orc.sim(x, fs, None, tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("sim", p)
    got = ctx.download()
    t = ctx.last_frame_count()
    assert t == len(theirs) and t > 9000
    idx, cnt = ctx.last_sim_indices(t, p.sim_number)
    ctx.close()
    differ = sum(set(idx[r, :cnt[r]].tolist()) != set(np.asarray(theirs[r]).tolist()) for r in range(t))
    assert differ == 0, differ
    ok = ~np.isnan(want)
    assert rms_err(got[ok], want[ok]) <= 2e-5


def test_size_limits_and_bad_arguments_raise():
    fs = 44100
    x = synth(4, fs, 2, 1)
    ctx = repet.Context(0)
    ctx.upload(x)
    p = repet.derive_params(fs)
    bad = repet.derive_params(fs)
    bad.window_length = 3000                                  # not a power of two
    with pytest.raises(RuntimeError):
        ctx.execute("sim", bad)
    bad = repet.derive_params(fs)
    bad.step_length = 512                                     # must be W/2
    with pytest.raises(ValueError):
        ctx.execute("sim", bad)
    with pytest.raises(ValueError):
        _native.check(_native.lib().repet_ctx_execute(ctx.handle, 17, p, None))   # unknown algorithm
    with pytest.raises(ValueError):
        repet.Context(99)                                                         # no such device
    ctx.execute("sim", p)                                     # the context is still usable afterwards
    assert np.all(np.isfinite(ctx.download()))
    ctx.close()
    # empty clip, as the reference behaves: sim/adaptive return an empty array, the period-based ones raise
    assert repet.sim(np.zeros((0, 2)), fs).shape == (0, 2)
    assert repet.adaptive(np.zeros((0, 2)), fs).shape == (0, 2)
    for algo in ("original", "extended", "simonline"):
        with pytest.raises(ValueError):
            getattr(repet, algo)(np.zeros((0, 2)), fs)


def test_foreground_and_spectrograms_on_device():
    """README.md:64-98 workflow kept on the device: foreground = audio - background, and the three
    spectrograms abs(_stft(mean over channels)) for specshow."""
    x, fs = golden_input("small_stereo")
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("original", p)
    bg = ctx.download()
    fg = ctx.foreground()
    assert rms_err(fg, np.asarray(x, dtype=np.float32).astype(np.float64) - bg) < 1e-7
    w, window, h = orc.stft_geometry(fs)
    f = w // 2 + 1
    for which, sig in (("mixture", np.array(x)), ("background", bg), ("foreground", fg)):
        got = ctx.spectrogram(which, w)
        want = np.abs(orc.stft(np.mean(sig, axis=1), window, h)[:f])
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) < 2e-5 * max(1.0, want.max())
    ctx.close()


@pytest.mark.parametrize("fs,channels,seconds,seed", [(8000, 2, 16, 2), (16000, 1, 14, 8), (44100, 2, 13, 4), (11025, 3, 14, 6),
                                                       (22050, 6, 13, 7)])
def test_streaming_online_equals_offline_simonline(fs, channels, seconds, seed):
    """push() in arbitrary chunks + finish() reproduces repet.simonline of the whole signal exactly."""
    x = synth(seconds, fs, channels, seed)
    want = repet.simonline(x, fs)
    rs = np.random.RandomState(seed)
    stream = repet.online(fs, channels)
    pieces, pos = [], 0
    sizes = [1, 7, 255, 256, 257, 4096, 1000, 30000]
    while pos < len(x):
        n = min(int(sizes[rs.randint(len(sizes))] * (1 + rs.rand())), len(x) - pos)
        pieces.append(stream.push(x[pos:pos + n]))
        pos += n
        # a hop is only emitted once its frame is complete: never ahead of the input
        assert sum(len(p) for p in pieces) <= pos
    pieces.append(stream.finish())
    stream.close()
    got = np.concatenate(pieces, axis=0)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    # and against the oracle, like every other variant: the plain 1e-4 bar
    assert assert_parity_modulo_near_ties("simonline", x, fs).branch == "strict"


def test_streaming_online_errors():
    fs = 8000
    x = synth(5, fs, 2, 3)                       # shorter than the 10-s buffer
    stream = repet.online(fs, 2)
    assert len(stream.push(x)) == 0 or np.all(stream.push(x[:0]) == 0)
    with pytest.raises(ValueError):
        stream.finish()                          # the reference raises for such a clip (repet.py:802)
    stream.close()
    with pytest.raises(ValueError):
        repet.online(fs, 0)                      # at least one channel (any number works: the streaming test runs 3 and 6)
    stream = repet.online(fs, 1)
    with pytest.raises(ValueError):
        stream.push(np.zeros((10, 2)))
    stream.close()


@pytest.mark.parametrize("case", ["cfg1_audio_file", "cfg1_surrogate"])
@pytest.mark.parametrize("algo", ALGOS)
def test_reference_example_clip(algo, case):
    """BASELINE.json configs[0]: the reference's own example clip (real music, README.md:62-75), replayed from its
    int16 PCM through wavread's normalisation, against the reference's outputs and integer intermediates -- where the
    reference tree is (the clip cannot be redistributed: skipped on the GPU box); and its SURROGATE, which runs everywhere:
    repet_synth.synth_song at the clip's exact shape (1 014 301 samples, 44.1 kHz stereo, 16-bit PCM values, T = 992 frames,
    `original`'s period 287 against the clip's 286), produced-music-like content (drum loop with humanised levels, bass,
    pad, a sung line, intro / breakdown / fade), all five variants against the reference's own outputs on it
    (tests/golden/make_golden.py --cases cfg1s)."""
    g = load_golden(case)
    x, fs = golden_input(case)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    y = ctx.download()
    stride = int(g["sample_stride"])
    assert y.shape == (1014301, 2)
    assert rms_err(y[::stride], g[f"{algo}.samples"]) <= RMS_TOL
    n = (len(y) // fs) * fs
    per_s = np.sqrt(np.mean(y[:n].reshape(-1, fs, 2) ** 2, axis=1))
    assert np.max(np.abs(per_s - g[f"{algo}.rms_per_second"])) < 3e-4
    if algo == "original":
        assert ctx.last_periods(1)[0] == int(g["original.period"]) == (286 if case == "cfg1_audio_file" else 287)
    if algo == "extended":
        assert np.array_equal(ctx.last_periods(16), g["extended.periods"])
    if algo == "adaptive":
        assert np.array_equal(ctx.last_periods(ctx.last_frame_count()), g["adaptive.periods"])
    if algo == "sim":
        t = ctx.last_frame_count()
        idx, cnt = ctx.last_sim_indices(t, p.sim_number)
        assert t == 992 and np.array_equal(cnt, g["sim.counts"])
        differ = sum(set(idx[f, :cnt[f]].tolist()) != set(row[row >= 0].tolist())
                     for row, f in zip(g["sim.indices"], g["sim.index_frames"]))
        assert differ == 0
    if algo == "simonline":
        rows = ctx.last_frame_count() - p.buffer_frames + 1
        idx, cnt = ctx.last_sim_indices(rows, p.sim_number)
        assert np.array_equal(cnt, g["simonline.counts"])
    ctx.close()


# Rows whose list may differ from the float64 oracle's: none. (Round 2 allowed one row of one case, a tie closer in the
# oracle's float64 similarity than the fp32 rounding of the spectra the first refinement starts from; the second level of
# the peak picking -- float64 spectra for verdicts closer than 2.5e-7, peaks_exact.hip / peaks_wave.hip -- decides it.)
@pytest.mark.parametrize("algo,seconds,fs,channels,seed,number", [
    ("sim", 60, 22050, 2, 1, 100), ("sim", 20, 96000, 1, 3, 100), ("sim", 90, 16000, 2, 4, 100),
    ("simonline", 45, 16000, 2, 5, 100), ("simonline", 30, 44100, 1, 6, 100),
    ("sim", 60, 22050, 2, 7, 12), ("sim", 50, 16000, 1, 8, 5), ("simonline", 40, 16000, 2, 9, 4),    # top-k cut active
    ("sim", 70, 22050, 2, -1, 100), ("sim", 45, 44100, 2, -2, 100), ("simonline", 40, 16000, 2, -3, 100),
    ("sim", 60, 16000, 1, -4, 8)])                                                    # seed < 0: synth_groove(-seed)
def test_similar_frame_lists_are_the_float64_references(algo, seconds, fs, channels, seed, number, monkeypatch):
    """The discrete half of REPET-SIM: with plain fp32 similarities 2-4 % of the rows pick another frame at a
    near-tie (tools/refine_probe.py); with the float64 near-tie refinement the lists match the oracle's. Both clip
    families (the groove clips carry a silent bar: NaN rows in the similarity matrix, NaN frames in the output)."""
    monkeypatch.setattr(repet, "similarity_number", number)
    x = synth(seconds, fs, channels, seed) if seed >= 0 else synth_groove(seconds, fs, channels, -seed)
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, orc.Params(similarity_number=number), tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    assert p.sim_number == number
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    stats, exact = ctx.last_refine_stats(), ctx.last_exact_stats()
    ctx.close()
    differ, named = list_difference_gaps(algo, tr, [idx[r, :cnt[r]] for r in range(len(theirs))], p)
    assert differ == 0, (differ, len(theirs), named, stats, exact)
    assert stats["flat_rows"] == 0 and (stats["elements_refined"] > 0 or seed < 0)    # (transients: short clips of the
    # second family may hold no near-tie at all)
    # the second level: the synth clips are float64 with bits below fp32 (their remainders travel), a few rows per hundred
    # go through it, and what it measures bounds the band it is triggered by: the largest difference between a float64
    # value of the fp32 spectra and of the float64 spectra stays below half of delta2 = 2.5e-7
    assert exact["input_has_remainders"] and exact["rows_handed_on"] == 0
    assert exact["level2_max_diff"] < 1.25e-7, exact
    ok = ~np.isnan(want)
    assert rms_err(got[ok], want[ok]) <= 2e-5                  # equal lists: fp32 arithmetic is all that is left


def test_similar_frame_lists_with_a_wide_window(monkeypatch):
    """similarity_distance = 12 s: a +-517-frame window spans more than 256 float4 groups of the row, so a thread
    owns several groups of one rival window in the near-tie refinement."""
    monkeypatch.setattr(repet, "similarity_distance", 12.0)
    monkeypatch.setattr(repet, "similarity_number", 128)
    fs = 22050
    x = synth(26.4, fs, 2, 1078)
    tr = orc.Trace()
    want = orc.sim(x, fs, orc.Params(similarity_distance=12.0, similarity_number=128), tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    assert p.sim_distance_frames > 500
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute("sim", p)
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    ctx.close()
    differ = sum(set(idx[r, :cnt[r]].tolist()) != set(np.asarray(theirs[r]).tolist()) for r in range(len(theirs)))
    assert differ == 0, differ
    assert rms_err(got, want) <= 1e-4


_PATHS_SCRIPT = """
import json, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import repet
from repet_synth import synth, synth_groove
out = {}
for algo, seconds, fs, ch, seed in (("sim", 60, 22050, 2, 1), ("simonline", 45, 16000, 2, 5), ("sim", 40, 44100, 2, 0)):
    x = synth(seconds, fs, ch, seed)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    t = ctx.last_frame_count()
    rows = t if algo == "sim" else t - p.buffer_frames + 1
    idx, cnt = ctx.last_sim_indices(rows, p.sim_number)
    ex = ctx.last_exact_stats()
    y = ctx.download()
    ctx.close()
    out[algo + str(fs)] = {"idx": idx.tolist(), "cnt": cnt.tolist(), "exact": ex, "sum": float(np.abs(y).sum())}
print(json.dumps(out))
"""


def _lists_under(env):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _PATHS_SCRIPT % (os.path.join(root, "repet-python_amd"), root)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env={**os.environ, **env}, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_second_level_paths_agree():
    """The second level of the peak picking has a fast path (the wavefront kernel's records + the lean float64 unit-row kernel
    + local_maxima_lite_kernel) and a general one (local_maxima_exact_kernel: rescans the row, no caps). REPET_PEAKS=block
    takes the workgroup kernel as first pass, whose rows all go through the general path: the lists and the audio must be
    the default's, bit for bit. REPET_PEAK_EXACT=0 (first level alone) must still
    produce lists, and go through no float64 spectra."""
    base = _lists_under({})
    for env in ({"REPET_PEAKS": "block"},):
        other = _lists_under(env)
        for key in base:
            assert other[key]["cnt"] == base[key]["cnt"], (env, key)
            assert other[key]["idx"] == base[key]["idx"], (env, key)
            assert other[key]["sum"] == base[key]["sum"], (env, key)
            assert other[key]["exact"]["rows_fast_path"] == 0 and other[key]["exact"]["rows_exact"] > 0
    assert any(base[key]["exact"]["rows_fast_path"] > 0 for key in base)
    off = _lists_under({"REPET_PEAK_EXACT": "0"})
    for key in base:
        assert off[key]["exact"]["rows_exact"] == 0 and off[key]["exact"]["unit_rows_f64"] == 0
        assert off[key]["cnt"] == base[key]["cnt"]


def test_segment_record_paths_agree():
    """Round 4: the first pass of the peak picking takes its candidates from segment records -- written by the epilogue of the
    256 x 256 Gram kernel (clips from 2 048 frames on) or by a pass over the matrix (shorter clips, the other Gram kernels).
    REPET_PEAK_SEGMENTS=0 restores the sweep over every element of the row, REPET_GRAM_SEGMENTS=0 takes the records from the
    pass over the matrix also behind the 256 x 256 kernel: lists, list lengths, second-level statistics and audio must be
    the default's, bit for bit."""
    base = _lists_under({})
    for env in ({"REPET_PEAK_SEGMENTS": "0"}, {"REPET_GRAM_SEGMENTS": "0"}):
        other = _lists_under(env)
        for key in base:
            assert other[key]["cnt"] == base[key]["cnt"], (env, key)
            assert other[key]["idx"] == base[key]["idx"], (env, key)
            assert other[key]["sum"] == base[key]["sum"], (env, key)
            assert other[key]["exact"] == base[key]["exact"], (env, key)


def test_level_one_stays_far_inside_the_second_levels_band():
    """The second level re-takes every verdict whose level-1 values (float64 dot products of the fp32 unit rows) are closer
    than delta2 = 2.5e-7. That band is derived (an error of about sqrt(2 / F) times the relative rms error of the
    fp32 spectrum, sigma ~ 1.2e-8 at F = 1 025, whatever the spectrum's sparsity) and checked here where it could break:
    sparse spectra (a pure tone, a square wave's comb) over a noise floor 100 dB down, 90 dB of level change inside frames,
    a passage just above the silence that gives NaN, 96 kHz with the 4 096-sample window. The largest level-1 / level-2
    difference the device meets (repet_ctx_last_exact_stats) must stay below delta2 / 4."""
    rs = np.random.RandomState(3)
    fs = 44100
    n = 40 * fs
    t = np.arange(n) / fs
    base = synth(40, fs, 2, 5)
    cases = [
        ("tone over -100 dB noise", 0.9 * np.sin(2 * np.pi * 997.0 * t)[:, None] * np.ones((1, 2)) + 1e-5 * rs.standard_normal((n, 2)), fs),
        ("90 dB gate inside frames", base * np.where((t * 7.3) % 1.0 < 0.5, 1.0, 3e-5)[:, None], fs),
        ("a passage at -140 dB", base * np.where((t >= 10) & (t < 20), 1e-7, 1.0)[:, None], fs),
        ("96 kHz", synth(30, 96000, 2, 6), 96000),
        ("square wave comb", np.sign(np.sin(2 * np.pi * 220.0 * t))[:, None] * np.array([[0.99, 0.7]]) + 1e-4 * rs.standard_normal((n, 2)), fs),
    ]
    met = 0
    for name, x, rate in cases:
        ctx = repet.Context(0)
        ctx.upload(np.clip(x, -1.0, 1.0))
        ctx.execute("sim", repet.derive_params(rate))
        ex = ctx.last_exact_stats()
        ctx.close()
        met += ex["rows_exact"]
        assert ex["level2_max_diff"] < 2.5e-7 / 4, (name, ex)
    assert met > 1000                                                      # (the second level really ran on these clips)


def test_staged_upload_splits_float64_exactly():
    """The host side of a float64 upload (hostio.hip): worker threads narrow the samples chunk by chunk and part by part, store
    remainders only from the first one that is not zero on, clear the shares of parts that met none when the chunk travels,
    send the plane behind the samples on its own stream. What arrives must be NumPy's split of the array to the bit -- for a
    clip whose remainders begin in the middle of a chunk and of a worker's part, for one that has them only in its last
    samples, for a PCM-exact one, and after a larger clip has used the same pinned buffers."""
    fs = 44100
    x = synth(42, fs, 2, 9)                                      # 3.7 M elements: four 1-M-element chunks
    pcm = np.round(x * 32768.0) / 32768.0
    cases = []
    mixed = x.copy()
    mixed[:651_217] = pcm[:651_217]                              # remainders from element 1 302 434 on: inside chunk 1
    cases.append(mixed)
    tail = pcm.copy()
    tail[-3:] = x[-3:]
    cases.append(tail)
    cases.append(synth(60, fs, 2, 4))                            # a larger clip in between
    cases.append(pcm)
    gap = x.copy()
    gap[1_000_000:1_600_000] = pcm[1_000_000:1_600_000]          # a PCM-exact stretch inside a noisy clip
    cases.append(gap[: 30 * fs])
    ctx = repet.Context(0)
    for k, clip in enumerate(cases):
        ctx.upload(clip)
        hi, lo, has = ctx.resident_input()
        want_hi, want_lo = parallel.split_float64(clip)
        assert np.array_equal(hi, want_hi), k
        assert has == (want_lo is not None), k
        assert np.array_equal(lo, want_lo if want_lo is not None else np.zeros_like(hi)), k
    ctx.close()


def test_remainders_of_float64_input_travel_only_when_needed():
    """A float64 clip whose samples are exact in fp32 (what wavread yields for PCM files, repet.py:929) uploads no remainders;
    the synth clip (float64 noise) does, and dropping them changes nothing audible (same lists on this clip)."""
    fs = 16000
    x = synth(30.0, fs, 2, 11)
    pcm = np.round(x * 32768.0).clip(-32768, 32767) / 32768.0          # a 16-bit file's samples
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(pcm)
    ctx.execute("sim", p)
    assert not ctx.last_exact_stats()["input_has_remainders"]
    y_pcm = ctx.download()
    ctx.upload(pcm.astype(np.float32))
    ctx.execute("sim", p)
    assert not ctx.last_exact_stats()["input_has_remainders"]
    assert np.array_equal(ctx.download(), y_pcm)                        # the same samples either way
    ctx.upload(x)
    ctx.execute("sim", p)
    assert ctx.last_exact_stats()["input_has_remainders"]
    ctx.close()
    assert rms_err(y_pcm, orc.sim(pcm, fs)) <= 1e-4


@pytest.mark.slow
@pytest.mark.parametrize("case,algo", [("cfg3_extended", "extended"), ("cfg4_adaptive", "adaptive"),
                                        ("cfg5_simonline", "simonline"), ("cfg2_sim", "original")])
def test_full_size_configs_against_reference_goldens(case, algo):
    """BASELINE.json configs at full size: the HIP path against strided samples, per-second RMS and integer
    intermediates of the reference itself (tests/golden/make_golden.py), plus size-independent properties."""
    g = load_golden(case)
    x, fs = golden_input(case)
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    y = ctx.download()
    stride = int(g["sample_stride"])
    n = (len(y) // fs) * fs
    per_s = np.sqrt(np.mean(y[:n].reshape(-1, fs, y.shape[1]) ** 2, axis=1))
    if algo == "simonline":
        # 30-s clips leave <= 10 similar frames per list, so ONE near-tie flip would move a frame's median visibly
        # (plain fp32 decisions: 4 of 861 frames, 5e-4 RMS). With the float64 near-tie refinement the lists are the
        # reference's, so the plain bar holds: RMS <= 1e-4 against the oracle AND against the reference's samples.
        outcome = assert_parity_modulo_near_ties(algo, x, fs)
        assert outcome.branch == "strict" and outcome.rms <= RMS_TOL
        assert rms_err(y[::stride], g[f"{algo}.samples"]) <= RMS_TOL
        assert np.max(np.abs(per_s - g[f"{algo}.rms_per_second"])) < 3e-4
        rows = ctx.last_frame_count() - p.buffer_frames + 1
        _, cnt = ctx.last_sim_indices(rows, p.sim_number)
        assert np.array_equal(cnt, g["simonline.counts"])
    else:
        assert rms_err(y[::stride], g[f"{algo}.samples"]) <= RMS_TOL
        assert np.max(np.abs(per_s - g[f"{algo}.rms_per_second"])) < 3e-4
    mix = np.sqrt(np.mean(np.array(x[:n]).reshape(-1, fs, y.shape[1]) ** 2, axis=1))
    assert np.all(per_s <= mix * 1.001 + 1e-6)               # a soft mask in (0, 1] never adds energy
    if algo == "extended":
        periods = ctx.last_periods(256)
        assert len(periods) == 119 and np.array_equal(periods, g["extended.periods"])
    if algo == "adaptive":
        periods = ctx.last_periods(ctx.last_frame_count())
        assert np.array_equal(periods, g["adaptive.periods"])                   # all 14 064 frames
    if algo == "original":
        assert ctx.last_periods(1)[0] == int(g["original.period"])
    if algo == "simonline":
        assert np.all(y[:(p.buffer_frames - 1) * p.step_length] == 0)     # first 10 s exactly zero (repet.py:834)
    # linearity in the input scale (eps = 2^-52 only matters for exact zeros): f(2x) == 2 f(x)
    ctx.upload(np.asarray(x) * 2.0)
    ctx.execute(algo, p)
    assert rms_err(ctx.download(), 2.0 * y) < 2e-6
    ctx.close()


@pytest.mark.parametrize("algo", ["sim", "extended"])
def test_timing_series_of_back_to_back_runs(algo):
    """bench.py's timed region: K runs enqueued without a host wait, each with its own per-stage events
    (repet_ctx_timing_series_begin / _end). Same stages as the blocking call reports, same result bits."""
    x = synth(40.0, 16000, 2, 5)
    p = repet.derive_params(16000)
    ctx = repet.Context(0)
    ctx.upload(x)
    blocking = ctx.execute(algo, p, timing=True)
    want = ctx.download()
    ctx.timing_series_begin(3)
    for _ in range(4):                      # the fourth run is beyond the series: it runs, untimed
        ctx.execute_async(algo, p)
    series = ctx.timing_series_end()
    assert series["steps"] == 3
    assert [s["name"] for s in series["stages"]] == [s["name"] for s in blocking["stages"]]
    assert all(s["ms"] > 0 for s in series["stages"])
    assert [s["bytes"] for s in series["stages"]] == [s["bytes"] for s in blocking["stages"]]
    assert abs(sum(s["ms"] for s in series["stages"]) - series["total_ms"]) <= 0.02 * series["total_ms"] + 1e-3
    assert np.array_equal(ctx.download(), want)
    again = ctx.execute(algo, p, timing=True)          # the blocking form still works afterwards
    assert [s["name"] for s in again["stages"]] == [s["name"] for s in blocking["stages"]]
