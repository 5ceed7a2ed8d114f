"""Shared helpers for the parity tests (input regeneration, fixture loading, error metrics)."""
import functools
import json
import os

import numpy as np

from repet_synth import synth, synth_groove

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE_WAV = "/root/reference/audio_file.wav"


@functools.lru_cache(maxsize=8)
def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_edge_cases():
    with open(os.path.join(GOLDEN, "edge_cases.json")) as fh:
        return json.load(fh)


@functools.lru_cache(maxsize=4)
def golden_input(name):
    """Regenerate the clip of a fixture from the synth formula and check it against the stored
    strided samples (guards against libm differences between hosts)."""
    g = load_golden(name)
    if name == "cfg1_audio_file":
        # BASELINE.json configs[0]: the reference's example clip. The audio itself is NOT in this repository (SURVEY 0:
        # a commercial song excerpt without a licence); only statistics of the reference's outputs on it are. So this
        # case runs where the reference tree is present (the build container) and is skipped elsewhere (the GPU box).
        import pytest
        if not os.path.exists(REFERENCE_WAV):
            pytest.skip("needs %s (build container only; the clip is not redistributed)" % REFERENCE_WAV)
        import scipy.io.wavfile
        _, pcm = scipy.io.wavfile.read(REFERENCE_WAV)
        x = pcm / pow(2, pcm.itemsize * 8 - 1)             # what the reference's wavread returns (repet.py:929)
    elif "family" in g and str(g["family"]) == "song":
        from repet_synth import synth_song
        x = synth_song(int(round(float(g["duration"]) * int(g["fs"]))), int(g["fs"]), int(g["channels"]), int(g["seed"]))
    else:
        make = synth_groove if "family" in g and str(g["family"]) == "groove" else synth
        x = make(float(g["duration"]), int(g["fs"]), int(g["channels"]), int(g["seed"]))
    stride = int(g["sample_stride"])
    assert np.max(np.abs(x[::stride] - g["input_samples"])) < 1e-12, "synth() drifted on this host"
    x.setflags(write=False)
    return x, int(g["fs"])


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def rms_err(a, b):
    return rms(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))


def list_difference_gaps(algo, tr, ours, prm):
    """Rows whose similar-frame list differs from the oracle's (trace `tr` of the oracle run), and for every frame that
    is in one list but not the other the distance, in the ORACLE's float64 similarity, to the decision it sat on:
    the largest other value of its +-d window (strict local-maximum test, repet.py:1318-1326), the smallest kept
    value when the top-K cut is active (repet.py:1335-1340), or the threshold. A small gap NAMES the tie.
    Returns (number of differing rows, [(row, frame, gap), ...])."""
    theirs = tr.items["similarity_indices"]
    dist = prm.sim_distance_frames
    differ, named = 0, []
    for r in range(len(theirs)):
        a, b = set(np.asarray(ours[r]).tolist()), set(np.asarray(theirs[r]).tolist())
        if a == b:
            continue
        differ += 1
        if algo == "sim":                      # the scanned vector and the position of a frame inside it
            vec = tr.items["similarity_matrix"][:, r]
            pos = {f: f for f in a | b}
        else:
            in_col, vec = tr.items["similarity_vectors"][r]
            where = {int(f): c for c, f in enumerate(in_col.tolist())}
            pos = {f: where[f] for f in a | b}
        kept = [vec[pos[f]] for f in b]
        cut = min(kept) if len(b) >= prm.sim_number and kept else None
        for f in a ^ b:
            i = pos[f]
            lo, hi = max(i - dist, 0), min(i + dist + 1, len(vec))
            window = np.concatenate((vec[lo:i], vec[i + 1:hi]))
            gap_window = abs(np.nanmax(window) - vec[i]) if len(window) > 0 else np.inf
            gap_cut = abs(vec[i] - cut) if cut is not None else np.inf
            gap_thr = abs(vec[i] - prm.sim_threshold)
            named.append((r, int(f), float(min(gap_window, gap_cut, gap_thr))))
    return differ, named


class ParityOutcome(int):
    """What assert_parity_modulo_near_ties found: the int value is the number of frames whose lists differed;
    `.branch` says which bar was met -- "strict" (plain RMS <= rms_tol, the north-star bar) or "ties" (the
    diagnostic fallback) -- and `.rms` is the plain RMS error either way."""
    def __new__(cls, differ, branch, rms):
        o = super().__new__(cls, differ)
        o.branch, o.rms = branch, rms
        return o


def assert_parity_modulo_near_ties(algo, x, fs, params_kwargs=None, rms_tol=1e-4, tie_tol=5e-6, strict_tol=2e-5,
                                   require_strict=True):
    """Parity check for the similarity variants (sim / simonline).

    The bar is the north star's: plain RMS error of background_signal <= `rms_tol` (1e-4). With the float64
    near-tie refinement in the peak kernel (peaks.hip) the engine's similar-frame lists are the reference's, so
    that bar is REQUIRED by default (`require_strict=True`): when it is missed the test fails, and the tie
    analysis below only runs to say why (which frames differ, and whether each is a float64 near-tie).

    `require_strict=False` keeps the old two-way policy for inputs where a tie can be NAMED (a similarity
    threshold that sits exactly on float64 rounding, similarity_distance = 0 with 300 candidates):
      1. plain RMS within `rms_tol` -> branch "strict";
      2. otherwise every frame that is in one list but not the other must be a near-tie in the ORACLE's
         float64 similarity (within `tie_tol` of the maximum of its +-d window, of the top-K cut, or of the
         threshold), and with the oracle forced to use the engine's lists the outputs must agree to
         `strict_tol` -> branch "ties".
    Returns a ParityOutcome (int = number of frames whose lists differed, .branch, .rms)."""
    import repet
    from oracle import repet_oracle as orc
    p = orc.Params(**(params_kwargs or {}))
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](np.array(x), fs, p, tr)
    got = getattr(repet, algo)(x, fs)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    err = rms_err(got[ok], want[ok])
    if err <= rms_tol:
        return ParityOutcome(0, "strict", err)
    prm = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, prm)
    t = ctx.last_frame_count()
    rows = t if algo == "sim" else max(t - prm.buffer_frames + 1, 0)
    idx, cnt = ctx.last_sim_indices(rows, prm.sim_number)
    stats = ctx.last_refine_stats()
    ctx.close()
    ours = [idx[r, :cnt[r]].astype(int) for r in range(rows)]
    differ, named = list_difference_gaps(algo, tr, ours, prm)
    if not require_strict:
        for r, f, gap in named:
            assert gap <= tie_tol, (r, f, gap)
    if require_strict:
        raise AssertionError(f"{algo}: rms {err:.3e} > {rms_tol:g}; {differ} of {rows} similar-frame lists differ from the "
                             f"float64 oracle's; (row, frame, gap to the nearest tie) of the first: {named[:8]}; refine stats {stats}")
    assert differ > 0, f"rms {err:.3e} above tolerance although every index list matches"
    forced = orc.ALGORITHMS[algo](np.array(x), fs, p, None, override_indices=ours)
    assert rms_err(got[ok], forced[ok]) <= strict_tol
    return ParityOutcome(differ, "ties", err)


def periodic_clip(fs, period_hops, seconds, channels, seed=3, jitter=0.0):
    """An EXACTLY periodic clip: one period of `period_hops` STFT hops of the synth() mixture, tiled bit for bit
    (SURVEY 7 hard part 1: the period is a multiple of the hop, there is no noise floor between periods, so frames
    one period apart have identical spectra and the similarity matrix holds exact ties). `jitter` > 0 adds white
    noise of that amplitude over the whole clip: the ties become differences far below fp32 resolution."""
    from oracle import repet_oracle as orc
    hop = orc.window_length_for(fs) // 2
    period = period_hops * hop
    base = synth(period / fs, fs, channels, seed)
    assert len(base) == period
    n = int(round(seconds * fs))
    x = np.tile(base, (-(-n // period), 1))[:n].copy()
    if jitter > 0:
        x += jitter * np.random.RandomState(seed + 1).standard_normal(x.shape)
    return x
