"""Pin the CPU oracle (oracle/repet_oracle.py) against golden vectors produced by the unmodified
reference (tests/golden/make_golden.py). CPU only; float64; tolerance 1e-10 max-abs on
background_signal and exact equality on every integer intermediate."""
import numpy as np
import pytest

from helpers import golden_input, load_golden
from oracle import repet_oracle as orc

# the synth family, and the second family (synth_groove: drifting tempo, transients, level steps, a silent bar, a changing bar)
FAST_CASES = ["small_mono", "small_stereo", "mid_stereo", "groove_small", "groove_mid"]
ALGOS = ["original", "extended", "adaptive", "sim", "simonline"]
TOL = 1e-10


def _run(case, algo):
    x, fs = golden_input(case)
    tr = orc.Trace()
    y = orc.ALGORITHMS[algo](np.array(x), fs, orc.Params(), tr)
    return y, tr.items, load_golden(case)


@pytest.mark.parametrize("case", FAST_CASES)
@pytest.mark.parametrize("algo", ALGOS)
def test_background_matches_reference(case, algo):
    y, _, g = _run(case, algo)
    stride = int(g["sample_stride"])
    assert y.dtype == np.float64 and y.shape[1] == int(g["channels"])
    want = g[f"{algo}.samples"]
    assert np.array_equal(np.isnan(y[::stride]), np.isnan(want))          # (sim on a clip with a silent bar: NaN frames)
    ok = ~np.isnan(want)
    assert np.max(np.abs(y[::stride][ok] - want[ok])) <= TOL
    assert np.allclose(np.sum(y, axis=0), g[f"{algo}.sum"], rtol=0, atol=1e-7, equal_nan=True)


@pytest.mark.parametrize("case", FAST_CASES + ["g44k_stereo", "groove_44k"])
def test_original_intermediates(case):
    _, tr, g = _run(case, "original")
    assert tr["repeating_period"] == int(g["original.period"])
    assert np.max(np.abs(tr["beat_spectrum"] - g["original.beat_spectrum"])) <= 1e-9 * np.max(g["original.beat_spectrum"])
    fstride = int(g["frame_stride"])
    rows = tr["mask_c0"][[0, 1, 5, 6, -1]][:, ::fstride]
    assert np.max(np.abs(rows - g["original.mask_c0_rows"])) <= 1e-9


@pytest.mark.parametrize("case", FAST_CASES + ["g44k_stereo", "groove_44k"])
def test_sim_intermediates(case):
    _, tr, g = _run(case, "sim")
    s = tr["similarity_matrix"]
    t = s.shape[0]
    cols, want = s[:, [0, t // 2, t - 1]], g["sim.similarity_columns"]
    assert np.array_equal(np.isnan(cols), np.isnan(want))
    assert np.max(np.abs(cols[~np.isnan(want)] - want[~np.isnan(want)])) <= 1e-12
    counts = np.array([len(ix) for ix in tr["similarity_indices"]])
    assert np.array_equal(counts, g["sim.counts"])
    for row, frame in zip(g["sim.indices"], g["sim.index_frames"]):
        assert np.array_equal(tr["similarity_indices"][frame], row[row >= 0])


@pytest.mark.parametrize("case", FAST_CASES)
def test_adaptive_and_extended_periods(case):
    _, tr, g = _run(case, "adaptive")
    assert np.array_equal(tr["repeating_periods"], g["adaptive.periods"])
    _, tr, g = _run(case, "extended")
    if "segment_periods" in tr:
        assert np.array_equal(tr["segment_periods"], g["extended.periods"])
    else:
        assert len(g["extended.periods"]) == 1


@pytest.mark.parametrize("case", FAST_CASES)
def test_simonline_indices(case):
    _, tr, g = _run(case, "simonline")
    b = tr["buffer_frames"]
    counts = np.array([len(ix) for ix in tr["similarity_indices"]])
    assert np.array_equal(counts, g["simonline.counts"])
    fstride = int(g["frame_stride"])
    for k, row in enumerate(g["simonline.buffer_indices"]):
        j = b - 1 + k * fstride
        cols = row[row >= 0]
        assert np.array_equal(tr["similarity_indices"][k * fstride], j - np.mod(j - cols, b))


def test_groove_family_has_what_it_says():
    """The second clip family carries the structure it is meant to test: a bar of exact zeros (sim and simonline divide
    0 by 0 in their cosine similarity there, repet.py:1220 / :1240: NaN frames; original / extended / adaptive stay finite) and
    adaptive periods that are NOT constant."""
    g = load_golden("groove_mid")
    x, fs = golden_input("groove_mid")
    assert int((np.asarray(x) == 0).all(axis=1).sum()) > fs                       # more than a second of digital silence
    assert np.isnan(g["sim.samples"]).any() and not np.isnan(g["original.samples"]).any()
    assert not np.isnan(g["adaptive.samples"]).any() and not np.isnan(g["extended.samples"]).any()
    assert np.isnan(g["simonline.samples"]).any()                  # (the silent bar lies behind the 10-s warm-up of this clip)
    assert len(np.unique(g["adaptive.periods"])) > 2


def test_extended_has_two_segments_at_8k():
    g = load_golden("small_stereo")
    assert len(g["extended.periods"]) == 2
    x, fs = golden_input("small_stereo")
    segs, overlap = orc.extended_plan(len(x), fs, orc.Params())
    assert segs == [(0, 80000), (40000, 88000)] and overlap == 40000


def test_half_to_even_buffer_length():
    # B = round(10*8000/256) = round(312.5) = 312 (banker's rounding, repet.py:787)
    assert round((10 * 8000) / 256) == 312


def test_localmaxima_matches_definition():
    rs = np.random.RandomState(0)
    for n, d, k in [(50, 3, 5), (200, 7, 100), (31, 40, 4), (1, 2, 3)]:
        v = rs.rand(n)
        v[rs.randint(0, n, size=max(1, n // 10))] = np.nan
        want = [i for i in range(n)
                if v[i] >= 0.2 and all(v[i] > v[max(i - d, 0):i]) and all(v[i] > v[i + 1:min(i + d + 1, n)])]
        want = np.array(want, dtype=int)
        order = np.argsort(v[want])[::-1][:k]
        vals, idx = orc.localmaxima(v, 0.2, d, k)
        assert np.array_equal(idx, want[order])


# ---- BASELINE.json configs[0]: the reference's own example clip (real music, 23 s, 44.1 kHz stereo) ----
@pytest.mark.parametrize("case", ["cfg1_audio_file", "cfg1_surrogate"])
@pytest.mark.parametrize("algo", ALGOS)
def test_reference_example_clip(algo, case):
    """The reference's own clip (where the reference tree is) and its redistributable surrogate of the same shape
    (repet_synth.synth_song; fixture from make_golden.py --cases cfg1s)."""
    y, tr, g = _run(case, algo)
    stride = int(g["sample_stride"])
    assert y.shape == (1014301, 2)
    assert np.max(np.abs(y[::stride] - g[f"{algo}.samples"])) <= TOL
    if algo == "original":
        assert tr["repeating_period"] == int(g["original.period"]) == (286 if case == "cfg1_audio_file" else 287)   # arg-max lag 285
    if algo == "extended":
        assert np.array_equal(tr["segment_periods"], g["extended.periods"])
    if algo == "adaptive":
        assert np.array_equal(tr["repeating_periods"], g["adaptive.periods"])
    if algo == "sim":
        assert np.array_equal(np.array([len(ix) for ix in tr["similarity_indices"]]), g["sim.counts"])
        for row, frame in zip(g["sim.indices"], g["sim.index_frames"]):
            assert np.array_equal(tr["similarity_indices"][frame], row[row >= 0])
    if algo == "simonline":
        assert np.array_equal(np.array([len(ix) for ix in tr["similarity_indices"]]), g["simonline.counts"])


# ---- BASELINE.json config sizes: oracle vs the reference's strided samples and integer intermediates ----
CONFIG_CASES = [("cfg5_simonline", "simonline"), ("cfg4_adaptive", "adaptive"), ("cfg3_extended", "extended")]
if __import__("os").environ.get("REPET_FULL_GOLDEN") == "1":      # 180-s sim: ~1.5 CPU-minutes for the oracle
    CONFIG_CASES += [("cfg2_sim", "sim"), ("cfg2_sim", "original"), ("cfg2_groove", "sim"), ("cfg2_groove", "original")]


@pytest.mark.parametrize("case,algo", CONFIG_CASES)
def test_config_size_goldens(case, algo):
    y, tr, g = _run(case, algo)
    stride = int(g["sample_stride"])
    want = g[f"{algo}.samples"]
    assert np.array_equal(np.isnan(y[::stride]), np.isnan(want))
    assert np.max(np.abs(y[::stride][~np.isnan(want)] - want[~np.isnan(want)])) <= TOL
    fs = int(g["fs"])
    n = (len(y) // fs) * fs
    per_s = np.sqrt(np.mean(y[:n].reshape(-1, fs, y.shape[1]) ** 2, axis=1))
    want_s = g[f"{algo}.rms_per_second"]
    assert np.array_equal(np.isnan(per_s), np.isnan(want_s))
    assert np.max(np.abs(per_s[~np.isnan(want_s)] - want_s[~np.isnan(want_s)])) <= 1e-9
    if algo == "adaptive":
        assert np.array_equal(tr["repeating_periods"], g["adaptive.periods"])
    if algo == "extended":
        assert np.array_equal(tr["segment_periods"], g["extended.periods"]) and len(g["extended.periods"]) == 119
    if algo == "simonline":
        assert np.array_equal(np.array([len(ix) for ix in tr["similarity_indices"]]), g["simonline.counts"])
    if algo == "sim":
        assert np.array_equal(np.array([len(ix) for ix in tr["similarity_indices"]]), g["sim.counts"])
