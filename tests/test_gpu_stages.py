"""Stage-level parity on the MI355X: every HIP stage export (through the C ABI) against the float64
oracle on the same seeded inputs. Integer results (periods, peak indices) must match exactly when both
sides see the same float32 values; floating-point stages are held to fp32-rounding tolerances."""
import numpy as np
import pytest

import repet
from helpers import golden_input
from oracle import repet_oracle as orc

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


@pytest.fixture(scope="module")
def clip():
    x, fs = golden_input("small_stereo")
    return np.array(x), fs


@pytest.mark.parametrize("fs,seconds", [(8000, 3.0), (16000, 2.0), (44100, 1.5), (96000, 0.5)])
def test_stft_matches_oracle(fs, seconds):
    from repet_synth import synth
    x = synth(seconds, fs, 1, 11)[:, 0]
    w, window, h = orc.stft_geometry(fs)
    got = repet._stft(x, window, h)
    want = orc.stft(x.astype(np.float32).astype(np.float64), window, h)
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-6


def test_stft_is_hermitian_and_handles_odd_lengths():
    from repet_synth import synth
    for n in (1, 255, 256, 257, 1000, 4097):
        x = synth(1.0, 8000, 1, 3)[:n, 0]
        w, window, h = orc.stft_geometry(8000)
        got = repet._stft(x, window, h)
        want = orc.stft(x.astype(np.float32).astype(np.float64), window, h)
        assert got.shape == want.shape == (w, int(np.ceil(n / h)) + 1)
        assert np.max(np.abs(got - want)) < 1e-5
        assert np.allclose(got[1:w // 2], np.conj(got[:w // 2:-1]))


@pytest.mark.parametrize("fs", [8000, 44100])
def test_istft_roundtrip_and_oracle(fs):
    from repet_synth import synth
    x = synth(2.0, fs, 1, 5)[:, 0]
    w, window, h = orc.stft_geometry(fs)
    spec = orc.stft(x, window, h)
    got = repet._istft(spec, window, h)
    want = orc.istft(spec, window, h)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 2e-6
    assert np.max(np.abs(got[:len(x)] - x)) < 2e-6          # COLA: exact reconstruction


def test_selfsimilarity(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x, window, h)
    v = np.mean(mag, axis=2).astype(np.float32).astype(np.float64)
    got = repet._selfsimilaritymatrix(v)
    want = orc.selfsimilaritymatrix(v)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 2e-6
    assert np.array_equal(got, got.T)                         # mirrored tiles are bit-identical


def test_selfsimilarity_silent_frame_gives_nan_row_and_column():
    rs = np.random.RandomState(0)
    v = rs.rand(40, 150)
    v[:, 17] = 0.0
    got = repet._selfsimilaritymatrix(v)
    assert np.all(np.isnan(got[17])) and np.all(np.isnan(got[:, 17]))
    keep = np.delete(np.arange(150), 17)
    assert np.all(np.isfinite(got[np.ix_(keep, keep)]))


def test_beat_spectrum_and_period(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x, window, h)
    p = np.power(np.mean(mag, axis=2), 2).astype(np.float32).astype(np.float64)
    got = repet._beatspectrum(p)
    want = orc.beatspectrum(p)
    assert _rel(got, want) < 2e-5
    pr = orc.period_range_frames(orc.Params(), fs, h)
    assert repet._periods(got, pr) == orc.periods(want, pr)
    # period kernel alone, exact on identical float32 input
    b32 = want.astype(np.float32)
    assert repet._periods(b32, pr) == orc.periods(b32.astype(np.float64), pr)


def test_beat_spectrogram_with_hole_quirk(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x[:6 * fs], window, h)
    p = np.power(np.mean(mag, axis=2), 2).astype(np.float32).astype(np.float64)
    seg_len, seg_step = 90, 40
    got = repet._beatspectrogram(p, seg_len, seg_step)
    want = orc.beatspectrogram(p, seg_len, seg_step)
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-5
    assert np.all(got[:, seg_step - 1] == 0)                  # repet.py:1202-1204 leaves this column zero
    pr = [8, 80]
    assert np.array_equal(repet._periods(want.astype(np.float32), pr),
                          orc.periods(want.astype(np.float32).astype(np.float64), pr))


def test_local_maxima_exact():
    rs = np.random.RandomState(1)
    for n, d, k in [(700, 31, 100), (700, 31, 5), (97, 200, 10), (64, 1, 100), (5, 2, 3), (300, 0, 20)]:
        m = rs.rand(6, n).astype(np.float32)
        m[2, rs.randint(0, n, 5)] = np.nan
        m[3] = 0.25                                          # plateau: no strict maximum anywhere
        m[4, ::7] = m[4, 3]                                  # exact ties
        for r in range(6):
            vals, idx = repet._localmaxima(m[r], 0.1, d, k)
            wv, wi = orc.localmaxima(m[r].astype(np.float64), 0.1, d, k)
            if len(np.unique(wv)) == len(wv):
                assert np.array_equal(idx, wi), (n, d, k, r)
            else:                                            # exact ties: any tied element may sit at the cut
                assert np.array_equal(np.sort(vals), np.sort(wv))
                assert np.array_equal(m[r][idx].astype(np.float64), vals) and len(set(idx)) == len(idx)


def test_local_maxima_from_segment_records():
    """Rows whose window reaches at least 31 elements either way are picked from SEGMENT RECORDS (largest, second largest
    and position of every aligned 32-element run; peaks_wave.hip) instead of a sweep over the row: smooth rows (the maxima
    of the neighbouring segments sit at their edges, so the cut segments have to be read), quantised rows (exact ties inside
    and across segments), NaNs, thresholds in the middle of the values, lengths that end inside a segment."""
    rs = np.random.RandomState(11)
    for n, d, k, thr in [(1000, 31, 100, 0.0), (2081, 43, 100, 0.3), (777, 47, 7, 0.0), (4097, 63, 100, 0.55), (96, 40, 5, 0.0),
                         (33, 31, 3, 0.0), (2500, 32, 100, 0.0)]:
        t = np.arange(n)
        rows = [rs.rand(n),
                0.5 + 0.3 * np.sin(2 * np.pi * t / 73.0) + 0.1 * np.sin(2 * np.pi * t / 517.0) + 1e-3 * rs.rand(n),   # smooth
                np.round(rs.rand(n) * 16) / 16,                                                                       # many exact ties
                0.5 + 0.4 * np.sin(2 * np.pi * t / (2.0 * d + 1.0)),                                                  # peaks one window apart
                np.where(rs.rand(n) < 0.01, np.nan, rs.rand(n)),
                np.full(n, 0.25)]
        rows[4][n // 2] = np.nan
        for r, row in enumerate(rows):
            m = row.astype(np.float32)
            vals, idx = repet._localmaxima(m, thr, d, k)
            wv, wi = orc.localmaxima(m.astype(np.float64), thr, d, k)
            assert len(idx) == len(wi), (n, d, k, r)
            if len(np.unique(wv)) == len(wv):
                assert np.array_equal(idx, wi), (n, d, k, r)
            else:                                            # exact ties among the survivors: any tied element may sit at the cut
                assert np.array_equal(np.sort(vals), np.sort(wv)), (n, d, k, r)
                assert np.array_equal(m[idx].astype(np.float64), vals) and len(set(idx)) == len(idx)
                if len(wi) < k:
                    assert set(idx.tolist()) == set(wi.tolist()), (n, d, k, r)


@pytest.mark.parametrize("t", [300, 2300])
def test_segment_records_of_the_similarity_matrix(t):
    """The records the peak picking of `sim` takes its candidates from, against NumPy on the matrix they belong to: written by the
    256 x 256 Gram kernel's epilogue (2 300 frames; the diagonal patches and the matrix's ragged edge take their own path) or by
    a pass over the matrix (300 frames). A silent frame gives a NaN row and column: NaN counts as +inf, so it is every
    segment's maximum wherever it lies, and its own row is all +inf."""
    rs = np.random.RandomState(17)
    v = np.abs(rs.standard_normal((257, t)))
    v[:, t // 3] = 0.0
    v[:, 40:44] = v[:, 40:41]                                    # four identical frames: exact ties inside segments
    s, top, second, at = repet._selfsimilarity_records(v)
    assert np.array_equal(s, s.T, equal_nan=True)
    n_seg = -(-t // 32)
    padded = np.full((t, n_seg * 32), -np.inf, dtype=np.float32)
    padded[:, :t] = np.where(np.isnan(s), np.inf, s)
    runs = padded.reshape(t, n_seg, 32)
    want_at = np.argmax(runs, axis=2)                             # (the lowest index among equals)
    want_top = np.take_along_axis(runs, want_at[:, :, None], axis=2)[:, :, 0]
    others = runs.copy()
    np.put_along_axis(others, want_at[:, :, None], -np.inf, axis=2)
    want_second = others.max(axis=2)
    assert np.array_equal(top, want_top)
    assert np.array_equal(second, want_second)
    assert np.array_equal(at, want_at)
    assert np.all(np.isinf(top[t // 3])) and np.all(top[:, (t // 3) // 32] == np.inf)
    assert np.any(top[:, 1] == second[:, 1])                      # the tied frames really produced ties


def test_indices_on_similarity_matrix(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x, window, h)
    s = orc.selfsimilaritymatrix(np.mean(mag, axis=2)).astype(np.float32)
    got = repet._indices(s, 0, 31, 100)
    want = orc.indices(s.astype(np.float64), 0, 31, 100)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_masks(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x, window, h)
    v = mag[:, :, 0].astype(np.float32).astype(np.float64)
    t = v.shape[1]
    for period in (32, 45, 166):
        assert np.max(np.abs(repet._mask(v, period) - orc.mask(v, period))) < 1e-6
    rs = np.random.RandomState(2)
    per = rs.randint(32, 160, size=t)
    for order in (1, 2, 3, 4, 5, 8):
        assert np.max(np.abs(repet._adaptivemask(v, per, order) - orc.adaptivemask(v, per, order))) < 1e-6
    lists = []
    for i in range(t):
        k = [0, 1, 2, 3, 7, 16, 25, 33, 49, 64, 81, 99, 100, 128, 150][i % 15]
        lists.append(rs.choice(t, size=min(k, t), replace=False))
    got = repet._simmask(v, lists)
    want = orc.simmask(v, lists)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.max(np.abs(got[ok] - want[ok])) < 1e-6


def test_similaritymatrix_and_acorr_helpers(clip):
    x, fs = clip
    w, window, h = orc.stft_geometry(fs)
    _, mag = orc.spectrogram_channels(x[:5 * fs], window, h)
    v = np.mean(mag, axis=2).astype(np.float32).astype(np.float64)            # (F, T)
    a, b = v[:, :150], v[:, 37:38]                                             # buffer vs one frame, as in simonline
    got = repet._similaritymatrix(a, b)
    want = orc.similaritymatrix(a, b)
    assert got.shape == want.shape == (150, 1)
    assert np.max(np.abs(got - want)) < 2e-6
    got = repet._similaritymatrix(v[:, :70], v[:, 20:150])
    assert np.max(np.abs(got - orc.similaritymatrix(v[:, :70], v[:, 20:150]))) < 2e-6
    p = np.power(v, 2).T[:120, :200]                                           # (rows = time, cols)
    got = repet._acorr(p)
    want = orc.acorr(p)
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-5
    assert np.allclose(np.mean(got, axis=1), repet._beatspectrum(p.T), rtol=2e-5, atol=0)


@pytest.mark.parametrize("t,f", [(1025, 128), (2048, 384), (2500, 129), (4097, 256), (7753, 257), (9000, 128), (17000, 128), (30720, 128)])
def test_rank_columns_against_numpy(t, f):
    """The rank transform behind sim's median (rank.hip): every column sorted by one workgroup (bitonic network,
    32 keys per thread; sizes 2^11 .. 2^15), code = 0x0400 + number of strictly smaller values, ties share a code."""
    rs = np.random.RandomState(t + f)
    v = np.abs(rs.standard_normal((f, t))).astype(np.float32) * np.exp(rs.uniform(-12, 3, size=(f, 1))).astype(np.float32)
    v[:, rs.randint(0, t, size=t // 7)] = v[:, rs.randint(0, t, size=t // 7)]      # ties between frames
    v[0, :] = 0.25                                                                   # a constant bin: every code equal
    v[1, : t // 2] = 0.0                                                             # zeros
    codes, ordered = repet._rank_columns(v)
    n = f // 128 * 128
    assert codes.shape == (n, t) and ordered.shape == (n, t)
    want_sorted = np.sort(v[:n], axis=1)
    assert np.array_equal(ordered, want_sorted)
    for b in range(n):
        want = np.searchsorted(want_sorted[b], v[b], side="left") + 0x0400
        assert np.array_equal(codes[b].astype(np.int64), want), b
    # what the mask kernel relies on: looking a code up in the sorted column returns the value itself
    assert np.array_equal(np.take_along_axis(ordered, codes.astype(np.int64) - 0x0400, axis=1), v[:n])


@pytest.mark.parametrize("t,f,longest", [(1400, 1025, 100), (2100, 513, 100), (1100, 2049, 100), (4200, 257, 128), (8200, 129, 37),
                                         (1025, 129, 100), (2048, 129, 64), (2049, 129, 100), (16385, 129, 101), (30720, 129, 25)])
def test_median_selection_on_rank_codes_against_numpy(t, f, longest):
    """The two rank-domain forms of sim's median (mask.hip: the packed 16-bit network; mask_bits.hip: the bit-sliced radix
    descent, lists of up to 100 and up to 128 entries, 11 to 15 code planes -- frame counts either side of a power of two --, 2 to
    32 blocks of 64 bins) against NumPy, on
    random magnitudes with ties and zeros and lists of every length from 0 to the longest, odd and even:
      * the INTEGERS the bit-sliced selection leaves -- rank of the lower median, rank of the upper one, "the lower median
        is below the frame's own value" -- are exactly what sorting the list's ranks gives;
      * both masks have the bits of the float selection (repet._simmask), NaN for an empty list (np.median, repet.py:1535)."""
    rs = np.random.RandomState(t + f + longest)
    v = np.abs(rs.standard_normal((f, t))).astype(np.float32) * np.exp(rs.uniform(-10, 2, size=(f, 1))).astype(np.float32)
    v[:, rs.randint(0, t, size=t // 9)] = v[:, rs.randint(0, t, size=t // 9)]        # ties between frames
    v[1, : t // 3] = 0.0                                                             # zeros
    v[2, :] = 0.5                                                                    # a constant bin
    lengths = rs.randint(0, longest + 1, size=t)
    lengths[:longest + 1] = np.arange(longest + 1)                                   # every length at least once
    lists = [rs.choice(t, size=k, replace=False) for k in lengths]
    for i in range(0, t, 3):                                                         # a frame is usually in its own list
        if len(lists[i]) and i not in lists[i]:
            lists[i][0] = i
    want = repet._simmask(v, lists)
    got_rank = repet._simmask_ranked(v, lists, "rank")
    got_bits, codes = repet._simmask_ranked(v, lists, "bits", want_codes=True)
    for got in (got_rank, got_bits):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(np.nan_to_num(got, nan=-1.0), np.nan_to_num(want, nan=-1.0))
    ranks = np.empty((f - 1, t), dtype=np.int64)
    for b in range(f - 1):
        ranks[b] = np.searchsorted(np.sort(v[b]), v[b], side="left")
    for i in range(0, t, 7):
        n = len(lists[i])
        if n == 0:
            continue
        r = np.sort(ranks[:, lists[i]], axis=1)
        lower, upper = r[:, (n - 1) // 2], r[:, n // 2]
        c = codes[:, i].astype(np.int64)
        assert np.array_equal(c & 0x7fff, lower), i
        assert np.array_equal(c >> 16, upper), i
        assert np.array_equal((c >> 15) & 1, (lower < ranks[:, i]).astype(np.int64)), i


def test_rank_columns_limits():
    with pytest.raises(RuntimeError):
        repet._rank_columns(np.ones((128, 1024), dtype=np.float32))      # too short: selecting on the floats is cheaper
    with pytest.raises(RuntimeError):
        repet._rank_columns(np.ones((128, 30721), dtype=np.float32))     # codes would reach +inf (0x7C00)
