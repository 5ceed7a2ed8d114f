"""Randomised stage-level checks on the MI355X (hypothesis): the index-heavy kernels against the oracle on
ragged / degenerate inputs -- NaNs, plateaus, exact ties, empty and over-long lists, odd lengths."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import repet
from oracle import repet_oracle as orc

pytestmark = pytest.mark.gpu
SETTINGS = dict(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))


@settings(**SETTINGS)
@given(n=st.integers(1, 1500), d=st.integers(0, 120), k=st.integers(1, 140), seed=st.integers(0, 2**31 - 1),
       kind=st.sampled_from(["uniform", "smooth", "quantised", "nan", "plateau"]), thr=st.sampled_from([0.0, 0.3, -1.0]))
def test_local_maxima_random(n, d, k, seed, kind, thr):
    rs = np.random.RandomState(seed)
    v = rs.rand(n).astype(np.float32)
    if kind == "smooth":
        v = np.convolve(rs.rand(n + 20), np.hanning(21) / 10.0, mode="valid")[:n].astype(np.float32)
    elif kind == "quantised":
        v = (np.round(v * 8) / 8).astype(np.float32)            # many exact ties
    elif kind == "nan":
        v[rs.randint(0, n, size=max(1, n // 15))] = np.nan
    elif kind == "plateau":
        v[:] = 0.5
    vals, idx = repet._localmaxima(v, thr, d, k)
    wv, wi = orc.localmaxima(v.astype(np.float64), thr, d, k)
    assert len(idx) == len(wi)
    all_v, _ = orc.localmaxima(v.astype(np.float64), thr, d, n)   # every peak, to see ties at the top-k cut
    if len(np.unique(all_v)) == len(all_v):
        assert np.array_equal(idx, wi)
    else:             # exact ties: numpy's (unstable) argsort decides the reference order; same multiset of values
        assert np.array_equal(np.sort(vals), np.sort(wv))
        assert len(set(idx.tolist())) == len(idx)
    # every reported index really is a strict local maximum above the threshold
    for i in idx:
        assert v[i] >= thr
        lo, hi = max(i - d, 0), min(i + d + 1, n)
        assert all(v[i] > v[lo:i]) and all(v[i] > v[i + 1:hi])


@settings(**SETTINGS)
@given(t=st.integers(1, 90), f=st.integers(1, 200), seed=st.integers(0, 2**31 - 1),
       max_len=st.sampled_from([1, 2, 3, 5, 9, 17, 33, 70, 100, 129, 200]))
def test_sim_mask_random_lists(t, f, seed, max_len):
    rs = np.random.RandomState(seed)
    v = (rs.rand(f, t) ** 3).astype(np.float32).astype(np.float64)
    v[rs.rand(f, t) < 0.02] = 0.0                               # exact zeros: mask = eps/eps = 1
    lists = [rs.randint(0, t, size=rs.randint(0, max_len + 1)) for _ in range(t)]   # ragged, may repeat, may be empty
    got = repet._simmask(v, lists)
    want = orc.simmask(v, lists)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.max(np.abs(got[ok] - want[ok]), initial=0.0) < 1e-6
    assert np.all((got[ok] > 0) & (got[ok] <= 1.0))


@settings(**SETTINGS)
@given(t=st.integers(1, 120), f=st.integers(1, 150), period=st.integers(1, 130), seed=st.integers(0, 2**31 - 1))
def test_period_mask_random(t, f, period, seed):
    rs = np.random.RandomState(seed)
    v = rs.rand(f, t).astype(np.float32).astype(np.float64)
    got = repet._mask(v, period)
    want = orc.mask(v, period)
    assert np.max(np.abs(got - want)) < 1e-6


@settings(**SETTINGS)
@given(t=st.integers(1, 100), f=st.integers(1, 150), order=st.integers(1, 9), seed=st.integers(0, 2**31 - 1))
def test_adaptive_mask_random(t, f, order, seed):
    rs = np.random.RandomState(seed)
    v = rs.rand(f, t).astype(np.float32).astype(np.float64)
    per = rs.randint(1, max(2, t), size=t)
    got = repet._adaptivemask(v, per, order)
    want = orc.adaptivemask(v, per, order)
    assert np.max(np.abs(got - want)) < 1e-6


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 9000), logw=st.integers(6, 11), seed=st.integers(0, 2**31 - 1))
def test_stft_istft_random_lengths(n, logw, seed):
    rs = np.random.RandomState(seed)
    w = 1 << logw
    x = rs.randn(n)
    window = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(w) / w)
    spec = repet._stft(x, window, w // 2)
    want = orc.stft(x.astype(np.float32).astype(np.float64), window.astype(np.float32).astype(np.float64), w // 2)
    assert spec.shape == want.shape
    assert np.max(np.abs(spec - want)) < 3e-6 * max(1.0, np.max(np.abs(want)))
    y = repet._istft(spec, window, w // 2)
    assert np.max(np.abs(y[:n] - x)) < 5e-6 * max(1.0, np.max(np.abs(x)))      # COLA reconstruction


@settings(max_examples=15, deadline=None, suppress_health_check=list(HealthCheck))
@given(t=st.integers(2, 300), f=st.integers(2, 300), seed=st.integers(0, 2**31 - 1))
def test_selfsimilarity_random_shapes(t, f, seed):
    rs = np.random.RandomState(seed)
    v = rs.rand(f, t).astype(np.float32).astype(np.float64)
    got = repet._selfsimilaritymatrix(v)
    want = orc.selfsimilaritymatrix(v)
    assert np.max(np.abs(got - want)) < 3e-6
    assert np.array_equal(got, got.T)
