"""Randomised end-to-end sweeps (tools/fuzz_parity.py) with fixed seeds: random sampling rates, channel counts,
durations and module parameters for all five variants against the float64 oracle, degenerate corners (error
behaviour included), and the streaming handle against the offline call for random chunkings."""
import importlib.util
import os

import numpy as np
import pytest

import repet

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.fixture(autouse=True)
def restore_module_parameters():
    yield
    for name, value in fuzz.DEFAULTS.items():
        setattr(repet, name, value)


@pytest.mark.parametrize("generator,n_cases,seed", [("random_case", 24, 11), ("edge_case", 60, 12)])
def test_random_parameters_against_the_oracle(generator, n_cases, seed):
    rs = np.random.RandomState(seed)
    records = [fuzz.run_case(k, *getattr(fuzz, generator)(rs)) for k in range(n_cases)]
    bad = [r for r in records if not r["ok"]]
    assert not bad, bad[:3]
    # the sweep must exercise results, not only error paths
    assert sum("rms" in r for r in records) >= n_cases // 3


def test_streaming_matches_offline_for_random_chunkings():
    rs = np.random.RandomState(13)
    records = [fuzz.stream_case(k, rs) for k in range(12)]
    bad = [r for r in records if not r["ok"]]
    assert not bad, bad[:3]
