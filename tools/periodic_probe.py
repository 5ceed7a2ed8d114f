#!/usr/bin/env python3
"""Exactly periodic clips (period = k hops, no noise floor): what the engine does against the float64 oracle.

SURVEY 7 hard part 1 names these inputs as adversarial: frames one period apart have IDENTICAL spectra, so the
similarity matrix holds exact ties. Prints, per case: list-length statistics of both sides, rows whose lists differ
(as sets, and modulo "same frame class" = equal index mod k), NaN counts, RMS error where both are finite, and the
near-tie refinement counters (flat_rows = rows with more near-ties than the refinement takes on).
    python tools/periodic_probe.py            # on the GPU box
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "repet-python_amd"), ROOT, os.path.join(ROOT, "tests")]

import repet  # noqa: E402
from helpers import periodic_clip  # noqa: E402
from oracle import repet_oracle as orc  # noqa: E402


def probe(algo, fs, k, seconds, jitter, channels=2):
    x = periodic_clip(fs, k, seconds, channels, seed=3, jitter=jitter)
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, None, tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, p)
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    stats = ctx.last_refine_stats()
    ctx.close()
    ours = [idx[r, :cnt[r]] for r in range(len(theirs))]
    tc = np.array([len(t) for t in theirs])
    differ = sum(set(a.tolist()) != set(np.asarray(b).tolist()) for a, b in zip(ours, theirs))
    differ_mod = sum(sorted((a % k).tolist()) != sorted((np.asarray(b) % k).tolist()) for a, b in zip(ours, theirs))
    both = ~np.isnan(got) & ~np.isnan(want)
    err = float(np.sqrt(np.mean((got[both] - want[both]) ** 2))) if both.any() else float("nan")
    print(f"{algo:9s} fs {fs} k {k:3d} d {p.sim_distance_frames} jitter {jitter:g}: rows {len(theirs)}, oracle cnt {tc.min()}/{tc.mean():.2f}/{tc.max()}, "
          f"engine cnt {cnt.min()}/{cnt.mean():.2f}/{cnt.max()}, cnt differ {int(np.sum(cnt != tc))}, lists differ {differ} (mod k: {differ_mod}), "
          f"nan oracle {int(np.isnan(want).sum())} engine {int(np.isnan(got).sum())} engine-only {int((np.isnan(got) & ~np.isnan(want)).sum())} "
          f"oracle-only {int((~np.isnan(got) & np.isnan(want)).sum())}, rms(both finite) {err:.3e}, {stats}", flush=True)


if __name__ == "__main__":
    for fs, ks in ((8000, (40, 12)), (44100, (64, 20))):
        for k in ks:
            for jitter in (0.0, 1e-7):
                for algo in ("sim", "simonline"):
                    probe(algo, fs, k, 24.0 if fs == 8000 else 30.0, jitter)
