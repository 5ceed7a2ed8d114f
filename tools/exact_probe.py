"""Second level of the peak picking (peaks_exact.hip) against the float64 oracle: rows whose similar-frame list differs,
the first pass's counters and the second level's, per case. Run on the GPU box:
    python tools/exact_probe.py [lists|periodic|all]            (REPET_PEAK_EXACT=0 for the first pass alone)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import repet  # noqa: E402
from repet_synth import synth  # noqa: E402
from oracle import repet_oracle as orc  # noqa: E402
from helpers import list_difference_gaps, periodic_clip, rms_err  # noqa: E402


def run(algo, x, fs, number=100, label=""):
    repet.similarity_number = number
    tr = orc.Trace()
    want = orc.ALGORITHMS[algo](x, fs, orc.Params(similarity_number=number), tr)
    theirs = tr.items["similarity_indices"]
    p = repet.derive_params(fs)
    ctx = repet.Context(0)
    ctx.upload(x)
    t0 = time.perf_counter()
    ctx.execute(algo, p)
    dt = time.perf_counter() - t0
    got = ctx.download()
    idx, cnt = ctx.last_sim_indices(len(theirs), p.sim_number)
    stats, exact = ctx.last_refine_stats(), ctx.last_exact_stats()
    ctx.close()
    ours = [idx[r, :cnt[r]] for r in range(len(theirs))]
    differ, named = list_difference_gaps(algo, tr, ours, p)
    nan_equal = bool(np.array_equal(np.isnan(got), np.isnan(want)))
    ok = ~np.isnan(want) & ~np.isnan(got)
    out = {"case": label, "rows": len(theirs), "rows_differing": differ, "gaps": sorted(g for _, _, g in named)[:6],
           "nan_positions_equal": nan_equal, "nan_ours": int(np.isnan(got).sum()), "nan_theirs": int(np.isnan(want).sum()),
           "rms": rms_err(got[ok], want[ok]) if ok.any() else None, "ms": round(dt * 1e3, 3), **stats, **exact}
    print(json.dumps(out), flush=True)
    return out


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("lists", "all"):
        for algo, seconds, fs, ch, seed, number in [
                ("sim", 60, 22050, 2, 1, 100), ("sim", 20, 96000, 1, 3, 100), ("sim", 90, 16000, 2, 4, 100),
                ("simonline", 45, 16000, 2, 5, 100), ("simonline", 30, 44100, 1, 6, 100),
                ("sim", 60, 22050, 2, 7, 12), ("sim", 50, 16000, 1, 8, 5), ("simonline", 40, 16000, 2, 9, 4),
                ("sim", 40, 44100, 2, 0, 100)]:
            run(algo, synth(seconds, fs, ch, seed), fs, number, f"{algo} {seconds}s {fs}Hz {ch}ch seed{seed} K{number}")
    if what in ("periodic", "all"):
        for algo in ("sim", "simonline"):
            run(algo, periodic_clip(8000, 12, 24.0, 2, jitter=1e-7), 8000, 100, f"{algo} periodic k=12 jitter 1e-7")
            run(algo, periodic_clip(8000, 40, 24.0, 2, jitter=1e-7), 8000, 100, f"{algo} periodic k=40 jitter 1e-7")
            run(algo, periodic_clip(8000, 12, 24.0, 2), 8000, 100, f"{algo} periodic k=12 exact")
            run(algo, periodic_clip(44100, 20, 30.0, 2, jitter=1e-7), 44100, 100, f"{algo} periodic 44.1k k=20 jitter 1e-7")


if __name__ == "__main__":
    main()
