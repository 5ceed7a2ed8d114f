"""NaN runs of the engine against the oracle on a clip with samples that are not finite (strict_reference): tools/strict_probe.py algo fs seconds channels dtype"""
import sys
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
import repet
from repet_synth import synth
from oracle import repet_oracle as orc

algo, fs, seconds, channels, dtype = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), np.dtype(sys.argv[5])
repet.strict_reference = True
x = synth(seconds, fs, channels, 11).astype(dtype)
n = len(x)
x[n // 3, 0] = np.nan
x[n // 2:n // 2 + 5000, channels - 1] = np.nan
x[(2 * n) // 3, 0] = np.inf
x[(2 * n) // 3 + 40000, channels - 1] = -np.inf
tr = orc.Trace()
with np.errstate(all="ignore"):
    want = orc.ALGORITHMS[algo](x.astype(np.float64), fs, None, tr)
got = getattr(repet, algo)(x, fs)


def runs(mask):
    rows = np.flatnonzero(mask.any(axis=1))
    out = []
    if len(rows):
        s = p = rows[0]
        for r in rows[1:]:
            if r != p + 1:
                out.append((int(s), int(p)))
                s = r
            p = r
        out.append((int(s), int(p)))
    return out


h = repet.derive_params(fs).step_length
print("oracle NaN runs (samples):", runs(np.isnan(want)), "in hops:", [(a / h, b / h) for a, b in runs(np.isnan(want))])
print("engine NaN runs (samples):", runs(np.isnan(got)), "in hops:", [(a / h, b / h) for a, b in runs(np.isnan(got))])
print("engine inf:", runs(np.isinf(got)))
ok = ~np.isnan(want) & ~np.isnan(got)
print("rms on common finite samples", float(np.sqrt(np.mean((got[ok] - want[ok]) ** 2))))
