"""NaN runs of the engine against the oracle on a clip with samples that are not finite (strict_reference): tools/strict_probe.py algo fs seconds channels dtype"""
import sys
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
import repet
from repet_synth import synth
from oracle import repet_oracle as orc

algo, fs, seconds, channels, dtype = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), np.dtype(sys.argv[5])
kind = sys.argv[6] if len(sys.argv) > 6 else "all"
repet.strict_reference = True
x = synth(seconds, fs, channels, 11 if kind == "all" else 3).astype(dtype)
n = len(x)
if kind == "all":
    x[n // 3, 0] = np.nan
    x[n // 2:n // 2 + 5000, channels - 1] = np.nan
    x[(2 * n) // 3, 0] = np.inf
    x[(2 * n) // 3 + 40000, channels - 1] = -np.inf
if kind in ("nan", "both"):
    x[n // 3, 0] = np.nan
if kind in ("inf", "both"):
    x[(2 * n) // 3 + 777, 1] = np.inf
    x[(5 * n) // 6, 0] = -np.inf
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
for ch in range(channels):
    print("channel", ch)
    print("  oracle NaN runs in hops:", [(a / h, (b + 1) / h) for a, b in runs(np.isnan(want[:, ch:ch + 1]))][:12], len(runs(np.isnan(want[:, ch:ch + 1]))))
    print("  engine NaN runs in hops:", [(a / h, (b + 1) / h) for a, b in runs(np.isnan(got[:, ch:ch + 1]))][:12], len(runs(np.isnan(got[:, ch:ch + 1]))))
print("engine inf:", runs(np.isinf(got)))
ok = ~np.isnan(want) & ~np.isnan(got)
print("rms on common finite samples", float(np.sqrt(np.mean((got[ok] - want[ok]) ** 2))))
