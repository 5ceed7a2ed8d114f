#!/usr/bin/env python3
"""How many 128-byte lines of the sorted columns one wave-load of the median lookup touches (mask_from_codes_kernel): with
the lanes on 64 adjacent bins of one frame (every lane its own column, so every wanted lane its own line) and, for
comparison, on 64 adjacent frames of one bin (one column; neighbouring frames' medians are close in rank). Measured at
cfg 2: 26.8 against 18.2 lines -- not enough to pay for two transposes through LDS.   usage (GPU box): tools/lines_probe.py"""
import os, sys
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(root, "repet-python_amd"), root]
import repet
from repet_synth import synth
fs = 44100
x = synth(180, fs, 2, 0)
p = repet.derive_params(fs)
c = repet.Context(0); c.upload(x); c.execute("sim", p)
codes = c.last_median_codes(1024)
lo = (codes & 0x7fff).astype(np.int64); hi = (codes >> 16).astype(np.int64); need = (codes >> 15) & 1
print("wanted fraction", need.mean(), "mean |hi-lo|", np.abs(hi - lo)[need == 1].mean())
T = codes.shape[1]
# distinct 128-B lines (32 ranks) per wave-load: lanes = 64 adjacent bins of one frame (now) vs 64 adjacent frames of one bin
def distinct(a, n, axis):
    a = np.where(n == 1, a >> 5, -1)
    if axis == 2:   # lanes = bins: every bin its own column -> every wanted lane its own line
        return (n.reshape(2, T, 16, 64).sum(-1)).mean()
    tt = (T // 64) * 64
    b = a[:, :tt].reshape(2, tt // 64, 64, 1024)
    b = np.sort(b, axis=2)
    d = (np.diff(b, axis=2) != 0).sum(2) + 1 - (b[:, :, 0] == -1)
    return d.mean()
print("lines per wave-load, lanes = bins:", distinct(lo, need, 2))
print("lines per wave-load, lanes = frames: lower", distinct(lo, need, 1), "upper", distinct(hi, need, 1))
