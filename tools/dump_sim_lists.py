#!/usr/bin/env python3
"""The similarity lists of the bench clip (cfg 2: 180 s, 44.1 kHz, stereo) as a binary file for tools/microbench/bitslice_select:
int32 T, K, pitch then idx[T][pitch] (entries past a list's end 0) and count[T].  usage: dump_sim_lists.py out.bin [seconds]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(root, "repet-python_amd"), root]
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
fs = 44100
x = synth(seconds, fs, 2, 0)
p = repet.derive_params(fs)
c = repet.Context(0)
c.upload(x)
c.execute("sim", p)
T = c.last_frame_count()
idx, cnt = c.last_sim_indices(T, p.sim_number)
pitch = 128
table = np.zeros((T, pitch), dtype=np.int32)
table[:, :idx.shape[1]] = np.maximum(idx, 0)
with open(sys.argv[1], "wb") as f:
    np.array([T, idx.shape[1], pitch], dtype=np.int32).tofile(f)
    table.tofile(f)
    cnt.astype(np.int32).tofile(f)
print("frames", T, "number", idx.shape[1], "mean list", cnt.mean(), "mean |j - t|", np.abs(idx[:, :8] - np.arange(T)[:, None]).mean())
