"""Diagnostic: block-level timeline of the similarity GEMM (needs build_diag/lib_gramtrace.so, -DREPET_GRAM_TRACE)."""
import ctypes, os, sys
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
os.environ["REPET_HIP_LIB"] = os.path.abspath("build_diag/lib_gramtrace.so")
import repet
from repet_synth import synth
x = synth(180, 44100, 2, 0)
ctx = repet.Context(0); ctx.upload(x); p = repet.derive_params(44100)
ctx.execute("sim", p); ctx.execute("sim", p)
lib = ctypes.CDLL(os.environ["REPET_HIP_LIB"])
buf = (ctypes.c_ulonglong * (4096 * 3))()
lib.repet_debug_gram_trace(buf)
a = np.array(buf[:], dtype=np.uint64).reshape(4096, 3)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.int64) / 100.0     # us (100 MHz)
end = (a[:, 1] - t0).astype(np.int64) / 100.0
xcc = (a[:, 2] >> np.uint64(32)).astype(int)
hw = (a[:, 2] & np.uint64(0xffffffff)).astype(int)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
print("blocks traced", len(a), "kernel span us", end.max(), "mean block us", (end - start).mean())
print("block duration percentiles us", np.percentile(end - start, [0, 10, 50, 90, 100]).round(1))
# occupancy over time
ts = np.linspace(0, end.max(), 41)
occ = [(int(((start <= t) & (end > t)).sum())) for t in ts]
print("resident blocks at 40 time points:", occ)
print("per-XCC block counts", np.bincount(xcc, minlength=8), "per-XCC last end us", [round(float(end[xcc == k].max()), 1) for k in range(8)])
ucu, cnt = np.unique(cuid, return_counts=True)
print("distinct CUs", len(ucu), "blocks per CU min/median/max", cnt.min(), int(np.median(cnt)), cnt.max())
late = np.argsort(end)[-10:]
print("last blocks: start/end", [(round(float(start[i]), 1), round(float(end[i]), 1)) for i in late])
