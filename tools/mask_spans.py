"""Per-wave timeline of the rank-domain mask kernel (diagnostic build: make -C repet-python_amd/csrc stamps).
usage: python tools/mask_spans.py"""
import ctypes, os, sys
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
os.environ["REPET_HIP_LIB"] = os.path.abspath("build_diag/lib_stamps.so")
import repet
from repet_synth import synth
x = synth(180, 44100, 2, 0)
ctx = repet.Context(0); ctx.upload(x); p = repet.derive_params(44100)
ctx.execute("sim", p); ctx.execute("sim", p)
lib = ctypes.CDLL(os.environ["REPET_HIP_LIB"])
n = 7700
buf = (ctypes.c_ulonglong * (4 * n))()
print("rc", lib.repet_debug_mask_spans(buf, n))
a = np.array(buf[:], dtype=np.int64).reshape(n, 4)
a = a[a[:, 1] > 0]
start, end = a[:, 0] * 10e-3, a[:, 1] * 10e-3          # us
dur = end - start
t0 = start.min()
span = end.max() - t0
print("sampled waves %d (one in sixteen)  kernel span %.1f us  wave time: mean %.2f  median %.2f  p90 %.2f  max %.2f us" % (
    len(a), span, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
issued, landed = a[:, 2] & 0xffffffff, a[:, 2] >> 32
print("cycles since the wave's start (median): gathers issued %.0f, all landed %.0f, network done %.0f; wave total %.0f cycles at 2.4 GHz" % (
    np.median(issued), np.median(landed), np.median(a[:, 3]), np.median(dur) * 2400))
print("mean concurrency of all waves (x16): %.0f" % (16 * dur.sum() / span))
for lo in range(0, int(span) + 1, 50):
    alive = 16 * np.sum((start - t0 <= lo) & (end - t0 > lo))
    print("  t = %3d us: about %5d waves alive" % (lo, alive))
