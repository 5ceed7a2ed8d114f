"""Where a row of the second peak-picking level (peaks_exact.hip) spends its time: 100 MHz ticks per phase, summed over the
rows, from a library built with -DREPET_EXACT_STAMPS (build_diag/lib_exact_stamps.so). GPU box:
    REPET_HIP_LIB=build_diag/lib_exact_stamps.so python tools/exact_phases.py [seconds fs channels]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
sys.path.insert(0, ROOT)
import repet  # noqa: E402
from repet import _native  # noqa: E402
from repet_synth import synth  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 44100
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 2
x = synth(dur, fs, ch, 0)
ctx = _native.default_context(0)
ctx.upload(x)
ctx.execute("sim", repet.derive_params(fs))
out = (C.c_int64 * 6)()
lib = _native.lib()
lib.repet_debug_exact_phases.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
lib.repet_debug_exact_phases(ctx._h, out)
ex = ctx.last_exact_stats()
rows = max(ex["rows_exact"], 1)
names = ["scan+rivals", "level1", "close?", "level2 (fft+dots)", "verdicts", "rank+cut+store"]
u = (C.c_uint64 * 8)()
try:
    lib.repet_debug_unit_stamps.argtypes = [C.POINTER(C.c_uint64)]
    lib.repet_debug_unit_stamps(u)
    n = max(ex["unit_rows_f64"], 1)
    print(json.dumps({"unit_rows": ex["unit_rows_f64"], "us_per_unit_row": {k: round(u[i] / 100.0 / n, 2) for i, k in enumerate(["loads", "fft", "split", "norm+store"])}}))
except AttributeError:
    pass
print(json.dumps({"rows": ex["rows_exact"], "us_per_row": {n: round(out[k] / 100.0 / rows, 2) for k, n in enumerate(names)}, **ex}))
