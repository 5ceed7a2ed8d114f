import ctypes, sys, os, numpy as np
sys.path[:0] = ["repet-python_amd", "."]
os.environ["REPET_HIP_LIB"] = os.path.abspath("build_diag/lib_stamps.so")
import repet
from repet_synth import synth
x = synth(180, 44100, 2, 0)
ctx = repet.Context(0); ctx.upload(x); p = repet.derive_params(44100)
ctx.execute("sim", p); ctx.execute("sim", p)
lib = ctypes.CDLL(os.environ["REPET_HIP_LIB"])
buf = (ctypes.c_ulonglong * 64)()
print("rc", lib.repet_debug_peak_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(8, 8)
for row in a:
    d = np.diff(row[:5])
    print("load %6d  passes %6d  test %6d  rank %6d  | total %6d cycles" % (d[0], d[1], d[2], d[3], row[4] - row[0]))
