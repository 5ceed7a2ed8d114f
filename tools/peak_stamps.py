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
    print("load %6d  passes %6d  test %6d  refine %6d  rank %6d  | total %6d cycles" % (
        row[1] - row[0], row[2] - row[1], row[5] - row[2], row[3] - row[5], row[4] - row[3], row[4] - row[0]))

if hasattr(lib, "repet_debug_wave_stamps"):
    print("wave kernel (peaks_wave.hip), cycles summed over the chunks of one row:")
    print("rc", lib.repet_debug_wave_stamps(buf))
    a = np.array(buf[:], dtype=np.int64).reshape(8, 8)
    for row in a:
        print("load+transpose %6d  doubling %6d  sweep %6d  decide %6d  rivals+refine %6d  rank %6d | total %6d" % (
            row[0], row[1], row[2], row[3], row[4], row[5], row[:6].sum()))

if hasattr(lib, "repet_debug_gram_stamps"):
    print("256 x 256 Gram kernel (gram_f16_big.hip), cycles of one workgroup:")
    g = (ctypes.c_ulonglong * 32)()
    print("rc", lib.repet_debug_gram_stamps(g))
    a = np.array(g[:], dtype=np.int64).reshape(8, 4)
    for row in a:
        print("K loop %7d  natural stores %6d  mirror %6d | total %7d" % (row[1] - row[0], row[2] - row[1], row[3] - row[2], row[3] - row[0]))
