"""s_memtime stamps of the FFT kernels (diagnostic build: make -C repet-python_amd/csrc stamps). Cycles of wave 0 of one
workgroup in 61, summed over the frames / hops that workgroup handles.
usage: python tools/fft_stamps.py [sim|extended|simonline]"""
import ctypes, os, sys
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
os.environ["REPET_HIP_LIB"] = os.path.abspath("build_diag/lib_stamps.so")
import repet
from repet_synth import synth

algo = sys.argv[1] if len(sys.argv) > 1 else "sim"
seconds = {"sim": 180, "extended": 600, "simonline": 30}[algo]
x = synth(seconds, 44100, 2, 0)
ctx = repet.Context(0); ctx.upload(x); p = repet.derive_params(44100)
lib = ctypes.CDLL(os.environ["REPET_HIP_LIB"])
buf = (ctypes.c_ulonglong * 128)()
ctx.execute(algo, p)
lib.repet_debug_fft_stamps(buf, 1)
if hasattr(lib, "repet_debug_reg_stamps"): lib.repet_debug_reg_stamps(buf, 1)
ctx.execute(algo, p)
print("rc", lib.repet_debug_fft_stamps(buf, 0), algo, ctx.stage_times() if hasattr(ctx, "stage_times") else "")
a = np.array(buf[:], dtype=np.int64).reshape(2, 8, 8)
if a[0].any(): print("stft_pair_kernel (block kernel, REPET_FFT_PATH=block; a run of stereo frames per workgroup):")
for row in (a[0] if a[0].any() else []):
    print("  prologue %6d | wait+window %6d  fetch+barrier %6d  fft %6d  split+stores %6d  mean+norm %6d  rows stored %6d | total %7d" % (
        row[0], row[1], row[2], row[3], row[4], row[5], row[6], row[:7].sum()))
if a[1].any(): print("istft_ola_kernel (block kernel; a run of hops x channels per workgroup):")
for row in (a[1] if a[1].any() else []):
    print("  prologue %6d  frame before %6d | fetch+repack+fft %7d  combine %6d  hop out %6d | total %7d" % (
        row[0], row[1], row[2], row[3], row[4], row[:5].sum()))

if hasattr(lib, "repet_debug_reg_stamps"):
    lib.repet_debug_reg_stamps(buf, 0)
    a = np.array(buf[:64], dtype=np.int64).reshape(8, 8)
    if a.any(): print("stft_reg_kernel (wave-per-frame forward kernel, the default at W = 2048; one wave, its frames x channels):")
    for row in (a if a.any() else []):
        print("  prologue %6d | loads waited %7d  fft %7d  split+stores %7d  mean rows %7d | total %7d" % (row[0], row[1], row[2], row[3], row[4], row[:5].sum()))
