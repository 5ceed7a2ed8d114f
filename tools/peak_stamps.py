import ctypes, sys, os, numpy as np
sys.path[:0] = ["repet-python_amd", "."]
os.environ["REPET_HIP_LIB"] = os.path.abspath(os.environ.get("STAMPS_LIB", "build_diag/lib_stamps.so"))
import repet
from repet_synth import synth
algo = os.environ.get("PEAK_ALGO", "sim")                     # sim (180 s) or simonline (30 s): the two users of the peak kernel
x = synth(float(os.environ.get("PEAK_SECONDS", 180 if algo == "sim" else 30)), 44100, 2, 0)
ctx = repet.Context(0); ctx.upload(x); p = repet.derive_params(44100)
ctx.execute(algo, p); ctx.execute(algo, p)
lib = ctypes.CDLL(os.environ["REPET_HIP_LIB"])
buf = (ctypes.c_ulonglong * 64)()
print("rc", lib.repet_debug_peak_stamps(buf))
a = np.array(buf[:], dtype=np.int64).reshape(8, 8)
for row in (a if a.any() else []):
    print("load %6d  passes %6d  test %6d  refine %6d  rank %6d  | total %6d cycles" % (
        row[1] - row[0], row[2] - row[1], row[5] - row[2], row[3] - row[5], row[4] - row[3], row[4] - row[0]))

if hasattr(lib, "repet_debug_wave_stamps"):
    print("wave kernel (peaks_wave.hip), cycles summed over the chunks of one row:")
    print("rc", lib.repet_debug_wave_stamps(buf))
    a = np.array(buf[:], dtype=np.int64).reshape(8, 8)
    for row in (a if a.any() else []):
        print("load+transpose %6d  doubling %6d  sweep %6d  decide %6d  rivals+refine %6d  rank %6d | total %6d" % (
            row[0], row[1], row[2], row[3], row[4], row[5], row[:6].sum()))

if hasattr(lib, "repet_debug_wave_spans"):
    T = ctx.last_frame_count() if algo == "sim" else ctx.last_frame_count() - p.buffer_frames + 1
    sp = (ctypes.c_ulonglong * (2 * T))()
    print("rc", lib.repet_debug_wave_spans(sp, T))
    sp = np.array(sp[:], dtype=np.int64).reshape(T, 2) * 10e-3          # 100 MHz ticks -> microseconds
    t0 = sp[:, 0].min()
    dur = sp[:, 1] - sp[:, 0]
    span = sp[:, 1].max() - t0
    print("rows %d  kernel span %.1f us  row time: mean %.1f  median %.1f  p90 %.1f  max %.1f us  | mean concurrency %.0f waves" % (
        T, span, dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum() / span))
    for lo in range(0, int(span) + 1, 20):
        alive = np.sum((sp[:, 0] - t0 <= lo) & (sp[:, 1] - t0 > lo))
        started = np.sum((sp[:, 0] - t0 >= lo) & (sp[:, 0] - t0 < lo + 20))
        print("  t = %3d us: %5d waves alive, %5d start in the next 20 us" % (lo, alive, started))

if hasattr(lib, "repet_debug_wave_phases") and os.environ.get("PEAK_PHASES"):
    T = ctx.last_frame_count() if algo == "sim" else ctx.last_frame_count() - p.buffer_frames + 1
    ph = (ctypes.c_uint * (10 * T))()
    print("rc", lib.repet_debug_wave_phases(ph, T))
    ph = np.array(ph[:], dtype=np.int64).reshape(T, 10)
    tot = ph[:, :8].sum(axis=1)
    order = np.argsort(tot)
    names = ["load+transpose", "doubling", "sweep", "decide", "verdicts", "rank", "rivals", "float64"]
    def show(rows, label):
        m = ph[rows].mean(axis=0)
        print("  %-22s" % label + "  ".join("%s %6d" % (n, v) for n, v in zip(names, m[:8])) + "  | total %7d  near-ties %5.1f  peaks %5.1f" % (m[:8].sum(), m[8], m[9]))
    show(order[: T // 2], "fastest half")
    show(order[T // 2: -T // 10], "next 40 %")
    show(order[-T // 10: -T // 100], "slow 9 %")
    show(order[-T // 100:], "slowest 1 %")
    print("  slowest rows:", order[-12:].tolist())

if hasattr(lib, "repet_debug_gram_stamps"):
    print("256 x 256 Gram kernel (gram_f16_big.hip), cycles of one workgroup:")
    g = (ctypes.c_ulonglong * 32)()
    print("rc", lib.repet_debug_gram_stamps(g))
    a = np.array(g[:], dtype=np.int64).reshape(8, 4)
    for row in a:
        print("K loop %7d  natural stores %6d  mirror %6d | total %7d" % (row[1] - row[0], row[2] - row[1], row[3] - row[2], row[3] - row[0]))

if hasattr(lib, "repet_debug_gram_spans"):
    T = ctx.last_frame_count()
    nt = -(-T // 256)
    n = nt * (nt + 1) // 2
    sp = (ctypes.c_ulonglong * (2 * n))()
    print("rc", lib.repet_debug_gram_spans(sp, n))
    sp = np.array(sp[:], dtype=np.int64).reshape(n, 2) * 10e-3
    sp = sp[sp[:, 1] > 0]
    t0 = sp[:, 0].min()
    dur = sp[:, 1] - sp[:, 0]
    span = sp[:, 1].max() - t0
    print("Gram workgroups %d  kernel span %.1f us  workgroup time: mean %.1f  min %.1f  max %.1f us | mean concurrency %.0f" % (
        len(sp), span, dur.mean(), dur.min(), dur.max(), dur.sum() / span))
    for lo in range(0, int(span) + 1, 20):
        alive = np.sum((sp[:, 0] - t0 <= lo) & (sp[:, 1] - t0 > lo))
        started = np.sum((sp[:, 0] - t0 >= lo) & (sp[:, 0] - t0 < lo + 20))
        print("  t = %3d us: %4d workgroups alive, %4d start in the next 20 us" % (lo, alive, started))
