"""Which kernel pair shows the packed-fp32 interaction (DESIGN.md "Contexts and concurrency")? Thread A repeats ONE
stage export on its own context and compares every result with its first; thread B repeats another stage on a second
context. With a library built WITH packed-fp32 ops (REPET_HIP_LIB=...) about one STFT result in five differs beside
the f16 similarity kernels; with the shipped library none may .
usage: [REPET_HIP_LIB=path/to/lib_with_packed_fp32.so] python tools/pk_pairs.py [iterations] [victim:aggressor]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet import _native  # noqa: E402
from repet_synth import synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lib = _native.lib()
fs, W, H = 16000, 1024, 512
x = np.ascontiguousarray(synth(9, fs, 1, 3)[:, 0], dtype=np.float32)
window = np.ascontiguousarray(0.54 - 0.46 * np.cos(2 * np.pi * np.arange(W) / W), dtype=np.float32)
T = lib.repet_frame_count(len(x), W, H, 1)
F = W // 2 + 1
rs = np.random.RandomState(0)
rows = np.ascontiguousarray(np.abs(rs.randn(2048, F)), dtype=np.float32)
spec0 = np.ascontiguousarray(rs.randn(T, F, 2), dtype=np.float32)
spec0[:, 0, 1] = 0
spec0[:, -1, 1] = 0
p = _native.ptr


def stage_stft(ctx, out):
    _native.check(lib.repet_stft(ctx.handle, p(x), len(x), p(window), W, H, 1, p(out), T))


def stage_istft(ctx, out):
    _native.check(lib.repet_istft(ctx.handle, p(spec0), T, p(window), W, H, p(out), len(out)))


def stage_selfsim(ctx, out):
    _native.check(lib.repet_selfsim(ctx.handle, p(rows), rows.shape[0], F, p(out)))


def stage_similarity(ctx, out):
    _native.check(lib.repet_similarity(ctx.handle, p(rows), rows.shape[0], p(rows), rows.shape[0], F, p(out)))


def stage_acorr(ctx, out):
    _native.check(lib.repet_acorr(ctx.handle, p(rows), rows.shape[0], F, p(out)))


STAGES = {
    "stft": (stage_stft, lambda: np.empty((T, F, 2), np.float32)),
    "istft": (stage_istft, lambda: np.empty(T * H - (W - H), np.float32)),
    "selfsim": (stage_selfsim, lambda: np.empty((2048, 2048), np.float32)),
    "similarity": (stage_similarity, lambda: np.empty((2048, 2048), np.float32)),
    "acorr": (stage_acorr, lambda: np.empty((2048, F), np.float32)),
}


def pair(victim, aggressor):
    ca, cb = repet.Context(0), repet.Context(0)
    fn_a, mk_a = STAGES[victim]
    want = mk_a()
    fn_a(ca, want)
    bad, done = [0], [False]

    def run_a():
        out = mk_a()
        for _ in range(iters):
            fn_a(ca, out)
            bad[0] += not np.array_equal(out, want, equal_nan=True)
        done[0] = True

    def run_b():
        if aggressor is None:
            return
        fn_b, mk_b = STAGES[aggressor]
        out = mk_b()
        while not done[0]:
            fn_b(cb, out)

    t0 = time.perf_counter()
    ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
    ta.start(); tb.start(); ta.join(); tb.join()
    print(f"victim {victim:10s} beside {str(aggressor):10s}: {bad[0]} of {iters} differ ({time.perf_counter() - t0:.1f} s)", flush=True)
    ca.close(); cb.close()


only = sys.argv[2].split(":") if len(sys.argv) > 2 else None
for victim in ("stft", "istft"):
    for aggressor in (None, "selfsim", "similarity", "acorr", "stft"):
        if only is None or (victim == only[0] and str(aggressor) == only[1]):
            pair(victim, aggressor)
