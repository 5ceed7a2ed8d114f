"""Uninitialised-read probe: fill the register files / LDS of the device with a NaN pattern (tools/microbench/
reg_poison.hip) before and beside each pipeline and compare every output with the clean run's. A kernel that reads a
register or LDS word it never wrote shows up as a mismatch here even with a single context.
usage: python tools/poison_stress.py [rounds]   (needs tools/microbench/libreg_poison.so)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "microbench", "libreg_poison.so"))
lib.poison_launch.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fs = 16000
x = synth(12, fs, 2, 70)
p = repet.derive_params(fs)
ctx = repet.Context(0)
ctx.upload(x)
for algo in ("original", "extended", "adaptive", "sim", "simonline"):
    ctx.execute(algo, p)
    want = ctx.download()
    for what, label in ((1, "registers"), (2, "LDS"), (3, "both")):
        for pattern in (0x7FC00000, 0x7F7FFFFF, 0xFFFFFFFF):
            bad_serial = bad_beside = 0
            for r in range(rounds):
                assert lib.poison_launch(what, pattern, 2048, 1) == 0
                ctx.execute(algo, p)
                bad_serial += not np.array_equal(ctx.download(), want, equal_nan=True)
                assert lib.poison_launch(what, pattern, 4096, 0) == 0
                ctx.execute_async(algo, p)
                assert lib.poison_launch(what, pattern, 4096, 0) == 0
                ctx.synchronize()
                bad_beside += not np.array_equal(ctx.download(), want, equal_nan=True)
            print(f"{algo:10s} poison {label:9s} pattern {pattern:08x}: after {bad_serial}/{rounds}, beside {bad_beside}/{rounds}", flush=True)
