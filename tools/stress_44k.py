"""Stress at 44.1 kHz (the wave-per-frame FFT kernels, mask plane, STFT-written power planes): three host threads with
their own contexts run random variants on random clips; every output must equal the single-threaded one bit for bit.
usage: python tools/stress_44k.py [rounds per thread]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd")); sys.path.insert(0, ROOT)
import repet
from repet_synth import synth
fs = 44100
clips = [synth(12 + i, fs, 2, 70 + i) for i in range(3)] + [synth(14, fs, 1, 90)]
algos = ["sim", "extended", "simonline", "adaptive", "original"]
p = repet.derive_params(fs)
want = {}
c0 = repet.Context(0)
for i, x in enumerate(clips):
    c0.upload(x)
    for a in algos:
        c0.execute(a, p); want[(i, a)] = c0.download()
c0.close()
bad = []
def worker(tid, rounds):
    ctx = repet.Context(0)
    rs = np.random.RandomState(tid)
    for r in range(rounds):
        i = rs.randint(len(clips)); a = algos[rs.randint(len(algos))]
        ctx.upload(clips[i]); ctx.execute(a, p); y = ctx.download()
        if not np.array_equal(y, want[(i, a)], equal_nan=True):
            bad.append((tid, r, i, a, float(np.nanmax(np.abs(y - want[(i, a)])))))
    ctx.close()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ts = [threading.Thread(target=worker, args=(t, rounds)) for t in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("pipelines", 3 * rounds, "damaged", len(bad), bad[:5])
