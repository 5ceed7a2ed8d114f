"""One host thread, two contexts, execute_async on both: device-side overlap without host threading.
usage: python tools/overlap_stress.py [rounds] [algo_a algo_b]   """
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
algos = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("sim", "sim")
fs = 16000
clips = [synth(9, fs, 2, 70 + i) for i in range(2)]
p = repet.derive_params(fs)
ctxs = [repet.Context(0), repet.Context(0)]
want = []
for c, x, a in zip(ctxs, clips, algos):
    c.upload(x)
    c.execute(a, p)
    want.append(c.download())
bad = [0, 0]
for r in range(rounds):
    for c, a in zip(ctxs, algos):
        c.execute_async(a, p)
    for c in ctxs:
        c.synchronize()
    for k, c in enumerate(ctxs):
        if not np.array_equal(c.download(), want[k]):
            bad[k] += 1
print("mismatching runs per context:", bad, "of", rounds, "each; algos", algos)
