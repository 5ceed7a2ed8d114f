"""One host thread, two contexts, execute_async on both: device-side overlap without host threading.
usage: REPET_NO_CHAIN=1 python tools/overlap_stress.py [rounds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
fs = 16000
clips = [synth(9, fs, 2, 70 + i) for i in range(2)]
p = repet.derive_params(fs)
ctxs = [repet.Context(0), repet.Context(0)]
want = []
for c, x in zip(ctxs, clips):
    c.upload(x)
    c.execute("sim", p)
    want.append(c.download())
bad = 0
for r in range(rounds):
    for c in ctxs:
        c.execute_async("sim", p)
    for c in ctxs:
        c.synchronize()
    for k, c in enumerate(ctxs):
        if not np.array_equal(c.download(), want[k]):
            bad += 1
print("mismatching runs:", bad, "of", 2 * rounds, "(one thread, two contexts, async)")
