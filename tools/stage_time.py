"""Per-stage device times of one variant on a synthetic clip, without output checks (for ablation builds).
usage: python tools/stage_time.py [algo] [seconds] [fs] [channels]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

algo = sys.argv[1] if len(sys.argv) > 1 else "sim"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
fs = int(sys.argv[3]) if len(sys.argv) > 3 else 44100
channels = int(sys.argv[4]) if len(sys.argv) > 4 else 2
ctx = repet.Context(0)
ctx.upload(synth(seconds, fs, channels, 0))
p = repet.derive_params(fs)
for _ in range(3):
    ctx.execute(algo, p)
acc = {}
n = 10
for _ in range(n):
    for s in ctx.execute(algo, p, timing=True)["stages"]:
        acc[s["name"]] = acc.get(s["name"], 0.0) + s["ms"]
print({k: round(v / n, 4) for k, v in acc.items()})
