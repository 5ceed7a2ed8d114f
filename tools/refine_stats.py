"""Near-tie refinement counters of repet.sim on the headline clip (run on the GPU box)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet import _native  # noqa: E402
from repet_synth import synth  # noqa: E402

dur = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 44100
x = synth(dur, fs, 2, 0)
ctx = _native.default_context(0)
ctx.upload(x)
ctx.execute("sim", repet.derive_params(fs))
print(json.dumps({"clip": f"{dur}s {fs}Hz", "frames": ctx.last_frame_count(), **ctx.last_refine_stats(), **ctx.last_exact_stats()}))
