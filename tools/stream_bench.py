"""Latency of the streaming online separator: push one hop (W/2 samples) at a time, as a live source would."""
import json, sys, time
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
import repet
from repet_synth import synth

fs, ch = 44100, 2
x = synth(40, fs, ch, 0).astype(np.float32)
out = {}
for hops_per_push in (1, 4, 16, 64):
    n = 1024 * hops_per_push
    s = repet.online(fs, ch)
    lat = []
    for pos in range(0, len(x) - n, n):
        t0 = time.perf_counter()
        s.push(x[pos:pos + n])
        lat.append(time.perf_counter() - t0)
    s.finish(); s.close()
    steady = np.array(lat[int(12 * fs / n):]) * 1e3      # after the 10-s warm-up buffer is full
    out[f"{hops_per_push}_hops_per_push"] = {"audio_ms_per_push": round(1e3 * n / fs, 2), "latency_ms_median": round(float(np.median(steady)), 3),
                                             "latency_ms_p95": round(float(np.percentile(steady, 95)), 3),
                                             "x_real_time": round(float(1e3 * n / fs / np.median(steady)), 1)}
print(json.dumps(out, indent=1))
