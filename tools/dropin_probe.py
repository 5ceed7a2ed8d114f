"""Where the drop-in call repet.sim(x, fs) spends its wall time: upload / execute / download of a resident context, beside
the whole call, for a float64 clip with remainders and for the same clip rounded to 16-bit PCM values."""
import os, sys, time
import numpy as np
sys.path[:0] = ["repet-python_amd", "."]
import repet
from repet_synth import synth
fs = 44100
x = synth(180, fs, 2, 0)
pcm = np.round(x * 32768.0).clip(-32768, 32767) / 32768.0
p = repet.derive_params(fs)
def best(fn, n=6):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts[1:]) * 1e3
for name, clip in (("float64 noisy", x), ("pcm-exact", pcm)):
    ctx = repet.Context(0)
    ctx.upload(clip); ctx.execute("sim", p); ctx.download()
    up = best(lambda: (ctx.upload(clip), ctx.synchronize()))
    ex = best(lambda: ctx.execute("sim", p))
    dn = best(lambda: ctx.download())
    tot = best(lambda: repet.sim(clip, fs))
    print(f"{name:14s} threads {os.environ.get('REPET_HOST_THREADS', 'default'):>7s}: upload {up:5.2f}  execute {ex:5.2f}  download {dn:5.2f}  sum {up + ex + dn:5.2f}  repet.sim {tot:5.2f} ms")
    ctx.close()
