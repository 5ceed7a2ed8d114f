"""Stress: two host threads with their own contexts running repet.sim concurrently; counts runs whose output differs
from the single-threaded result. usage: python tools/thread_stress.py [rounds]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
import repet  # noqa: E402
from repet_synth import synth  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 2
algo = sys.argv[3] if len(sys.argv) > 3 else "sim"
fs = 16000
clips = [synth(9, fs, 2, 70 + i) for i in range(4)]
want, want_idx = [], []
_ctx = repet.Context(0)
_p = repet.derive_params(fs)
for x in clips:
    _ctx.upload(x)
    _ctx.execute(algo, _p)
    want.append(_ctx.download())
    want_idx.append(_ctx.last_sim_indices(_ctx.last_frame_count(), _p.sim_number) if algo == "sim" else None)
_ctx.close()
bad = []


def work(ids):
    ctx = repet.Context(0)
    p = repet.derive_params(fs)
    for r in range(rounds):
        for i in ids:
            ctx.upload(clips[i])
            ctx.execute(algo, p)
            y = ctx.download()
            if not np.array_equal(y, want[i]):
                d = np.abs(y - want[i])
                nz = np.flatnonzero(d.max(axis=1))
                other = [k for k in ids if k != i][0]
                stale = float(np.abs(y[nz] - want[other][nz]).max())        # does the damage equal the other clip's output?
                rows, detail = [], []
                extra = [int(nz[0]), int(nz[-1]), len(nz), [int(np.count_nonzero(d[:, c])) for c in range(d.shape[1])], stale]
                if algo == "sim":
                    idx, cnt = ctx.last_sim_indices(ctx.last_frame_count(), p.sim_number)
                    rows = [int(k) for k in np.flatnonzero(np.any(idx != want_idx[i][0], axis=1) | (cnt != want_idx[i][1]))]
                    detail = [(k, idx[k, :cnt[k]].tolist(), want_idx[i][0][k, :want_idx[i][1][k]].tolist()) for k in rows[:2]]
                if os.environ.get("REPET_STRESS_DETAIL") and len(bad) < 3:
                    W = p.window_length
                    rel = nz - (nz[0] // (W // 2)) * (W // 2)
                    print("detail clip", i, "first", int(nz[0]), "W", W, "rel rows", rel[:40].tolist(), "...", rel[-8:].tolist(), flush=True)
                    print("  residues mod 4:", np.bincount(rel % 4, minlength=4).tolist(), "diffs ch0", (y - want[i])[nz[:12], 0].tolist(),
                          "ch1", (y - want[i])[nz[:12], 1].tolist(), flush=True)
                    dd = (y - want[i])[nz]
                    print("  |diff| min/median/max", float(np.abs(dd).max(axis=1).min()), float(np.median(np.abs(dd).max(axis=1))), float(np.abs(dd).max()), flush=True)
                bad.append((r, i, float(d.max()), int(np.count_nonzero(d)), int(np.flatnonzero(d.max(axis=1))[0]), rows, detail, extra))
    ctx.close()


import time  # noqa: E402
_t0 = time.perf_counter()
threads = [threading.Thread(target=work, args=(ids,)) for ids in ([[0, 2], [1, 3]] if n_threads == 2 else [[0, 1, 2, 3]])]
for th in threads:
    th.start()
for th in threads:
    th.join()
_dt = time.perf_counter() - _t0
print("mismatching runs:", len(bad), "of", rounds * 4, "threads", n_threads, algo, f"-- {_dt:.1f} s, {rounds * 4 / _dt:.0f} clips/s")
for b in bad[:10]:
    print(b)
