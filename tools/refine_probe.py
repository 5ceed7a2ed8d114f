"""Experiment: how many peak-picking rows differ from the float64 oracle (a) on the shipped fp32 path,
(b) if the similarity of the same fp32 spectra were accumulated in float64. Run on the GPU box."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
sys.path.insert(0, ROOT)

import repet  # noqa: E402
from repet import _native  # noqa: E402
from repet_synth import synth  # noqa: E402
from oracle import repet_oracle as oracle  # noqa: E402


def rows_differing(a, b):
    return sum(1 for x, y in zip(a, b) if set(map(int, x)) != set(map(int, y)))


def probe(duration, fs, channels, seed):
    x = synth(duration, fs, channels, seed)
    w, window, h = oracle.stft_geometry(fs)
    tr = oracle.Trace()
    oracle.sim(x, fs, trace=tr)
    ref = tr.items["similarity_indices"]
    t = len(ref)
    ctx = _native.default_context(0)
    ctx.upload(x)
    ctx.execute("sim", repet.derive_params(fs))
    idx, cnt = ctx.last_sim_indices(t, repet.similarity_number)
    gpu = [idx[i, :cnt[i]] for i in range(t)]
    stats = ctx.last_refine_stats()
    mags = []
    for c in range(channels):
        full = repet._stft(x[:, c], window, h)
        mags.append(np.abs(full[: w // 2 + 1]))
    v = np.mean(np.stack(mags, axis=2), axis=2)
    s64 = oracle.selfsimilaritymatrix(v)
    dist = int(round(repet.similarity_distance * fs / h))
    emu = oracle.indices(s64, repet.similarity_threshold, dist, repet.similarity_number)
    s_ref = tr.items["similarity_matrix"]
    return {"clip": f"{duration}s {fs}Hz {channels}ch seed{seed}", "rows": t,
            "rows_differing_gpu": rows_differing(gpu, ref), "refine": stats, "rows_differing_f64_accumulate": rows_differing(emu, ref),
            "max_abs_S_err_f64_accumulate": float(np.nanmax(np.abs(s64 - s_ref)))}


def ambiguity(duration, fs, channels, seed):
    """Fraction of rows whose peak decisions sit within delta of a tie, and the fp32 Gram error."""
    x = synth(duration, fs, channels, seed)
    w, window, h = oracle.stft_geometry(fs)
    spec, mag = oracle.spectrogram_channels(x, window, h)
    v = np.mean(mag, axis=2)
    s = oracle.selfsimilaritymatrix(v)
    s32 = repet._selfsimilaritymatrix(v)
    err = np.abs(s32 - s)
    d = int(round(repet.similarity_distance * fs / h))
    n = s.shape[0]
    out = {"clip": f"{duration}s {fs}Hz {channels}ch", "rows": n, "d": d, "max_S_err_fp32": float(np.nanmax(err)),
           "rms_S_err_fp32": float(np.sqrt(np.nanmean(err ** 2)))}
    pad = np.full((n, d), -np.inf)
    sp = np.concatenate([pad, s, pad], axis=1)
    # max of the others in the window, by brute force over offsets
    m = np.full_like(s, -np.inf)
    for o in range(1, d + 1):
        m = np.maximum(m, sp[:, d - o:d - o + n])
        m = np.maximum(m, sp[:, d + o:d + o + n])
    for delta in (1e-6, 2e-6, 4e-6):
        near = (np.abs(s - m) <= delta) & (s >= -delta)
        rows = int(np.count_nonzero(near.any(axis=1)))
        out[f"rows_near_tie_{delta:g}"] = rows
        out[f"elements_near_tie_{delta:g}"] = int(np.count_nonzero(near))
        # cut ambiguity
        cut_rows = 0
        peaks = (s > m) & (s >= 0)
        for r in range(n):
            vals = np.sort(s[r][peaks[r]])[::-1]
            k = repet.similarity_number
            if len(vals) > k and vals[k - 1] - vals[k] <= delta:
                cut_rows += 1
        out[f"rows_cut_tie_{delta:g}"] = cut_rows
    return out


if __name__ == "__main__":
    for o in (probe(60, 22050, 2, 1), probe(20, 96000, 1, 3), probe(40, 44100, 2, 2), probe(90, 16000, 2, 4)):
        print(json.dumps(o))
    if "--ambiguity" in sys.argv:
        for o in (ambiguity(60, 22050, 2, 1), ambiguity(40, 44100, 2, 2)):
            print(json.dumps(o))
