#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED reference.

Runs only in the build container (it needs /root/reference). It imports the reference module
under the name ``repet_reference`` with the two SciPy aliases the reference needs on SciPy >= 1.13
applied from outside, feeds it the deterministic ``synth`` clips, and stores
*data only*: strided float64 samples of ``background_signal``, per-second RMS, and the integer /
float intermediates the reference's private helpers returned during the call (captured by wrapping
the module attributes, not by editing the reference). No reference source text is stored.

    python tests/golden/make_golden.py --cases small,mid,g44k,edge      (~1 min)
    python tests/golden/make_golden.py --cases config                   (~6 CPU-min)
    python tests/golden/make_golden.py --cases groove,groove_config     (second clip family; ~4 CPU-min)
    python tests/golden/make_golden.py --cases cfg1s                    (surrogate of the reference's example clip; ~1 CPU-min)
"""
import argparse
import hashlib
import importlib.util
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "repet-python_amd"))
from repet_synth import synth, synth_groove, synth_song  # noqa: E402

ALGOS = ("original", "extended", "adaptive", "sim", "simonline")

# name -> (duration_s, fs, channels, seed, sample_stride, algorithms)
CASES = {
    "small_mono": (16, 8000, 1, 1, 7, ALGOS),
    "small_stereo": (16, 8000, 2, 2, 7, ALGOS),
    "mid_stereo": (20, 16000, 2, 3, 31, ALGOS),
    "g44k_stereo": (16, 44100, 2, 4, 97, ALGOS),
}
# the second clip family (repet_synth.synth_groove: drifting tempo, broadband transients, level steps, inharmonic partials,
# a bar of digital silence, a bar that changes length) -- same tuple, generated with synth_groove
GROOVE_CASES = {
    "groove_small": (16, 8000, 2, 1, 7, ALGOS),
    "groove_mid": (22, 16000, 2, 2, 31, ALGOS),
    "groove_44k": (20, 44100, 2, 3, 97, ALGOS),
}
GROOVE_CONFIG_CASES = {
    "cfg2_groove": (180, 44100, 2, 0, 1009, ("sim", "original")),
}
CONFIG_CASES = {
    "cfg2_sim": (180, 44100, 2, 0, 1009, ("sim", "original")),
    "cfg3_extended": (600, 44100, 2, 0, 1009, ("extended",)),
    "cfg4_adaptive": (300, 48000, 1, 0, 1009, ("adaptive",)),
    "cfg5_simonline": (30, 44100, 2, 0, 1009, ("simonline",)),
}


def load_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    import scipy.signal
    import scipy.signal.windows as w
    scipy.signal.hamming = w.hamming
    scipy.signal.triang = w.triang
    spec = importlib.util.spec_from_file_location("repet_reference", "/root/reference/repet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Capture:
    """Record what the reference's private helpers return during one public call."""

    NAMES = ("_beatspectrum", "_beatspectrogram", "_periods", "_selfsimilaritymatrix", "_indices",
             "_localmaxima", "_mask", "_adaptivemask", "_simmask")

    def __init__(self, ref):
        self.ref = ref
        self.calls = {}
        self.saved = {}

    def __enter__(self):
        for name in self.NAMES:
            fn = getattr(self.ref, name)
            self.saved[name] = fn
            setattr(self.ref, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for name, fn in self.saved.items():
            setattr(self.ref, name, fn)

    def _wrap(self, name, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            self.calls.setdefault(name, []).append(out)
            return out
        return inner


def pad_indices(lists, width):
    out = np.full((len(lists), width), -1, dtype=np.int32)
    cnt = np.zeros(len(lists), dtype=np.int32)
    for i, ix in enumerate(lists):
        out[i, :len(ix)] = ix
        cnt[i] = len(ix)
    return out, cnt


def per_second_rms(y, fs):
    n = (len(y) // fs) * fs
    if n == 0:
        return np.sqrt(np.mean(y ** 2, axis=0, keepdims=True))
    return np.sqrt(np.mean(y[:n].reshape(-1, fs, y.shape[1]) ** 2, axis=1))


def run_case(ref, name, spec, frame_stride, clip=None, family="synth"):
    dur, fs, ch, seed, stride, algos = spec
    x = (synth_groove if family == "groove" else synth)(dur, fs, ch, seed) if clip is None else clip
    data = {"fs": fs, "duration": dur, "channels": ch, "seed": seed, "sample_stride": stride, "family": family,
            "input_samples": x[::stride].copy(),
            "frame_stride": frame_stride}
    meta = {}
    for algo in algos:
        t0 = time.perf_counter()
        with Capture(ref) as cap, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = getattr(ref, algo)(x.copy(), fs)
        dt = time.perf_counter() - t0
        meta[algo] = {"seconds": round(dt, 3), "sha256_f64": hashlib.sha256(y.tobytes()).hexdigest(),
                      "shape": list(y.shape)}
        print(f"  {name}.{algo}: {dt:.2f}s  rms={np.sqrt(np.mean(y**2)):.6f}", flush=True)
        data[f"{algo}.samples"] = y[::stride].copy()
        data[f"{algo}.rms_per_second"] = per_second_rms(y, fs)
        data[f"{algo}.sum"] = np.sum(y, axis=0)
        c = cap.calls
        if algo == "original":
            data["original.beat_spectrum"] = c["_beatspectrum"][0]
            data["original.period"] = np.int64(c["_periods"][0])
            data["original.mask_c0_rows"] = c["_mask"][0][[0, 1, 5, 6, -1]][:, ::frame_stride]
        elif algo == "extended":
            data["extended.periods"] = np.array(c["_periods"], dtype=np.int64)
            data["extended.beat_spectrum_seg0"] = c["_beatspectrum"][0]
        elif algo == "adaptive":
            bsg = c["_beatspectrogram"][0]
            data["adaptive.periods"] = np.asarray(c["_periods"][0], dtype=np.int64)
            data["adaptive.beat_columns"] = bsg[:, ::max(1, bsg.shape[1] // 8)]
            data["adaptive.mask_c0_rows"] = c["_adaptivemask"][0][[0, 1, 5, 6, -1]][:, ::frame_stride]
        elif algo == "sim":
            s = c["_selfsimilaritymatrix"][0]
            t = s.shape[0]
            data["sim.similarity_columns"] = s[:, [0, t // 2, t - 1]]
            idx, cnt = pad_indices(c["_indices"][0], 100)
            data["sim.index_frames"] = np.arange(0, t, frame_stride)
            data["sim.indices"] = idx[::frame_stride]
            data["sim.counts"] = cnt
            data["sim.mask_c0_rows"] = c["_simmask"][0][[0, 1, 5, 6, -1]][:, ::frame_stride]
        elif algo == "simonline":
            lists = [ix for (_, ix) in c["_localmaxima"]]
            idx, cnt = pad_indices(lists, 100)
            data["simonline.buffer_indices"] = idx[::frame_stride]
            data["simonline.counts"] = cnt
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **data)
    with open(os.path.join(HERE, f"{name}.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


def outcome(ref, algo, x, fs):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y = getattr(ref, algo)(x, fs)
        return {"ok": True, "shape": list(y.shape), "nan_count": int(np.isnan(y).sum()),
                "all_zero": bool(np.all(y == 0)), "dtype": str(y.dtype)}
    except Exception as e:  # noqa: BLE001 - the exception type IS the recorded behaviour
        return {"ok": False, "error": type(e).__name__}


def run_edge(ref):
    fs = 44100
    rec = {}
    base = synth(16, fs, 2, 5)
    rec["one_dim_input"] = {a: outcome(ref, a, base[:, 0].copy(), fs) for a in ALGOS}
    rec["original_2.9s"] = outcome(ref, "original", base[:int(2.9 * fs)].copy(), fs)
    rec["original_3.2s"] = outcome(ref, "original", base[:int(3.2 * fs)].copy(), fs)
    rec["sim_0.5s"] = outcome(ref, "sim", base[:int(0.5 * fs)].copy(), fs)
    rec["simonline_9s"] = outcome(ref, "simonline", base[:9 * fs].copy(), fs)
    rec["simonline_441343"] = outcome(ref, "simonline", base[:441343].copy(), fs)
    rec["simonline_441344"] = outcome(ref, "simonline", base[:441344].copy(), fs)
    rec["simonline_441345"] = outcome(ref, "simonline", base[:441345].copy(), fs)
    x149 = synth(14.9, fs, 2, 5)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rec["extended_14.9s_equals_original"] = bool(
            np.array_equal(ref.extended(x149.copy(), fs), ref.original(x149.copy(), fs)))
    rec["int16_input"] = outcome(ref, "original", (base[:4 * fs] * 32767).astype(np.int16), fs)
    rec["float32_input"] = outcome(ref, "original", base[:4 * fs].astype(np.float32), fs)
    gap = base[:8 * fs].copy()
    gap[100000:140000] = 0.0
    rec["silence_gap_sim"] = outcome(ref, "sim", gap.copy(), fs)
    rec["silence_gap_original"] = outcome(ref, "original", gap.copy(), fs)
    rec["silence_gap_adaptive"] = outcome(ref, "adaptive", gap.copy(), fs)
    # where exactly the NaNs of the silent gap land (first/last NaN sample per channel)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = ref.sim(gap.copy(), fs)
    nan_rows = np.flatnonzero(np.isnan(y).any(axis=1))
    rec["silence_gap_sim"]["nan_first_last"] = [int(nan_rows[0]), int(nan_rows[-1])] if len(nan_rows) else []
    keep = base[:4 * fs].copy()
    before = keep.copy()
    ref.original(keep, fs)
    rec["input_not_mutated"] = bool(np.array_equal(keep, before))
    rec["rounding"] = {"round_312.5": round(312.5), "buffer_frames_8k": round(10 * 8000 / 256)}
    with open(os.path.join(HERE, "edge_cases.json"), "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)
    print(json.dumps(rec, indent=1, sort_keys=True))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="small,mid,g44k,edge")
    args = ap.parse_args()
    want = set(args.cases.split(","))
    ref = load_reference()
    if "small" in want:
        run_case(ref, "small_mono", CASES["small_mono"], 1)
        run_case(ref, "small_stereo", CASES["small_stereo"], 1)
    if "mid" in want:
        run_case(ref, "mid_stereo", CASES["mid_stereo"], 2)
    if "g44k" in want:
        run_case(ref, "g44k_stereo", CASES["g44k_stereo"], 2)
    if "edge" in want:
        run_edge(ref)
    if "config" in want:
        for name, spec in CONFIG_CASES.items():
            run_case(ref, name, spec, 16)
    if "groove" in want:
        run_case(ref, "groove_small", GROOVE_CASES["groove_small"], 1, family="groove")
        run_case(ref, "groove_mid", GROOVE_CASES["groove_mid"], 2, family="groove")
        run_case(ref, "groove_44k", GROOVE_CASES["groove_44k"], 2, family="groove")
    if "groove_config" in want:
        for name, spec in GROOVE_CONFIG_CASES.items():
            run_case(ref, name, spec, 16, family="groove")
    if "cfg1s" in want:
        # The surrogate of BASELINE.json configs[0] that CAN travel to the GPU box: repet_synth.synth_song, the example clip's
        # exact shape (1 014 301 samples, 44.1 kHz stereo, 16-bit PCM values), produced-music-like content, all five variants.
        clip = synth_song(1014301, 44100, 2, 0)
        run_case(ref, "cfg1_surrogate", (1014301 / 44100, 44100, 2, 0, 211, ALGOS), 4, clip=clip, family="song")
    if "cfg1" in want:
        # BASELINE.json configs[0]: the reference's own example clip (README.md:62-75), read the way its wavread
        # does (repet.py:914-931). Only outputs/statistics are stored; the audio itself is not redistributed.
        import scipy.io.wavfile
        fs, pcm = scipy.io.wavfile.read("/root/reference/audio_file.wav")
        clip = pcm / pow(2, pcm.itemsize * 8 - 1)
        run_case(ref, "cfg1_audio_file", (len(pcm) / fs, fs, pcm.shape[1], -1, 211, ALGOS), 4, clip=clip)


if __name__ == "__main__":
    main()
