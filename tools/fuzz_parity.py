"""Randomised end-to-end parity sweep on the GPU box: random sampling rates, channel counts, durations and module
parameters for all five variants, HIP path vs the float64 oracle. Prints one JSON line per case and a summary;
exit code 1 if any case breaks the 1e-4 RMS bar without being a proven near-tie (tests/helpers.py policy).

usage: python tools/fuzz_parity.py [n_cases] [seed] [edge|stream]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "repet-python_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import repet  # noqa: E402
from repet_synth import synth, synth_groove  # noqa: E402
from oracle import repet_oracle as orc  # noqa: E402

DEFAULTS = {k: getattr(repet, k) for k in ("cutoff_frequency", "period_range", "segment_length", "segment_step",
                                           "filter_order", "similarity_threshold", "similarity_distance",
                                           "similarity_number", "buffer_length")}


def random_case(rs):
    fs = int(rs.choice([4000, 8000, 11025, 16000, 22050, 32000, 44100, 48000, 96000]))
    channels = int(rs.choice([1, 1, 2, 2, 2, 3, 4, 6]))
    algo = str(rs.choice(["original", "extended", "adaptive", "sim", "simonline"]))
    seconds = float(rs.uniform(11.0, 40.0 if fs <= 48000 else 20.0))
    params = {}
    if rs.rand() < 0.6:
        params["cutoff_frequency"] = float(rs.choice([0, 50, 100, 300]))
    if rs.rand() < 0.4:
        lo = float(rs.choice([0.5, 1, 2]))
        params["period_range"] = (lo, float(lo + rs.choice([3, 5, 9])))
    if algo == "extended" and rs.rand() < 0.6:
        seg = float(rs.choice([5, 8, 10]))
        params["segment_length"], params["segment_step"] = seg, float(seg / rs.choice([2, 4]))
    if algo == "adaptive" and rs.rand() < 0.6:
        params["filter_order"] = int(rs.choice([1, 3, 5, 7, 9]))
        seg = float(rs.choice([5, 8, 10]))
        params["segment_length"], params["segment_step"] = seg, float(seg / rs.choice([2, 4]))
    if algo in ("sim", "simonline"):
        if rs.rand() < 0.6:
            params["similarity_number"] = int(rs.choice([1, 3, 10, 40, 100, 150]))
        if rs.rand() < 0.5:
            params["similarity_distance"] = float(rs.choice([0.0, 0.1, 0.5, 1.0, 2.5]))
        if rs.rand() < 0.4:
            params["similarity_threshold"] = float(rs.choice([0.0, 0.3, 0.6, 0.9]))
        if algo == "simonline" and rs.rand() < 0.5:
            params["buffer_length"] = float(rs.choice([3, 5, 10]))
    return algo, fs, channels, seconds, params


def edge_case(rs):
    """Degenerate corners: clips around the minimum lengths, odd parameter values, inverted ranges."""
    fs = int(rs.choice([4000, 8000, 16000, 22050, 44100, 48000]))
    channels = int(rs.choice([1, 2, 3, 5]))
    algo = str(rs.choice(["original", "extended", "adaptive", "sim", "simonline"]))
    seconds = float(rs.choice([0.3, 1.0, 2.9, 3.3, 6.0, 9.9, 10.2, 12.0, 15.1, 21.0, 26.0]) + rs.uniform(0, 0.4))
    params = {}
    if rs.rand() < 0.5:
        params["cutoff_frequency"] = float(rs.choice([0, 1, 99.9, 1000, fs / 2, fs]))
    if rs.rand() < 0.5:
        params["period_range"] = tuple(float(v) for v in rs.choice([0, 0.3, 1, 2, 4, 8, 15], size=2))
    if rs.rand() < 0.6:
        params["segment_length"] = float(rs.choice([1, 3, 5, 10, 20]))
        params["segment_step"] = float(rs.choice([0.5, 1, 2.5, 5, 7, 12]))
    if rs.rand() < 0.5:
        params["filter_order"] = int(rs.choice([1, 2, 4, 5, 8, 12, 30]))
    if rs.rand() < 0.6:
        params["similarity_number"] = int(rs.choice([1, 2, 7, 64, 100, 128, 129, 300]))
    if rs.rand() < 0.5:
        params["similarity_distance"] = float(rs.choice([0.0, 0.01, 0.3, 1.0, 5.0, 12.0]))
    if rs.rand() < 0.4:
        params["similarity_threshold"] = float(rs.choice([0.0, 0.5, 0.99, 1.0]))
    if rs.rand() < 0.5:
        params["buffer_length"] = float(rs.choice([0.5, 2, 5, 10, 20]))
    return algo, fs, channels, seconds, params


def run_case(k, algo, fs, channels, seconds, params):
    from helpers import assert_parity_modulo_near_ties, rms_err
    # both clip families: the decaying-note clips and (every third case) the drum clips with a silent bar
    x = synth_groove(seconds, fs, channels, 1000 + k) if k % 3 == 2 else synth(seconds, fs, channels, 1000 + k)
    for name, value in DEFAULTS.items():
        setattr(repet, name, params.get(name, value))
    rec = {"case": k, "algo": algo, "fs": fs, "channels": channels, "seconds": round(seconds, 2), "params": params}
    t0 = time.perf_counter()
    try:
        try:
            want = orc.ALGORITHMS[algo](x, fs, orc.Params(**params))
            want_err = None
        except Exception as e:  # noqa: BLE001 - the exception type is the expected behaviour
            want, want_err = None, type(e).__name__
        try:
            got = getattr(repet, algo)(x, fs)
            got_err = None
        except Exception as e:  # noqa: BLE001
            got, got_err = None, type(e).__name__
        if want_err or got_err:
            rec["oracle_error"], rec["engine_error"] = want_err, got_err
            rec["ok"] = (want_err is not None) == (got_err is not None)
        else:
            same_nan = bool(np.array_equal(np.isnan(got), np.isnan(want)))
            okm = ~np.isnan(want) & ~np.isnan(got)
            err = rms_err(got[okm], want[okm]) if okm.any() else 0.0
            rec["rms"] = float(err)
            rec["same_nan"] = same_nan
            if err <= 1e-4 and same_nan:
                rec["ok"] = True
            elif algo in ("sim", "simonline") and params.get("similarity_threshold", 0.0) >= 1.0 and err <= 1e-4:
                # a threshold of exactly 1 asks whether the float64 self-similarity of a frame rounded to >= 1.0:
                # which frames keep themselves (mask 1) and which get an empty list (NaN) is rounding noise of
                # the reference itself; only the non-NaN samples are compared
                rec["ok"] = True
                rec["ill_conditioned"] = "similarity_threshold >= 1"
            elif algo in ("sim", "simonline") and same_nan:
                # (round 2 accepted lists that differed at proven float64 near-ties; with the second level of the peak
                # picking the lists are the oracle's, so this is a failure -- the analysis only says where)
                outcome = assert_parity_modulo_near_ties(algo, x, fs, params, require_strict=False)
                rec["near_tie_rows"], rec["branch"] = int(outcome), outcome.branch
                rec["ok"] = os.environ.get("FUZZ_ALLOW_TIES") == "1"
            else:
                rec["ok"] = False
    except AssertionError as e:
        rec["ok"] = False
        rec["assertion"] = str(e)[:300]
    rec["wall_s"] = round(time.perf_counter() - t0, 2)
    return rec


def stream_case(k, rs):
    """Streaming handle vs the offline call: random chunkings and parameters, bit-exact."""
    fs = int(rs.choice([8000, 16000, 22050, 44100, 48000]))
    channels = int(rs.choice([1, 2, 4]))
    params = {"buffer_length": float(rs.choice([2, 3, 5, 10])), "similarity_number": int(rs.choice([1, 5, 30, 100, 200])),
              "similarity_distance": float(rs.choice([0.0, 0.2, 1.0])), "similarity_threshold": float(rs.choice([0.0, 0.4])),
              "cutoff_frequency": float(rs.choice([0, 100, 500]))}
    seconds = params["buffer_length"] + float(rs.uniform(0.5, 15.0))
    for name, value in DEFAULTS.items():
        setattr(repet, name, params.get(name, value))
    x = synth(seconds, fs, channels, 5000 + k)
    rec = {"case": k, "mode": "stream", "fs": fs, "channels": channels, "seconds": round(seconds, 2), "params": params}
    try:
        want = repet.simonline(x, fs)
    except Exception as e:  # noqa: BLE001
        rec["offline_error"] = type(e).__name__
        want = None
    try:
        stream = repet.online(fs, channels)
        pieces, pos = [], 0
        scale = int(rs.choice([1, 50, 700, 5000, 60000]))
        while pos < len(x):
            n = min(1 + int(scale * rs.rand() * 2), len(x) - pos)
            pieces.append(stream.push(x[pos:pos + n]))
            pos += n
        pieces.append(stream.finish())
        stream.close()
        got = np.concatenate(pieces, axis=0)
    except Exception as e:  # noqa: BLE001
        rec["stream_error"] = type(e).__name__
        got = None
    if want is None or got is None:
        rec["ok"] = (want is None) == (got is None)
    else:
        rec["ok"] = bool(got.shape == want.shape and np.array_equal(got, want, equal_nan=True))
    return rec


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "stream":
        rs = np.random.RandomState(seed)
        recs = [stream_case(k, rs) for k in range(n)]
        for rec in recs:
            print(json.dumps(rec), flush=True)
        bad = sum(not r["ok"] for r in recs)
        print(json.dumps({"cases": n, "failed": bad}))
        return 1 if bad else 0
    gen = edge_case if (len(sys.argv) > 3 and sys.argv[3] == "edge") else random_case
    rs = np.random.RandomState(seed)
    bad = 0
    worst = 0.0
    for k in range(n):
        rec = run_case(k, *gen(rs))
        print(json.dumps(rec), flush=True)
        bad += not rec["ok"]
        worst = max(worst, rec.get("rms", 0.0))
    print(json.dumps({"cases": n, "failed": bad, "worst_rms": worst}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
