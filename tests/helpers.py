"""Shared helpers for the parity tests (input regeneration, fixture loading, error metrics)."""
import functools
import json
import os

import numpy as np

from repet_synth import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=8)
def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def load_edge_cases():
    with open(os.path.join(GOLDEN, "edge_cases.json")) as fh:
        return json.load(fh)


@functools.lru_cache(maxsize=4)
def golden_input(name):
    """Regenerate the clip of a fixture from the synth formula and check it against the stored
    strided samples (guards against libm differences between hosts)."""
    g = load_golden(name)
    x = synth(float(g["duration"]), int(g["fs"]), int(g["channels"]), int(g["seed"]))
    stride = int(g["sample_stride"])
    assert np.max(np.abs(x[::stride] - g["input_samples"])) < 1e-12, "synth() drifted on this host"
    x.setflags(write=False)
    return x, int(g["fs"])


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def rms_err(a, b):
    return rms(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
