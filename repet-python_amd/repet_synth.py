"""Deterministic synthetic clips for the REPET parity tests and bench.py.

This is the ``synth(duration_s, fs, channels, seed)`` recipe of SURVEY.md section 8d: a repeating
four-note background (period 1.7-2.1 s, never a multiple of the STFT hop), a non-repeating
frequency-modulated foreground, and a 1e-3 white-noise floor that keeps every frame non-silent
(no NaN path, no exact ties). It is input generation only - no part of the separation path.
"""
import numpy as np

_NOTES_HZ = (110.0, 164.81, 220.0, 329.63)


def synth(duration_s, fs, channels, seed=0):
    """Return a float64 ``(N, channels)`` mixture in [-1, 1], ``N = round(duration_s * fs)``."""
    rs = np.random.RandomState(seed)  # legacy MT19937: stream is frozen across NumPy versions
    n = int(round(duration_s * fs))
    t = np.arange(n) / float(fs)

    period = 1.7 + 0.1 * (seed % 5)
    phase = np.mod(t, period) / period
    background = np.zeros(n)
    for k, f0 in enumerate(_NOTES_HZ):
        gate = (phase >= k / 4.0) & (phase < (k + 1) / 4.0)
        envelope = np.exp(-8.0 * np.mod(phase - k / 4.0, 0.25))
        tone = (np.sin(2 * np.pi * f0 * t) + 0.5 * np.sin(2 * np.pi * 2 * f0 * t)
                + 0.25 * np.sin(2 * np.pi * 3 * f0 * t))
        background += gate * envelope * tone

    f_inst = (440.0 + 200.0 * np.sin(2 * np.pi * 0.13 * t)
              + 90.0 * np.sin(2 * np.pi * 0.031 * t * t / max(duration_s, 1)))
    foreground = (0.6 * np.sin(2 * np.pi * np.cumsum(f_inst) / fs)
                  * (0.5 + 0.5 * np.sin(2 * np.pi * 0.37 * t)))

    out = np.empty((n, channels))
    for c in range(channels):
        out[:, c] = (0.25 * background * (1 - 0.2 * c)
                     + 0.25 * (0.5 + 0.3 * (c - (channels - 1) / 2.0)) * foreground
                     + 1e-3 * rs.standard_normal(n))
    return np.clip(out, -1.0, 1.0)
