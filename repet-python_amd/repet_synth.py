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


def _burst(rs, n, fs, kind):
    """One fixed percussive hit of `n` samples (the SAME samples at every occurrence: that is what repeats)."""
    t = np.arange(n) / float(fs)
    if kind == "kick":                       # a decaying sine sweep, 120 -> 45 Hz
        f = 45.0 + 75.0 * np.exp(-t / 0.03)
        return np.sin(2 * np.pi * np.cumsum(f) / fs) * np.exp(-t / 0.09)
    noise = rs.standard_normal(n)
    if kind == "snare":                      # broadband noise + a 190 Hz body
        return (0.8 * noise * np.exp(-t / 0.05) + 0.5 * np.sin(2 * np.pi * 190.0 * t) * np.exp(-t / 0.04))
    # hat: differentiated (high-passed) noise, very short
    return np.diff(noise, prepend=0.0) * np.exp(-t / 0.012)


def synth_groove(duration_s, fs, channels, seed=0, silence=True):
    """A second, structurally different clip family (float64 ``(N, channels)`` in [-1, 1]): a drum pattern of broadband
    transients (kick / snare / hat, identical samples at every hit) whose tempo DRIFTS by +-4 % over 23 s and whose bar
    changes from four beats to three in the last third (the repeating period is not constant), an inharmonic bell on some
    beats (partials at 1 : 2.76 : 5.40 : 8.93), level steps of -6 dB / +4 dB over whole passages, one bar of DIGITAL
    SILENCE (exact zeros, no noise floor: all-zero STFT frames) and a non-repeating gliding voice with vibrato in front.
    Nothing in it is tied to the STFT hop, and nothing is shared with ``synth``."""
    rs = np.random.RandomState(1000 + seed)
    n = int(round(duration_s * fs))
    t = np.arange(n) / float(fs)
    hit_len = int(round(0.25 * fs))
    hits = {k: _burst(rs, hit_len, fs, k) for k in ("kick", "snare", "hat")}
    bell_len = int(round(0.9 * fs))
    tb = np.arange(bell_len) / float(fs)
    f_bell = 311.0 + 7.0 * (seed % 4)
    bell = sum(a * np.sin(2 * np.pi * f_bell * r * tb) * np.exp(-tb * d)
               for r, a, d in ((1.0, 1.0, 3.0), (2.76, 0.6, 4.5), (5.40, 0.35, 6.0), (8.93, 0.2, 8.0)))

    # beat times with a drifting tempo: beat length 0.46 s * (1 + 0.04 sin(2 pi t / 23))
    beat0 = 0.46 + 0.01 * (seed % 3)
    times, tt = [], 0.05
    while tt < duration_s:
        times.append(tt)
        tt += beat0 * (1.0 + 0.04 * np.sin(2 * np.pi * tt / 23.0))
    layers = {k: np.zeros(n + bell_len) for k in ("kick", "snare", "hat", "bell")}
    bar_change = 2.0 * duration_s / 3.0
    b = 0
    for k, when in enumerate(times):
        i = int(round(when * fs))
        beats_per_bar = 4 if when < bar_change else 3
        pos = b % beats_per_bar
        b = b + 1 if pos < beats_per_bar - 1 else 0
        if pos == 0:
            layers["kick"][i:i + hit_len] += hits["kick"]
        if pos == 2 or (beats_per_bar == 3 and pos == 1):
            layers["snare"][i:i + hit_len] += hits["snare"]
        layers["hat"][i:i + hit_len] += 0.6 * hits["hat"]
        half = i + int(round(0.5 * beat0 * fs))
        layers["hat"][half:half + hit_len] += 0.35 * hits["hat"][:max(0, min(hit_len, n + bell_len - half))]
        if pos == 1 and (k // 8) % 2 == 0:
            layers["bell"][i:i + bell_len] += bell
    # level steps over whole passages
    gain = np.ones(n)
    gain[(t >= 0.30 * duration_s) & (t < 0.42 * duration_s)] = 0.5
    gain[(t >= 0.55 * duration_s) & (t < 0.62 * duration_s)] = 1.6
    # the voice: a glide with vibrato, five harmonics, slow tremolo
    f_voice = 520.0 + 160.0 * np.sin(2 * np.pi * 0.071 * t + seed) + 14.0 * np.sin(2 * np.pi * 5.3 * t)
    ph = 2 * np.pi * np.cumsum(f_voice) / fs
    voice = sum(np.sin(h * ph) / h for h in range(1, 6)) * (0.45 + 0.35 * np.sin(2 * np.pi * 0.23 * t + 1.0))

    pan = {"kick": 0.0, "snare": 0.25, "hat": -0.4, "bell": 0.5}
    out = np.empty((n, channels))
    for c in range(channels):
        side = 0.0 if channels == 1 else 2.0 * c / (channels - 1) - 1.0           # -1 .. 1
        drums = sum(layers[k][:n] * a * (1.0 + 0.5 * side * pan[k])
                    for k, a in (("kick", 0.5), ("snare", 0.3), ("hat", 0.2), ("bell", 0.22)))
        out[:, c] = 0.8 * gain * (drums + 0.16 * (1.0 - 0.3 * side) * voice) + 1e-3 * rs.standard_normal(n)
    if silence and duration_s >= 12:
        s0 = int(round(0.47 * duration_s * fs))
        out[s0:s0 + int(round(4 * beat0 * fs))] = 0.0                               # one bar of digital silence
    return np.clip(out, -1.0, 1.0)
