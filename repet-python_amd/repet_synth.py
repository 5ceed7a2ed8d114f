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


def synth_song(number_samples, fs, channels=2, seed=0):
    """A stand-in for a short excerpt of produced music of EXACTLY ``number_samples`` samples, as 16-bit PCM would hold it
    (every value a multiple of 2^-15, like ``wavread`` of an int16 file): a four-bar loop of 6.64 s -- drums with per-hit
    level humanisation, an eighth-note bass line and a detuned chord pad following a four-chord progression -- under a
    non-repeating sung line (pentatonic random walk, vibrato, a formant-shaped harmonic series, phrases and rests), with an
    arrangement: an intro without drums, a one-bar breakdown, a fade-out. It is the surrogate of BASELINE.json configs[0]
    (the reference's ``audio_file.wav``: 1 014 301 samples, 44.1 kHz stereo, which cannot be redistributed): same shape,
    same kind of spectrum and level profile, a repeating period of about 286 frames of 1 024 samples."""
    rs = np.random.RandomState(7000 + seed)
    n = int(number_samples)
    t = np.arange(n) / float(fs)
    beat = 6.64 / 16.0                                     # sixteen beats to the loop
    n_beats = int(np.ceil(t[-1] / beat)) + 1 if n else 0
    hit_len = int(round(0.22 * fs))
    hits = {k: _burst(rs, hit_len, fs, k) for k in ("kick", "snare", "hat")}
    layers = {k: np.zeros(n + hit_len) for k in ("kick", "snare", "hat")}
    drums_on = lambda when: when >= 8 * beat and not (36 * beat <= when < 40 * beat)          # intro, breakdown
    crash_len = int(round(1.4 * fs))
    tc = np.arange(crash_len) / float(fs)
    crash = np.diff(rs.standard_normal(crash_len), prepend=0.0) * np.exp(-tc / 0.45)         # the same cymbal at every loop start
    layers["crash"] = np.zeros(n + crash_len)
    for b in range(n_beats):
        when = 0.02 + b * beat
        i = int(round(when * fs))
        if i >= n or not drums_on(when):
            continue
        level = 10.0 ** (rs.uniform(-1.0, 1.0) / 20.0)                                       # +-1 dB per hit
        if b % 16 == 0:
            layers["crash"][i:i + crash_len] += level * crash
        if b % 16 < 8:                                                                       # bars 1-2: kick on 1 and 3, snare on 2 and 4
            which = "kick" if b % 2 == 0 else "snare"
            layers[which][i:i + hit_len] += level * hits[which]
        else:                                                                                # bars 3-4: a syncopated pattern
            if b % 4 in (0, 3):
                layers["kick"][i:i + hit_len] += level * hits["kick"]
            if b % 4 == 1:
                layers["snare"][i:i + hit_len] += level * hits["snare"]
            j = int(round((when + 0.5 * beat) * fs))
            if b % 4 == 1 and j < n:
                layers["kick"][j:j + hit_len] += 0.9 * level * hits["kick"]
            if b % 4 == 2 and j < n:
                layers["snare"][j:j + hit_len] += level * hits["snare"]
        if b % 16 == 15:                                                                     # the fill that closes the loop
            for q in (0.25, 0.5, 0.75):
                j = int(round((when + q * beat) * fs))
                if j < n:
                    layers["snare"][j:j + hit_len] += (0.5 + 0.5 * q) * level * hits["snare"]
        for off, a in ((0.0, 0.6), (0.5, 0.4)):
            j = int(round((when + off * beat) * fs))
            if j < n:
                layers["hat"][j:j + hit_len] += a * 10.0 ** (rs.uniform(-1.5, 1.5) / 20.0) * hits["hat"]
    # bass and pad follow the bar's chord: A minor, F, C, G (roots in Hz; triads as frequency ratios)
    roots = (55.0, 43.65, 65.41, 49.0)
    triads = ((1.0, 1.1892, 1.4983), (1.0, 1.2599, 1.4983), (1.0, 1.2599, 1.4983), (1.0, 1.2599, 1.4983))
    bar = np.minimum((t / (4 * beat)).astype(int) % 4, 3)
    f_root = np.asarray(roots)[bar]
    ph_root = 2 * np.pi * np.cumsum(f_root) / fs
    eighth = np.mod(t, beat / 2) / (beat / 2)
    bass = (np.sin(ph_root) + 0.5 * np.sin(2 * ph_root) + 0.2 * np.sin(3 * ph_root)) * np.exp(-3.0 * eighth) * (t >= 4 * beat)
    pads = []
    for side in (-1.0, 1.0):
        pad = np.zeros(n)
        for k in range(3):
            ratio = np.asarray([tr[k] for tr in triads])[bar] * 4.0 * (1.0 + 0.0015 * side * (k + 1))
            pad += np.sin(ratio * ph_root + 0.7 * k) / 3.0
        pads.append(pad * (0.75 + 0.25 * np.sin(2 * np.pi * 0.9 * t + side)))
    # the sung line: a note every half beat to two beats from a pentatonic walk, rests between phrases
    scale = 220.0 * 2.0 ** (np.array([0, 3, 5, 7, 10, 12, 15, 17, 19]) / 12.0)
    f_note = np.zeros(n)
    on = np.zeros(n)
    pos, degree = int(round(2.5 * beat * fs)), 4
    while pos < n:
        length = int(round(rs.choice([0.5, 1.0, 1.0, 1.5, 2.0]) * beat * fs))
        if rs.rand() < 0.22:                                                                 # a rest
            pos += length
            continue
        degree = int(np.clip(degree + rs.randint(-2, 3), 0, len(scale) - 1))
        end = min(n, pos + length)
        f_note[pos:end] = scale[degree]
        k = np.arange(end - pos) / float(fs)
        on[pos:end] = np.minimum(1.0, k / 0.03) * np.minimum(1.0, (k[-1] - k) / 0.05 + 0.02)
        pos = end
    f_voice = np.where(f_note > 0, f_note, 220.0) * (1.0 + 0.012 * np.sin(2 * np.pi * 5.6 * t + 0.4 * np.sin(2 * np.pi * 0.31 * t)))
    ph_v = 2 * np.pi * np.cumsum(f_voice) / fs
    voice = np.zeros(n)
    for h in range(1, 13):                                                                   # formants near 700 and 1 200 Hz
        fh = h * 440.0
        weight = np.exp(-((fh - 700.0) / 260.0) ** 2) + 0.6 * np.exp(-((fh - 1200.0) / 300.0) ** 2) + 0.08 / h
        voice += weight * np.sin(h * ph_v)
    voice *= on
    arrangement = np.ones(n)
    arrangement[t < 8 * beat] = 0.5                                                          # intro 6 dB down
    fade = max(1, int(round(1.5 * fs)))
    arrangement[n - min(fade, n):] *= np.linspace(1.0, 0.0, min(fade, n))
    out = np.empty((n, channels))
    for c in range(channels):
        side = 0.0 if channels == 1 else 2.0 * c / (channels - 1) - 1.0
        drums = (0.5 * layers["kick"][:n] + 0.3 * (1.0 + 0.2 * side) * layers["snare"][:n] + 0.18 * (1.0 - 0.4 * side) * layers["hat"][:n]
                 + 0.3 * (1.0 - 0.3 * side) * layers["crash"][:n])
        pad = pads[0] if side <= 0 else pads[1]
        mix = drums + 0.5 * bass + 0.26 * pad + 0.2 * (1.0 + 0.25 * side) * voice
        out[:, c] = arrangement * mix + 3e-4 * rs.standard_normal(n)
    out *= 0.89 / max(np.max(np.abs(out)), 1e-9)
    return np.round(out * 32768.0).clip(-32768, 32767) / 32768.0
