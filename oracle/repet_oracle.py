"""CPU ORACLE for the REPET hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the algorithm in the reference module ``repet.py`` (zafarrafii/
REPET-Python). Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this file, and only as the *checker* (or as the timed CPU baseline). The shipped path
(``repet-python_amd/``) never imports it and has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors of its own ("parity unpinned by the
reference"), so this oracle is pinned against outputs of the reference itself, generated in the build
container by ``tests/golden/make_golden.py`` (which imports the unmodified reference) and committed
as fixtures under ``tests/golden/``. ``tests/test_oracle_golden.py`` re-checks the oracle against
those fixtures wherever the tests run.

The stage decomposition mirrors the device pipeline (batched framing, frame-parallel ``simonline``,
independent segments for ``extended``) rather than the reference's loop structure; each function
cites the reference lines it restates (``repet.py:a-b``).
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.signal.windows

EPS = np.finfo(float).eps  # repet.py:1446,1504,1541,882 use np.finfo(float).eps = 2**-52


@dataclass
class Params:
    """The nine module-level parameters of the reference (repet.py:42-63), same defaults."""
    cutoff_frequency: float = 100
    period_range: tuple = (1, 10)
    segment_length: float = 10
    segment_step: float = 5
    filter_order: int = 5
    similarity_threshold: float = 0
    similarity_distance: float = 1
    similarity_number: int = 100
    buffer_length: float = 10


@dataclass
class Trace:
    """Integer/float intermediates captured for stage-level parity checks."""
    items: dict = field(default_factory=dict)

    def put(self, key, value):
        self.items[key] = value


# ----------------------------------------------------------------------------- derived sizes
def window_length_for(fs):
    """repet.py:130 -- power of two covering 40 ms."""
    return pow(2, int(np.ceil(np.log2(0.04 * fs))))


def stft_geometry(fs):
    """(W, window, H) as every public function derives them (repet.py:130-132)."""
    w = window_length_for(fs)
    return w, scipy.signal.windows.hamming(w, sym=False), int(w / 2)


def centred_frame_count(n, w, h):
    """repet.py:1018-1028."""
    pad = int(np.floor(w / 2))
    return int(np.ceil(((n + 2 * pad) - w) / h)) + 1


def period_range_frames(p, fs, h):
    """repet.py:165-167 (np.round: half to even)."""
    return np.round(np.array(p.period_range) * fs / h).astype(int)


def cutoff_bins(p, fs, w):
    """repet.py:173 (Python round: half to even)."""
    return round(p.cutoff_frequency * w / fs)


# ----------------------------------------------------------------------------- STFT / iSTFT
def stft(x, window, h):
    """Centred STFT of one channel, all W bins, (W, T) complex (repet.py:1001-1060)."""
    n = len(x)
    w = len(window)
    pad = int(np.floor(w / 2))
    t = centred_frame_count(n, w, h)
    padded = np.zeros(t * h + (w - h))
    padded[pad:pad + n] = x
    frames = np.lib.stride_tricks.sliding_window_view(padded, w)[::h][:t] * window
    return np.fft.fft(frames, axis=1).T


def istft(spec, window, h):
    """Overlap-add inverse of :func:`stft` (repet.py:1063-1105)."""
    w, t = spec.shape
    frames = np.real(np.fft.ifft(spec, axis=0))  # (W, T)
    y = np.zeros(t * h + (w - h))
    if w % h == 0:
        # frame j lands on samples [j*h, j*h+w): add it chunk by chunk of h samples; for a given
        # sample the chunks arrive in the same (frame-ascending) order as the reference's loop
        # only up to commutativity, which is exact for the two contributors of h = w/2.
        rows = y.reshape(-1, h)
        for q in range(w // h - 1, -1, -1):
            rows[q:q + t] += frames[q * h:(q + 1) * h].T
    else:
        for j in range(t):
            y[j * h:j * h + w] += frames[:, j]
    y = y[w - h:len(y) - (w - h)]
    return y / sum(window[0:w:h])


def spectrogram_channels(x, window, h):
    """STFT of every channel: (W, T, C) complex and its (F, T, C) magnitude (repet.py:149-158)."""
    w = len(window)
    spec = np.stack([stft(x[:, c], window, h) for c in range(x.shape[1])], axis=2)
    return spec, np.abs(spec[0:int(w / 2) + 1])


def resynthesize(mask, spec_c, window, h, n):
    """Mirror the (F, T) mask to W bins, apply, invert, truncate (repet.py:188-200)."""
    full = np.concatenate((mask, mask[-2:0:-1]), axis=0)
    return istft(full * spec_c, window, h)[0:n]


# ----------------------------------------------------------------------------- beat spectrum
def acorr(m):
    """Unbiased autocorrelation of each column by Wiener-Khinchin (repet.py:1108-1139)."""
    r = m.shape[0]
    psd = np.power(np.abs(np.fft.fft(m, n=2 * r, axis=0)), 2)
    ac = np.real(np.fft.ifft(psd, axis=0))[0:r]
    return ac / np.arange(r, 0, -1)[:, np.newaxis]


def beatspectrum(power_spec):
    """(F, T) -> (T,) mean over frequency of the per-bin autocorrelation (repet.py:1142-1158)."""
    return np.mean(acorr(power_spec.T), axis=1)


def beatspectrogram(power_spec, seg_len, seg_step):
    """Sliding beat spectrum with the reference's replicate-with-a-hole rule (repet.py:1161-1206)."""
    t = power_spec.shape[1]
    left = int(np.ceil((seg_len - 1) / 2))
    right = int(np.floor((seg_len - 1) / 2))
    padded = np.pad(power_spec, ((0, 0), (left, right)))
    out = np.zeros((seg_len, t))
    for i in range(0, t, seg_step):
        b = beatspectrum(padded[:, i:i + seg_len])
        out[:, i] = b
        # slice end is i+seg_step-1 (exclusive): column i+seg_step-1 keeps its zeros
        out[:, i:min(i + seg_step - 1, t)] = b[:, np.newaxis]
    return out


def periods(beat, prange):
    """arg-max lag + 1 + p0, scalar or per column (repet.py:1249-1291)."""
    hi = min(prange[1], int(np.floor(beat.shape[0] / 3)))
    return np.argmax(beat[prange[0]:hi], axis=0) + 1 + prange[0]


# ----------------------------------------------------------------------------- similarity
def selfsimilaritymatrix(m):
    """Cosine self-similarity of the columns (repet.py:1209-1225)."""
    m = m / np.sqrt(np.sum(np.power(m, 2), axis=0))
    return np.matmul(m.T, m)


def similaritymatrix(a, b):
    """Cosine similarity between two column sets (repet.py:1228-1246)."""
    a = a / np.sqrt(np.sum(np.power(a, 2), axis=0))
    b = b / np.sqrt(np.sum(np.power(b, 2), axis=0))
    return np.matmul(a.T, b)


def _trailing_max(v, d):
    """out[..., i] = max(v[..., max(i-d,0):i]) (NaN-propagating; -inf where the window is empty)."""
    n = v.shape[-1]
    out = np.full(v.shape, -np.inf)
    if d <= 0:
        return out
    # doubling: span[k][..., i] = max(v[..., i-k:i]) for k a power of two, then stitch d from bits
    span = np.full(v.shape, -np.inf)
    span[..., 1:] = v[..., :-1]
    k = 1
    covered = 0
    remaining = d
    while True:
        if remaining & k:
            # out currently covers v[i-covered:i]; extend with the k samples before that
            shifted = np.full(v.shape, -np.inf)
            if covered < n:
                shifted[..., covered:] = span[..., :n - covered]
            out = np.maximum(out, shifted)
            covered += k
            remaining -= k
        if remaining == 0:
            break
        nxt = np.full(v.shape, -np.inf)
        nxt[..., k:] = span[..., :n - k] if k < n else nxt[..., k:]
        span = np.maximum(span, nxt)
        k *= 2
    return out


def localmaxima_mask(v, min_value, d):
    """Boolean strict-local-maximum test of repet.py:1315-1329 along the last axis."""
    left = _trailing_max(v, d)
    right = _trailing_max(v[..., ::-1], d)[..., ::-1]
    with np.errstate(invalid="ignore"):
        return (v >= min_value) & (v > left) & (v > right)


def _top_indices(v, keep, number_values):
    """Candidates sorted by value descending, first ``number_values`` kept (repet.py:1331-1343)."""
    cand = np.flatnonzero(keep)
    vals = v[cand]
    order = np.argsort(vals)[::-1][:min(number_values, len(vals))]
    return vals[order], cand[order]


def localmaxima(v, min_value, d, number_values):
    """(values, indices) of the peaks of a vector (repet.py:1294-1345)."""
    v = np.asarray(v, dtype=float)
    return _top_indices(v, localmaxima_mask(v, min_value, d), number_values)


def indices(sim, threshold, d, number):
    """Similar-frame index list of every frame: column i of ``sim`` is scanned (repet.py:1348-1383)."""
    out = []
    for lo in range(0, sim.shape[0], 64):        # 64 columns at a time keeps the scan in cache
        by_col = np.ascontiguousarray(sim[:, lo:lo + 64].T)
        keep = localmaxima_mask(by_col, threshold, d)
        out.extend(_top_indices(by_col[i], keep[i], number)[1] for i in range(by_col.shape[0]))
    return out


# ----------------------------------------------------------------------------- masks
def soft_mask(v, model):
    """min, then (W+eps)/(V+eps) (repet.py:1441-1448)."""
    return (np.minimum(v, model) + EPS) / (v + EPS)


def mask(v, period):
    """Period-median repeating mask (repet.py:1386-1458)."""
    f, t = v.shape
    s = int(np.ceil(t / period))
    padded = np.zeros((f, s * period))
    padded[:, :t] = v
    cube = padded.reshape(f, s, period)      # cube[f, seg, q] = V[f, seg*period + q]
    full = t - (s - 1) * period             # q < full: all s segments hold real data
    model = np.empty((f, period))
    model[:, :full] = np.median(cube[:, :, :full], axis=1)
    model[:, full:] = np.median(cube[:, :s - 1, full:], axis=1)
    m = soft_mask(cube, model[:, np.newaxis, :])
    return m.reshape(f, s * period)[:, :t]


def adaptivemask(v, per, order):
    """Local-period median mask (repet.py:1461-1508)."""
    f, t = v.shape
    taps = np.arange(1, order + 1) - int(np.ceil(order / 2))
    model = np.zeros((f, t))
    for i in range(t):
        idx = i + taps * per[i]
        idx = idx[(idx >= 0) & (idx < t)]
        model[:, i] = np.median(v[:, idx], axis=1)
    return soft_mask(v, model)


def simmask(v, sim_indices, chunk=192):
    """Similarity-median mask (repet.py:1511-1545); frames with equal list length are batched."""
    f, t = v.shape
    by_frame = np.ascontiguousarray(v.T)     # (T, F): a similar frame is one contiguous row
    model = np.empty((t, f))
    counts = np.array([len(ix) for ix in sim_indices])
    for k in np.unique(counts):
        frames = np.flatnonzero(counts == k)
        if k == 0:
            model[frames] = np.nan           # np.median of an empty slice
            continue
        for lo in range(0, len(frames), chunk):
            sel = frames[lo:lo + chunk]
            gather = by_frame[np.stack([sim_indices[i] for i in sel])]   # (nb, k, F)
            model[sel] = np.median(gather, axis=1)
    return soft_mask(v, model.T)


# ----------------------------------------------------------------------------- public variants
def _finish(masks_fn, spec, mag, window, h, n, cut, trace=None):
    out = np.zeros((n, mag.shape[2]))
    for c in range(mag.shape[2]):
        m = masks_fn(mag[:, :, c])
        m[1:cut + 1, :] = 1                  # repet.py:185
        if trace is not None and c == 0:
            trace.put("mask_c0", m.copy())
        out[:, c] = resynthesize(m, spec[:, :, c], window, h, n)
    return out


def original(x, fs, p=None, trace=None):
    """repet.py:67-202."""
    p = p or Params()
    n, _ = np.shape(x)
    w, window, h = stft_geometry(fs)
    spec, mag = spectrogram_channels(x, window, h)
    beat = beatspectrum(np.power(np.mean(mag, axis=2), 2))
    prange = period_range_frames(p, fs, h)
    period = periods(beat, prange)
    cut = cutoff_bins(p, fs, w)
    if trace is not None:
        trace.put("beat_spectrum", beat)
        trace.put("repeating_period", int(period))
    return _finish(lambda v: mask(v, period), spec, mag, window, h, n, cut, trace)


def extended_plan(n, fs, p):
    """Segment starts/lengths and overlap of repet.py:266-281,306-322."""
    seg_len = round(p.segment_length * fs)
    seg_step = round(p.segment_step * fs)
    if n < seg_len + seg_step:
        return [(0, n)], 0
    count = 1 + int(np.floor((n - seg_len) / seg_step))
    segs = [(j * seg_step, seg_len) for j in range(count - 1)]
    segs.append(((count - 1) * seg_step, n - (count - 1) * seg_step))
    return segs, seg_len - seg_step


def extended(x, fs, p=None, trace=None):
    """repet.py:205-419: ``original`` per segment, triangular cross-fade of the overlaps."""
    p = p or Params()
    n, c = np.shape(x)
    segs, overlap = extended_plan(n, fs, p)
    if len(segs) == 1:
        return original(x, fs, p, trace)
    tri = scipy.signal.windows.triang(2 * overlap)
    out = np.zeros((n, c))
    per_seg = []
    for j, (start, length) in enumerate(segs):
        tr = Trace()
        piece = original(x[start:start + length], fs, p, tr)
        per_seg.append(tr.items["repeating_period"])
        if j > 0:
            out[start:start + overlap] *= tri[overlap:, np.newaxis]
            piece[:overlap] *= tri[:overlap, np.newaxis]
        out[start:start + length] += piece
    if trace is not None:
        trace.put("segment_periods", np.array(per_seg))
    return out


def segment_weights(j, segs, overlap):
    """Weight every sample of segment j ends up with after the in-place cross-fade of repet.py:380-414: its own
    rising half of triang(2*overlap) (j > 0) times the falling half that EVERY later segment applies to what lies
    under its first ``overlap`` samples (one factor at the default 50 % overlap, several when step < overlap)."""
    start, length = segs[j]
    w = np.ones(length)
    tri = scipy.signal.windows.triang(2 * overlap)
    if j > 0:
        w[:overlap] *= tri[:overlap]
    for later_start, _ in segs[j + 1:]:
        lo = later_start - start
        hi = min(lo + overlap, length)
        if lo < hi:
            w[lo:hi] *= tri[overlap:overlap + hi - lo]
    return w


def extended_range(x, fs, first, count, p=None):
    """Contribution of segments [first, first+count) to :func:`extended`: sum_j w_j * original(segment_j) with
    w_j = :func:`segment_weights` (the reference's cross-fade is linear in the segments). Used to check the
    multi-GPU segment sharding."""
    p = p or Params()
    n, c = np.shape(x)
    segs, overlap = extended_plan(n, fs, p)
    out = np.zeros((n, c))
    if len(segs) == 1:
        return original(x, fs, p) if (first, count) == (0, 1) else out
    for j in range(first, first + count):
        start, length = segs[j]
        piece = original(x[start:start + length], fs, p)
        out[start:start + length] += piece * segment_weights(j, segs, overlap)[:, np.newaxis]
    return out


def adaptive(x, fs, p=None, trace=None):
    """repet.py:422-568."""
    p = p or Params()
    n, _ = np.shape(x)
    w, window, h = stft_geometry(fs)
    spec, mag = spectrogram_channels(x, window, h)
    seg_len = int(round(p.segment_length * fs / h))
    seg_step = int(round(p.segment_step * fs / h))
    bsg = beatspectrogram(np.power(np.mean(mag, axis=2), 2), seg_len, seg_step)
    per = periods(bsg, period_range_frames(p, fs, h))
    cut = cutoff_bins(p, fs, w)
    if trace is not None:
        trace.put("beat_spectrogram", bsg)
        trace.put("repeating_periods", per)
    return _finish(lambda v: adaptivemask(v, per, p.filter_order), spec, mag, window, h, n, cut, trace)


def sim(x, fs, p=None, trace=None, override_indices=None):
    """repet.py:571-709. ``override_indices`` (tests only) replaces the peak-picking result, to separate
    discrete near-tie decisions from numerical error when checking an fp32 implementation."""
    p = p or Params()
    n, _ = np.shape(x)
    w, window, h = stft_geometry(fs)
    spec, mag = spectrogram_channels(x, window, h)
    s = selfsimilaritymatrix(np.mean(mag, axis=2))
    dist = int(round(p.similarity_distance * fs / h))
    idx = indices(s, p.similarity_threshold, dist, p.similarity_number)
    if override_indices is not None:
        idx = [np.asarray(ix, dtype=int) for ix in override_indices]
    cut = cutoff_bins(p, fs, w)
    if trace is not None:
        trace.put("similarity_matrix", s)
        trace.put("similarity_indices", idx)
    return _finish(lambda v: simmask(v, idx), spec, mag, window, h, n, cut, trace)


def online_frame_count(n, w, h):
    """repet.py:781."""
    return int(np.ceil((n - w) / h + 1))


def simonline(x, fs, p=None, trace=None, override_indices=None):
    """repet.py:712-911, evaluated frame-parallel.

    The circular buffer at step j holds frames j-B+1..j; buffer column c holds frame
    ``j - ((j - c) mod B)``. Peak picking runs in buffer-column order (repet.py:837,861).
    """
    p = p or Params()
    n, ch = np.shape(x)
    w, window, h = stft_geometry(fs)
    f = int(w / 2 + 1)
    t = online_frame_count(n, w, h)
    b = round((p.buffer_length * fs) / h)
    if n < (b - 2) * h + w:
        # the warm-up loop slices b-1 whole frames out of the unpadded signal (repet.py:795-810)
        raise ValueError("operands could not be broadcast together: signal shorter than the buffer")
    total = (t - 1) * h + w
    padded = np.zeros((total, ch))
    padded[:n] = x
    dist = int(round(p.similarity_distance * fs / h))
    cut = cutoff_bins(p, fs, w)

    frames = np.stack([np.lib.stride_tricks.sliding_window_view(padded[:, c], w)[::h][:t] * window
                       for c in range(ch)], axis=0)           # (C, T, W)
    spec = np.fft.fft(frames, axis=2)                         # (C, T, W)
    mag = np.abs(spec[:, :, :f])                              # (C, T, F)
    mean_mag = np.mean(np.moveaxis(mag, 0, 2), axis=2)        # (T, F)
    unit = mean_mag / np.sqrt(np.sum(np.power(mean_mag, 2), axis=1))[:, np.newaxis]

    out = np.zeros((total, ch))
    cols = np.arange(b)
    all_idx = []
    for j in range(b - 1, t):
        in_col = j - np.mod(j - cols, b)                      # frame held by each buffer column
        simvec = unit[in_col] @ unit[j]
        _, peaks = localmaxima(simvec, p.similarity_threshold, dist, p.similarity_number)
        similar = in_col[peaks]
        if override_indices is not None:
            similar = np.asarray(override_indices[j - (b - 1)], dtype=int)
        if trace is not None:
            trace.items.setdefault("similarity_vectors", []).append((in_col, simvec))
        all_idx.append(similar)
        for c in range(ch):
            cur = mag[c, j]
            model = np.median(mag[c, similar], axis=0) if len(similar) else np.full(f, np.nan)
            m = (np.minimum(model, cur) + EPS) / (cur + EPS)
            m[1:cut + 1] = 1
            full = np.concatenate((m, m[-2:0:-1]))
            out[j * h:j * h + w, c] += np.real(np.fft.ifft(full * spec[c, j]))
    if trace is not None:
        trace.put("similarity_indices", all_idx)
        trace.put("buffer_frames", b)
    return out[0:n] / sum(window[0:w:h])


ALGORITHMS = {"original": original, "extended": extended, "adaptive": adaptive,
              "sim": sim, "simonline": simonline}
