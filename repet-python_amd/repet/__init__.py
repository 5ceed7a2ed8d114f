"""repet -- MI355X-native REPET (REpeating Pattern Extraction Technique), drop-in for zafarrafii/REPET-Python.

Same call surface as the reference module ``repet.py``::

    background_signal = repet.original(audio_signal, sampling_frequency)    # repet.py:67
    background_signal = repet.extended(audio_signal, sampling_frequency)    # repet.py:205
    background_signal = repet.adaptive(audio_signal, sampling_frequency)    # repet.py:422
    background_signal = repet.sim(audio_signal, sampling_frequency)         # repet.py:571
    background_signal = repet.simonline(audio_signal, sampling_frequency)   # repet.py:712

``audio_signal`` is ``(number_samples, number_channels)``; the result is a fresh float64 array of the
same shape. The nine module-level parameters below have the reference's names and defaults
(repet.py:42-63) and are read at call time, so ``repet.period_range = [1, 5]`` before a call behaves as
it does there. All arithmetic runs in hand-written HIP kernels for gfx950 behind ``librepet_hip.so``
(``include/repet_hip.h``); this module only validates, derives the integer sizes with the reference's
own rounding rules, and crosses the C ABI. There is no NumPy/CPU implementation of the separation
path here: without the library or a GPU the calls raise.
"""
import numpy as np

from . import _native
from ._native import Context  # noqa: F401  (device-resident API used by bench.py)
from ._native import OnlineSeparator as _OnlineSeparator

# ---- public parameters (repet.py:42-63) ----------------------------------------------------------------
cutoff_frequency = 100
period_range = [1, 10]
segment_length = 10
segment_step = 5
filter_order = 5
similarity_threshold = 0
similarity_distance = 1
similarity_number = 100
buffer_length = 10

# Not a parameter of the reference: what to do with samples that are NOT FINITE. repet.py computes on (a NaN sample makes the
# frames that hold it NaN; repet.py:125 has no input check). True (default since round 6: the reference's behaviour): the
# samples are let through and every variant returns what the reference returns for NaN samples (``sim`` / ``simonline``: NaN
# on the samples of the affected frames only; the period family: also at the same position of every period, and the period
# ``period_range[0] + 1``); an infinite sample is treated as NaN. False: such input raises ValueError
# (REPET_FLAG_REFUSE_NONFINITE). Read at call time like the others.
strict_reference = True

_device = 0  # HIP device used by the one-shot calls


def set_device(index):
    """Select the HIP device for subsequent calls (the reference has no such notion)."""
    global _device
    _device = int(index)


def device_host_cpus(device=None):
    """CPUs of the NUMA node the device hangs off, among those this process may use ([] when unknown or nothing to choose)."""
    import ctypes as C
    dev = _device if device is None else int(device)
    n = C.c_int32()
    buf = (C.c_int32 * 4096)()
    _native.check(_native.lib().repet_device_host_cpus(dev, buf, 4096, C.byref(n)))
    return [int(buf[i]) for i in range(min(n.value, 4096))]


def bind_host_to_device(device=None):
    """Bind the calling process (thread) to the CPUs of the device's NUMA node -- what ``numactl --cpunodebind`` would do for a
    GPU job. Arrays first touched afterwards live there too. Returns the previous affinity mask (``os.sched_setaffinity(0,
    previous)`` undoes it), or None when there is nothing to choose. Not done implicitly: the affinity of the caller's
    threads is the caller's business (the library pins only its own conversion threads)."""
    import os
    cpus = device_host_cpus(device)
    if not cpus:
        return None
    previous = os.sched_getaffinity(0)
    os.sched_setaffinity(0, cpus)
    return previous


# ---- sizes derived exactly as the reference derives them ---------------------------------------------------
def _window_length(sampling_frequency):
    return pow(2, int(np.ceil(np.log2(0.04 * sampling_frequency))))  # repet.py:130


def derive_params(sampling_frequency):
    """Snapshot the module parameters into the integer ``repet_params`` of the C ABI.

    Python ``round`` and ``np.round`` are half-to-even; every expression below is the one the reference
    evaluates (line cited), so e.g. an 8 kHz clip gets a 312-frame online buffer (``round(312.5)``).
    """
    fs = sampling_frequency
    w = _window_length(fs)
    h = int(w / 2)                                                            # repet.py:132
    pr = np.round(np.array(period_range) * fs / h).astype(int)                # repet.py:165
    p = _native.Params()
    p.window_length = w
    p.step_length = h
    p.period_lo = int(pr[0])
    p.period_hi = int(pr[1])
    p.cutoff_bins = int(round(cutoff_frequency * w / fs))                     # repet.py:173
    p.filter_order = int(filter_order)
    p.seg_len_frames = int(round(segment_length * fs / h))                    # repet.py:519
    p.seg_step_frames = int(round(segment_step * fs / h))                     # repet.py:520
    p.sim_distance_frames = int(round(similarity_distance * fs / h))          # repet.py:670
    p.sim_number = int(similarity_number)
    p.buffer_frames = int(round((buffer_length * fs) / h))                    # repet.py:787
    p.seg_len_samples = int(round(segment_length * fs))                       # repet.py:266
    p.seg_step_samples = int(round(segment_step * fs))                        # repet.py:267
    p.sim_threshold = float(similarity_threshold)
    p.flags = 0 if strict_reference else _native.FLAG_REFUSE_NONFINITE
    return p


def release_workspaces():
    """Free the device workspaces the one-shot calls of THIS thread keep between calls (the reference has no such
    notion: its arrays die with the call). Worth calling after a one-off long ``sim``: the similarity matrix of a
    10-minute clip is several GB of HBM."""
    _native.check(_native.lib().repet_release_thread_ctx())


def _separate(algo, audio_signal, sampling_frequency):
    number_samples, number_channels = np.shape(audio_signal)   # 1-D input: ValueError, like repet.py:125
    params = derive_params(sampling_frequency)
    signal, code = _native.as_input(audio_signal)
    background_signal = _native.result_array((number_samples, number_channels))
    lib = _native.lib()
    if lib.repet_device_count() < 1:
        raise RuntimeError("no HIP device visible: the REPET engine has no CPU fallback")
    _native.check(lib.repet_run(_native.ALGO_IDS[algo], _native.ptr(signal), code, number_samples,
                                number_channels, params, _native.ptr(background_signal), _device, None))
    return background_signal


def original(audio_signal, sampling_frequency):
    """Original REPET: one repeating period, period-median model (repet.py:67-202)."""
    return _separate("original", audio_signal, sampling_frequency)


def extended(audio_signal, sampling_frequency):
    """REPET extended: ``original`` on 10-s segments with triangular cross-fades (repet.py:205-419)."""
    return _separate("extended", audio_signal, sampling_frequency)


def adaptive(audio_signal, sampling_frequency):
    """Adaptive REPET: time-varying period from a beat spectrogram (repet.py:422-568)."""
    return _separate("adaptive", audio_signal, sampling_frequency)


def sim(audio_signal, sampling_frequency):
    """REPET-SIM: repeating frames found through the cosine self-similarity matrix (repet.py:571-709)."""
    return _separate("sim", audio_signal, sampling_frequency)


def simonline(audio_signal, sampling_frequency):
    """Online REPET-SIM over a circular buffer of past frames (repet.py:712-911)."""
    return _separate("simonline", audio_signal, sampling_frequency)


def online(sampling_frequency, number_channels):
    """Streaming form of :func:`simonline` (the reference needs the whole signal up front): returns an object with
    ``push(chunk) -> newly final background samples`` and ``finish() -> the remaining ones``. The module parameters are
    snapshotted now; the concatenated output equals ``simonline`` of the concatenated input."""
    return _OnlineSeparator(derive_params(sampling_frequency), number_channels, _device)


def run_batch(algo, audio_signals, sampling_frequency, n_devices=1, transport="host", device=None, depth=None):
    """Separate a list of independent clips, dealt longest-first over ``n_devices`` GPUs of this process. ``transport``:
    "host" -- every device moves its own clips over its own PCIe link; "rccl" -- the clips enter through device 0 and
    travel to their devices (and the results back) as grouped ncclSend / ncclRecv over xGMI. ``device`` / ``depth``: the clips
    one after another through that ONE device with ``depth`` (default 2) of them in flight -- upload, kernels and download of
    neighbouring clips side by side (``repet_run_stream``); every result is what the one-shot call returns, bit for bit."""
    import ctypes as C
    params = derive_params(sampling_frequency)
    ins, outs, ns, cs, code = [], [], [], [], None
    for a in audio_signals:
        n, c = np.shape(a)
        arr, k = _native.as_input(a)
        if code is None:
            code = k
        elif k != code:
            arr, k = np.ascontiguousarray(arr, dtype=np.float64), _native.F64
            if code != _native.F64:
                ins = [np.ascontiguousarray(x, dtype=np.float64) for x in ins]
                code = _native.F64
        ins.append(arr)
        outs.append(np.empty((n, c), dtype=np.float64))
        ns.append(n)
        cs.append(c)
    count = len(ins)
    in_ptrs = (C.c_void_p * count)(*[x.ctypes.data for x in ins])
    out_ptrs = (C.c_void_p * count)(*[x.ctypes.data for x in outs])
    if device is not None or depth is not None:
        _native.check(_native.lib().repet_run_stream(
            _native.ALGO_IDS[algo], count, in_ptrs, code if code is not None else _native.F64,
            (C.c_int64 * count)(*ns), (C.c_int32 * count)(*cs), params, out_ptrs, int(_device if device is None else device),
            int(2 if depth is None else depth)))
        return outs
    entry = _native.lib().repet_run_batch_rccl if transport == "rccl" else _native.lib().repet_run_batch
    _native.check(entry(
        _native.ALGO_IDS[algo], count, in_ptrs, code if code is not None else _native.F64,
        (C.c_int64 * count)(*ns), (C.c_int32 * count)(*cs), params, out_ptrs, int(n_devices)))
    return outs


def last_batch_info():
    """What this thread's last ``run_batch`` did: transport, clips that went through send / receive, clips whose fp32
    remainder plane was resident when they were separated, RCCL groups completed."""
    import ctypes as C
    out = (C.c_int64 * 4)()
    _native.check(_native.lib().repet_last_batch_info(out))
    return {"transport": "rccl" if out[0] == 1 else "host", "clips_sent": int(out[1]), "clips_with_remainders": int(out[2]), "rccl_groups": int(out[3])}


# ---- private helpers of the reference, kept callable (README.md:79 uses repet._stft) -----------------------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _stft(audio_signal, window_function, step_length):
    """STFT of one channel, ``(window_length, number_frames)`` complex with all bins (repet.py:1001-1060)."""
    x = _f32(audio_signal)
    window = _f32(window_function)
    w = len(window)
    lib = _native.lib()
    t = lib.repet_frame_count(len(x), w, int(step_length), 1)
    f = w // 2 + 1
    spec = np.empty((t, f, 2), dtype=np.float32)
    _native.check(lib.repet_stft(_native.default_context(_device).handle, _native.ptr(x), len(x),
                                 _native.ptr(window), w, int(step_length), 1, _native.ptr(spec), t))
    half = (spec[..., 0] + 1j * spec[..., 1]).astype(complex).T          # (F, T)
    return np.concatenate((half, np.conj(half[-2:0:-1])), axis=0)


def _istft(audio_stft, window_function, step_length):
    """Inverse STFT by overlap-add (repet.py:1063-1105); the mirrored bins are implied by the first half."""
    window = _f32(window_function)
    w, t = np.shape(audio_stft)
    f = w // 2 + 1
    half = np.ascontiguousarray(np.asarray(audio_stft)[:f].T)
    spec = np.empty((t, f, 2), dtype=np.float32)
    spec[..., 0] = half.real
    spec[..., 1] = half.imag
    n_out = t * int(step_length) - (w - int(step_length))
    y = np.empty(n_out, dtype=np.float32)
    _native.check(_native.lib().repet_istft(_native.default_context(_device).handle, _native.ptr(spec), t,
                                            _native.ptr(window), w, int(step_length), _native.ptr(y), n_out))
    return y.astype(np.float64)


def _selfsimilaritymatrix(data_matrix):
    """Cosine self-similarity between the columns (repet.py:1209-1225)."""
    rows = _f32(np.asarray(data_matrix).T)
    t, f = rows.shape
    s = np.empty((t, t), dtype=np.float32)
    _native.check(_native.lib().repet_selfsim(_native.default_context(_device).handle, _native.ptr(rows), t, f,
                                              _native.ptr(s)))
    return s.astype(np.float64)


def _selfsimilarity_records(data_matrix):
    """(similarity matrix, largest, second largest, offset of the largest) per row and aligned run of 32 columns: the segment
    records the peak picking of ``sim`` works from (``repet_selfsim_records``)."""
    rows = _f32(np.asarray(data_matrix).T)
    t, f = rows.shape
    n_seg = -(-t // 32)
    s = np.empty((t, t), dtype=np.float32)
    top, second, at = (np.empty((t, n_seg), dtype=np.float32), np.empty((t, n_seg), dtype=np.float32), np.empty((t, n_seg), dtype=np.int32))
    _native.check(_native.lib().repet_selfsim_records(_native.default_context(_device).handle, _native.ptr(rows), t, f, _native.ptr(s),
                                                      _native.ptr(top), _native.ptr(second), _native.ptr(at)))
    return s, top, second, at


def _similaritymatrix(data_matrix1, data_matrix2):
    """Cosine similarity between the columns of two matrices (repet.py:1228-1246)."""
    a = _f32(np.asarray(data_matrix1).T)
    b = _f32(np.asarray(data_matrix2).T)
    out = np.empty((a.shape[0], b.shape[0]), dtype=np.float32)
    _native.check(_native.lib().repet_similarity(_native.default_context(_device).handle, _native.ptr(a), a.shape[0],
                                                 _native.ptr(b), b.shape[0], a.shape[1], _native.ptr(out)))
    return out.astype(np.float64)


def _acorr(data_matrix):
    """Unbiased autocorrelation of every column (repet.py:1108-1139)."""
    x = _f32(data_matrix)
    out = np.empty_like(x)
    _native.check(_native.lib().repet_acorr(_native.default_context(_device).handle, _native.ptr(x), x.shape[0],
                                            x.shape[1], _native.ptr(out)))
    return out.astype(np.float64)


def _beatspectrum(audio_spectrogram):
    """Beat spectrum of an (already squared) spectrogram (repet.py:1142-1158)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    beat = np.empty(t, dtype=np.float32)
    _native.check(_native.lib().repet_beat_spectrum(_native.default_context(_device).handle, _native.ptr(rows),
                                                    t, f, _native.ptr(beat), t))
    return beat.astype(np.float64)


def _beatspectrogram(audio_spectrogram, segment_length, segment_step):
    """Sliding beat spectrum, ``(segment_length, number_times)`` (repet.py:1161-1206)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    out = np.empty((t, int(segment_length)), dtype=np.float32)
    _native.check(_native.lib().repet_beat_spectrogram(_native.default_context(_device).handle,
                                                       _native.ptr(rows), t, f, int(segment_length),
                                                       int(segment_step), _native.ptr(out)))
    return out.T.astype(np.float64)


def _periods(beat_spectrogram, period_range):
    """Repeating period(s): arg-max lag + 1 + period_range[0] (repet.py:1249-1291)."""
    b = np.asarray(beat_spectrogram)
    cols = _f32(b[np.newaxis, :] if b.ndim == 1 else b.T)
    n_cols, n_lags = cols.shape
    out = np.empty(n_cols, dtype=np.int32)
    _native.check(_native.lib().repet_periods(_native.default_context(_device).handle, _native.ptr(cols),
                                              n_cols, n_lags, int(period_range[0]), int(period_range[1]),
                                              _native.ptr(out)))
    return int(out[0]) if b.ndim == 1 else out.astype(int)


def _local_maxima_rows(rows, minimum_value, minimum_distance, number_values):
    rows = _f32(rows)
    n_rows, n_cols = rows.shape
    idx = np.empty((n_rows, int(number_values)), dtype=np.int32)
    cnt = np.empty(n_rows, dtype=np.int32)
    _native.check(_native.lib().repet_local_maxima(_native.default_context(_device).handle, _native.ptr(rows),
                                                   n_rows, n_cols, float(minimum_value), int(minimum_distance),
                                                   int(number_values), _native.ptr(idx), _native.ptr(cnt)))
    return idx, cnt


def _localmaxima(data_vector, minimum_value, minimum_distance, number_values):
    """Values and indices of the top local maxima of a vector (repet.py:1294-1345)."""
    v = np.asarray(data_vector, dtype=float)
    idx, cnt = _local_maxima_rows(v[np.newaxis, :], minimum_value, minimum_distance, number_values)
    keep = idx[0, :cnt[0]].astype(int)
    return v[keep], keep


def _indices(similarity_matrix, similarity_threshold, similarity_distance, similarity_number):
    """Similar-frame indices of every frame: column i of the matrix is scanned (repet.py:1348-1383)."""
    idx, cnt = _local_maxima_rows(np.asarray(similarity_matrix).T, similarity_threshold, similarity_distance,
                                  similarity_number)
    return [idx[i, :cnt[i]].astype(int) for i in range(len(cnt))]


def _mask(audio_spectrogram, repeating_period):
    """Period-median repeating mask (repet.py:1386-1458)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    out = np.empty((t, f), dtype=np.float32)
    _native.check(_native.lib().repet_mask_period(_native.default_context(_device).handle, _native.ptr(rows), t,
                                                  f, int(repeating_period), _native.ptr(out)))
    return out.T.astype(np.float64)


def _adaptivemask(audio_spectrogram, repeating_periods, filter_order):
    """Local-period median mask (repet.py:1461-1508)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    per = np.ascontiguousarray(repeating_periods, dtype=np.int32)
    out = np.empty((t, f), dtype=np.float32)
    _native.check(_native.lib().repet_mask_adaptive(_native.default_context(_device).handle, _native.ptr(rows),
                                                    t, f, _native.ptr(per), int(filter_order), _native.ptr(out)))
    return out.T.astype(np.float64)


def _simmask(audio_spectrogram, similarity_indices):
    """Similarity-median mask from per-frame index lists (repet.py:1511-1545)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    width = max(1, max((len(ix) for ix in similarity_indices), default=1))
    idx = np.full((t, width), -1, dtype=np.int32)
    cnt = np.zeros(t, dtype=np.int32)
    for i, ix in enumerate(similarity_indices):
        idx[i, :len(ix)] = ix
        cnt[i] = len(ix)
    out = np.empty((t, f), dtype=np.float32)
    _native.check(_native.lib().repet_mask_sim(_native.default_context(_device).handle, _native.ptr(rows), t, f,
                                               _native.ptr(idx), _native.ptr(cnt), width, _native.ptr(out)))
    return out.T.astype(np.float64)


def _simmask_ranked(audio_spectrogram, similarity_indices, path="bits", want_codes=False):
    """``_simmask`` the way ``sim`` computes it on clips of more than 1 024 frames: through the rank transform of every bin
    (``_rank_columns``) and the packed 16-bit selection network (``path="rank"``) or the bit-sliced selection (``"bits"``).
    ``(F, T)`` with ``F - 1`` a multiple of 128 (a power of two for ``"bits"``); ``want_codes``: also the ``uint32 (F - 1, T)``
    words the bit-sliced selection leaves (lower median's rank | flag << 15 | upper median's rank << 16)."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    width = max(2, max((len(ix) for ix in similarity_indices), default=2))
    idx = np.full((t, width), -1, dtype=np.int32)
    cnt = np.zeros(t, dtype=np.int32)
    for i, ix in enumerate(similarity_indices):
        idx[i, :len(ix)] = ix
        cnt[i] = len(ix)
    out = np.empty((t, f), dtype=np.float32)
    codes = np.empty((t, f - 1), dtype=np.uint32) if want_codes else None
    _native.check(_native.lib().repet_mask_sim_ranked(_native.default_context(_device).handle, _native.ptr(rows), t, f,
                                                      _native.ptr(idx), _native.ptr(cnt), width, {"rank": 1, "bits": 2}[path],
                                                      _native.ptr(out), _native.ptr(codes) if want_codes else None))
    return (out.T.astype(np.float64), codes.T) if want_codes else out.T.astype(np.float64)


def _rank_columns(audio_spectrogram):
    """Rank transform behind the median of ``sim`` (no counterpart in the reference: np.median at repet.py:1535 only
    needs the order of a bin's magnitudes). ``(F, T)`` -> (codes ``(n, T)`` uint16 = 0x0400 + number of strictly smaller
    magnitudes of the bin, sorted ``(n, T)``) for the first ``n = F // 128 * 128`` bins."""
    rows = _f32(np.asarray(audio_spectrogram).T)
    t, f = rows.shape
    n = f // 128 * 128
    codes = np.empty((t, n), dtype=np.uint16)
    ordered = np.empty((n, t), dtype=np.float32)
    _native.check(_native.lib().repet_rank_columns(_native.default_context(_device).handle, _native.ptr(rows), t, f,
                                                   _native.ptr(codes), _native.ptr(ordered)))
    return codes.T, ordered


# ---- file / display utilities of the reference (host side, off the hot path) --------------------------------
def wavread(audio_file):
    """Read a WAVE file, integers scaled to [-1, 1) by their bit depth (repet.py:914-931): the array SciPy would return,
    divided by ``2 ** (8 * itemsize - 1)`` -- 24-bit PCM counts as int32 (sample in the top three bytes), 8-bit PCM is
    unsigned, and float files are divided too (a quirk of the reference, kept). The header is parsed by the library
    (``repet_wav_parse``); formats it does not handle go through ``scipy.io.wavfile`` like the reference."""
    image = np.fromfile(audio_file, dtype=np.uint8)
    info = _native.WavInfo()
    lib = _native.lib()
    if lib.repet_wav_parse(_native.ptr(image), image.size, info) != 0:
        import scipy.io.wavfile
        sampling_frequency, samples = scipy.io.wavfile.read(audio_file)
        return samples / pow(2, samples.itemsize * 8 - 1), sampling_frequency
    count = info.n_samples * info.n_channels
    width = info.bytes_per_sample
    raw = image[info.data_offset:info.data_offset + count * width]
    if info.format == 3:
        samples = raw.view("<f4" if width == 4 else "<f8")
    elif width == 1:
        samples = raw
    elif width == 3:                                          # packed 24-bit -> int32 with the sample in the top bytes
        wide = np.zeros((count, 4), dtype=np.uint8)
        wide[:, 1:] = raw.reshape(count, 3)
        samples = wide.view("<i4").ravel()
    else:
        samples = raw.view("<i2" if width == 2 else "<i4")
    samples = samples.reshape(info.n_samples, info.n_channels) if info.n_channels > 1 else samples.reshape(info.n_samples)
    return samples / pow(2, samples.itemsize * 8 - 1), int(info.sampling_frequency)


def wavwrite(audio_signal, sampling_frequency, audio_file):
    """Write a WAVE file with the dtype as given (repet.py:934-946): byte for byte what ``scipy.io.wavfile.write``
    produces for float64 / float32 (IEEE-float format: 18-byte fmt chunk, fact chunk) and int16 / int32 / uint8 (PCM)
    arrays; other dtypes go to SciPy (which raises for most of them, as in the reference)."""
    import struct
    data = np.asarray(audio_signal)
    if data.dtype.name not in ("float64", "float32", "int16", "int32", "uint8") or data.ndim not in (1, 2) or data.nbytes > 0xFFFFFF00:
        import scipy.io.wavfile
        scipy.io.wavfile.write(audio_file, sampling_frequency, audio_signal)
        return
    channels = 1 if data.ndim == 1 else data.shape[1]
    item = data.dtype.itemsize
    is_float = data.dtype.kind == "f"
    fmt = struct.pack("<HHIIHH", 3 if is_float else 1, channels, int(sampling_frequency), int(sampling_frequency) * item * channels,
                      channels * item, item * 8) + (b"\x00\x00" if is_float else b"")
    head = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if is_float:
        head += b"fact" + struct.pack("<II", 4, data.shape[0])
    head += b"data" + struct.pack("<I", data.nbytes)
    body = np.ascontiguousarray(data.astype(data.dtype.newbyteorder("<"), copy=False))
    with open(audio_file, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + len(head) + data.nbytes) + b"WAVE" + head)
        body.tofile(fh)


def separate_file(algo, audio_file, background_file=None, foreground_file=None, dtype=np.float64):
    """``wavread`` -> ``algo`` -> ``wavwrite`` of the README example (README.md:62-72) without the samples visiting host
    arrays in between: the file's raw PCM goes to the device, is decoded and normalised there, separated, and the
    background / foreground (``audio - background``) come back as finished file images. Returns the sampling frequency.
    The files equal ``wavwrite(algo(*wavread(audio_file)))`` resp. the same for ``audio - background``."""
    ctx = _native.default_context(_device)
    sampling_frequency = ctx.upload_wav(audio_file)
    ctx.execute(algo, derive_params(sampling_frequency))
    if background_file is not None:
        ctx.write_wav(background_file, "background", dtype)
    if foreground_file is not None:
        ctx.write_wav(foreground_file, "foreground", dtype)
    return sampling_frequency


def specshow(audio_spectrogram, time_duration, maximum_frequency, xtick_step=1, ytick_step=1000):
    """Show a spectrogram in dB with second / Hz ticks (repet.py:949-997)."""
    import matplotlib.pyplot as plt
    number_frequencies, number_times = np.shape(audio_spectrogram)
    per_second = number_times / time_duration
    per_hertz = number_frequencies / maximum_frequency
    plt.imshow(20 * np.log10(audio_spectrogram), aspect="auto", cmap="jet", origin="lower")
    plt.xticks(ticks=np.arange(xtick_step * per_second, number_times, xtick_step * per_second),
               labels=np.arange(xtick_step, time_duration, xtick_step).astype(int))
    plt.yticks(ticks=np.arange(ytick_step * per_hertz, number_frequencies, ytick_step * per_hertz),
               labels=np.arange(ytick_step, maximum_frequency, ytick_step).astype(int))
    plt.xlabel("Time (s)")
    plt.ylabel("Frequency (Hz)")
