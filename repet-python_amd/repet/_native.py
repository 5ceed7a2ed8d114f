"""ctypes binding of librepet_hip.so (include/repet_hip.h). No CPU fallback: if the library or a GPU
is missing the calls raise, loudly."""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("REPET_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "librepet_hip.so"))

ABI_VERSION = 4
ORIGINAL, EXTENDED, ADAPTIVE, SIM, SIMONLINE = range(5)
ALGO_IDS = {"original": ORIGINAL, "extended": EXTENDED, "adaptive": ADAPTIVE, "sim": SIM, "simonline": SIMONLINE}
F32, F64, I16 = 0, 1, 2
MAX_STAGES = 16

ERR_BAD_ARG, ERR_TOO_SHORT, ERR_HIP, ERR_OOM, ERR_LIMIT = -1, -2, -3, -4, -5
FLAG_STRICT_REFERENCE = 1      # (ABI 3 opt-in; the default since ABI 4)
FLAG_REFUSE_NONFINITE = 2


class Params(C.Structure):
    _fields_ = [("window_length", C.c_int32), ("step_length", C.c_int32), ("period_lo", C.c_int32),
                ("period_hi", C.c_int32), ("cutoff_bins", C.c_int32), ("filter_order", C.c_int32),
                ("seg_len_frames", C.c_int32), ("seg_step_frames", C.c_int32),
                ("sim_distance_frames", C.c_int32), ("sim_number", C.c_int32), ("buffer_frames", C.c_int32),
                ("flags", C.c_int32), ("seg_len_samples", C.c_int64), ("seg_step_samples", C.c_int64),
                ("sim_threshold", C.c_double)]


class Settings(C.Structure):
    """repet_settings of include/repet_hip.h: the nine module-level parameters of the reference."""
    _fields_ = [("cutoff_frequency", C.c_double), ("period_range", C.c_double * 2), ("segment_length", C.c_double),
                ("segment_step", C.c_double), ("similarity_threshold", C.c_double), ("similarity_distance", C.c_double),
                ("buffer_length", C.c_double), ("filter_order", C.c_int32), ("similarity_number", C.c_int32)]


class WavInfo(C.Structure):
    """repet_wav_info of include/repet_hip.h."""
    _fields_ = [("format", C.c_int32), ("n_channels", C.c_int32), ("sampling_frequency", C.c_int32),
                ("bits_per_sample", C.c_int32), ("bytes_per_sample", C.c_int32), ("reserved0", C.c_int32),
                ("data_offset", C.c_int64), ("n_samples", C.c_int64)]


class Timing(C.Structure):
    _fields_ = [("n_stages", C.c_int32), ("reserved0", C.c_int32), ("total_ms", C.c_float),
                ("stage_ms", C.c_float * MAX_STAGES), ("stage_name", (C.c_char * 24) * MAX_STAGES),
                ("stage_bytes", C.c_double * MAX_STAGES), ("stage_flops", C.c_double * MAX_STAGES)]

    def as_dict(self):
        stages = []
        for i in range(self.n_stages):
            stages.append({"name": self.stage_name[i].value.decode(), "ms": float(self.stage_ms[i]),
                           "bytes": float(self.stage_bytes[i]), "flops": float(self.stage_flops[i])})
        return {"total_ms": float(self.total_ms), "stages": stages}


_P = C.c_void_p
_SIGNATURES = {
    "repet_abi_version": (C.c_int, []),
    "repet_device_count": (C.c_int, []),
    "repet_device_host_cpus": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    "repet_last_error": (C.c_char_p, []),
    "repet_default_settings": (None, [C.POINTER(Settings)]),
    "repet_derive_params": (C.c_int, [C.POINTER(Settings), C.c_double, C.POINTER(Params)]),
    "repet_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "repet_ctx_destroy": (C.c_int, [_P]),
    "repet_ctx_upload": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int32]),
    "repet_ctx_upload_batch": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int32, C.c_int32]),
    "repet_ctx_execute": (C.c_int, [_P, C.c_int, C.POINTER(Params), C.POINTER(Timing)]),
    "repet_ctx_download": (C.c_int, [_P, _P]),
    "repet_ctx_upload_device": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32]),
    "repet_ctx_upload_device_split": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32]),
    "repet_ctx_download_device": (C.c_int, [_P, _P]),
    "repet_last_batch_info": (C.c_int, [C.POINTER(C.c_int64)]),
    "repet_ctx_download_input": (C.c_int, [_P, _P, _P, C.POINTER(C.c_int32)]),
    "repet_ctx_set_window": (C.c_int, [_P, C.c_int64, C.c_int64]),
    "repet_ctx_set_strict_reference": (C.c_int, [_P, C.c_int]),
    "repet_ctx_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "repet_ctx_result_view": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "repet_ctx_input_view": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_int64)]),
    "repet_ctx_download_from": (C.c_int, [_P, _P, C.c_int64, _P]),
    "repet_ctx_execute_extended_range_async": (C.c_int, [_P, C.POINTER(Params), C.c_int64, C.c_int64]),
    "repet_wav_parse": (C.c_int, [_P, C.c_int64, C.POINTER(WavInfo)]),
    "repet_ctx_upload_wav": (C.c_int, [_P, _P, C.c_int64, C.POINTER(WavInfo)]),
    "repet_ctx_result_wav": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "repet_ctx_set_sampling_frequency": (C.c_int, [_P, C.c_int32]),
    "repet_host_alloc": (C.c_void_p, [C.c_size_t]),
    "repet_host_free": (None, [C.c_void_p]),
    "repet_ctx_execute_async": (C.c_int, [_P, C.c_int, C.POINTER(Params)]),
    "repet_ctx_synchronize": (C.c_int, [_P]),
    "repet_ctx_timing_series_begin": (C.c_int, [_P, C.c_int32]),
    "repet_ctx_timing_series_end": (C.c_int, [_P, C.POINTER(Timing), C.POINTER(C.c_int32)]),
    "repet_ctx_download_foreground": (C.c_int, [_P, _P]),
    "repet_ctx_spectrogram": (C.c_int, [_P, C.c_int, C.c_int32, _P, C.c_int64]),
    "repet_extended_segment_count": (C.c_int64, [C.c_int64, C.POINTER(Params)]),
    "repet_ctx_execute_extended_range": (C.c_int, [_P, C.POINTER(Params), C.c_int64, C.c_int64, C.POINTER(Timing)]),
    "repet_release_thread_ctx": (C.c_int, []),
    "repet_median_network_info": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "repet_run": (C.c_int, [C.c_int, _P, C.c_int, C.c_int64, C.c_int32, C.POINTER(Params), _P, C.c_int,
                            C.POINTER(Timing)]),
    "repet_run_batch": (C.c_int, [C.c_int, C.c_int32, C.POINTER(_P), C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int32), C.POINTER(Params), C.POINTER(_P), C.c_int32]),
    "repet_run_batch_rccl": (C.c_int, [C.c_int, C.c_int32, C.POINTER(_P), C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int32), C.POINTER(Params), C.POINTER(_P), C.c_int32]),
    "repet_run_stream": (C.c_int, [C.c_int, C.c_int32, C.POINTER(_P), C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(Params), C.POINTER(_P), C.c_int, C.c_int32]),
    "repet_host_conversion_selftest": (C.c_int64, [C.c_int64, C.c_uint32]),
    "repet_frame_count": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "repet_stft": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64]),
    "repet_istft": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P, C.c_int64]),
    "repet_selfsim": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "repet_selfsim_records": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "repet_similarity": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64, C.c_int32, _P]),
    "repet_acorr": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "repet_beat_spectrum": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int32]),
    "repet_beat_spectrogram": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "repet_periods": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "repet_local_maxima": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, _P, _P]),
    "repet_mask_period": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "repet_mask_adaptive": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, C.c_int32, _P]),
    "repet_mask_sim": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P]),
    "repet_rank_columns": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    "repet_mask_sim_ranked": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "repet_ctx_last_periods": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "repet_ctx_last_median_path": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "repet_ctx_last_median_codes": (C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    "repet_ctx_last_sim_indices": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32]),
    "repet_ctx_last_frame_count": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "repet_ctx_last_refine_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "repet_ctx_last_exact_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "repet_online_open": (C.c_int, [C.c_int, C.c_int32, C.POINTER(Params), C.POINTER(_P)]),
    "repet_online_push": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "repet_online_finish": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "repet_online_close": (C.c_int, [_P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64 with the system library's soname, so
    whichever is loaded first serves both; with the system one first, a later ``import torch`` finds no GPU (its other
    bundled libraries do not match). When a PyTorch-ROCm is installed its runtime is therefore loaded first -- found on
    disk, not imported. REPET_HIP_RUNTIME=system skips this."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("REPET_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load librepet_hip.so once. Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"librepet_hip.so not found at {LIB_PATH}: build it with `make -C repet-python_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _preload_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.repet_abi_version() != ABI_VERSION:
            raise RuntimeError("librepet_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().repet_last_error().decode(errors="replace")
    if rc in (ERR_BAD_ARG, ERR_TOO_SHORT):
        raise ValueError(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise RuntimeError(f"librepet_hip error {rc}: {msg}")


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_input(audio_signal):
    """(array, dtype code) in a layout/dtype the ABI takes without a host-side conversion pass."""
    a = np.asarray(audio_signal)
    if a.dtype == np.float64:
        code = F64
    elif a.dtype == np.float32:
        code = F32
    elif a.dtype == np.int16:
        code = I16
    else:
        a = a.astype(np.float64)
        code = F64
    return np.ascontiguousarray(a), code


def result_array(shape):
    """A fresh C-contiguous float64 array for a result, backed by a pinned host buffer of the library's recycling pool
    (repet_host_alloc): when the caller drops the array its buffer goes back to the pool, so the next result lands in
    memory that is already faulted in and pinned -- a fresh np.empty of a 3-minute clip takes 31 000 page faults on
    first touch. Falls back to np.empty when the pool declines (REPET_PINNED_RESULTS=0, or too much outstanding)."""
    count = 1
    for d in shape:
        count *= int(d)
    if count == 0 or os.environ.get("REPET_PINNED_RESULTS") == "0":
        return np.empty(shape, dtype=np.float64)
    address = lib().repet_host_alloc(count * 8)
    if not address:
        return np.empty(shape, dtype=np.float64)
    buffer = (C.c_double * count).from_address(address)
    weakref.finalize(buffer, lib().repet_host_free, address)      # the array (and every view of it) keeps `buffer` alive
    return np.frombuffer(buffer, dtype=np.float64, count=count).reshape(shape)


class Context:
    """One device context: upload a clip once, execute variants on the resident copy, download."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        if lib().repet_device_count() < 1:
            raise RuntimeError("no HIP device visible: the REPET engine has no CPU fallback")
        check(lib().repet_ctx_create(int(device), C.byref(self._h)))
        self.shape = None

    def close(self):
        if self._h:
            lib().repet_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    @property
    def handle(self):
        return self._h

    def upload(self, audio_signal):
        n, c = np.shape(audio_signal)
        a, code = as_input(audio_signal)
        check(lib().repet_ctx_upload(self._h, ptr(a), code, n, c))
        self.shape = (n, c)

    def upload_batch(self, audio_signals):
        """Equal-shape clips ``(number_clips, number_samples, number_channels)`` (or a list of such clips) made
        resident together; ``download`` then returns the same shape. ``simonline`` runs every stage once over
        all of them."""
        clips = np.stack([np.asarray(a) for a in audio_signals]) if not isinstance(audio_signals, np.ndarray) else audio_signals
        if clips.ndim != 3:
            raise ValueError("audio_signals must be (number_clips, number_samples, number_channels)")
        b, n, c = clips.shape
        a, code = as_input(clips.reshape(b * n, c))
        check(lib().repet_ctx_upload_batch(self._h, ptr(a), code, n, c, b))
        self.shape = (b, n, c)

    def upload_device(self, data_ptr, number_samples, number_channels, number_clips=1, remainder_ptr=None):
        """fp32 interleaved samples already in device memory (e.g. ``tensor.data_ptr()`` of what an RCCL recv filled;
        its producer must have finished: synchronise that stream first). No host bounce. ``remainder_ptr``: the fp32
        remainders of a float64 waveform (``x - float64(float32(x))``) in the same layout -- with them the float64 decisions
        of the peak picking see what a single-GPU call on the float64 array sees."""
        check(lib().repet_ctx_upload_device_split(self._h, C.c_void_p(int(data_ptr)), C.c_void_p(int(remainder_ptr)) if remainder_ptr else None,
                                                  int(number_samples), int(number_channels), int(number_clips)))
        self.shape = (number_samples, number_channels) if number_clips == 1 else (number_clips, number_samples, number_channels)

    def resident_input(self):
        """(fp32 samples, fp32 remainders, has_remainders) of the resident clip as the engine holds it."""
        hi = np.empty(self.shape, dtype=np.float32)
        lo = np.empty(self.shape, dtype=np.float32)
        flag = C.c_int32(0)
        check(lib().repet_ctx_download_input(self._h, ptr(hi), ptr(lo), C.byref(flag)))
        return hi, lo, bool(flag.value)

    def download_device(self, data_ptr):
        """The result as fp32 interleaved samples into device memory of at least ``prod(self.shape)`` floats."""
        check(lib().repet_ctx_download_device(self._h, C.c_void_p(int(data_ptr))))

    def set_strict_reference(self, on=True):
        """NaN / infinite samples are let through as repet.py lets them (the default since ABI 4); ``on=False``: host uploads
        with such samples are refused (REPET_FLAG_REFUSE_NONFINITE)."""
        check(lib().repet_ctx_set_strict_reference(self._h, 1 if on else 0))

    def stream(self):
        """The context's HIP stream as an integer handle (``torch.cuda.ExternalStream(ctx.stream())`` orders torch's
        sends, receives and adds behind a run without a host wait)."""
        h = C.c_void_p()
        check(lib().repet_ctx_stream(self._h, C.byref(h)))
        return int(h.value or 0)

    def result_view(self):
        """(device pointer, number of fp32 values) of the last run's result inside the context: borrowed, valid until the
        next upload; writes to it are ordered on ``stream()``."""
        ptr_, n = C.c_void_p(), C.c_int64()
        check(lib().repet_ctx_result_view(self._h, C.byref(ptr_), C.byref(n)))
        return int(ptr_.value or 0), int(n.value)

    def input_view(self):
        """(pointer to the resident fp32 samples, pointer to their fp32 remainders or None, number of values)."""
        hi, lo, n = C.c_void_p(), C.c_void_p(), C.c_int64()
        check(lib().repet_ctx_input_view(self._h, C.byref(hi), C.byref(lo), C.byref(n)))
        return int(hi.value or 0), (int(lo.value) if lo.value else None), int(n.value)

    def download_from(self, data_ptr, shape):
        """fp32 values in device memory (e.g. a result received from a peer) widened into a fresh float64 host array through
        the context's pinned ring."""
        out = result_array(shape)
        check(lib().repet_ctx_download_from(self._h, C.c_void_p(int(data_ptr)), int(np.prod(shape)), ptr(out)))
        return out

    def execute_extended_range_async(self, params, first, n_segments):
        check(lib().repet_ctx_execute_extended_range_async(self._h, C.byref(params), int(first), int(n_segments)))

    def set_window(self, number_samples_total, first_sample):
        """The resident samples are ``[first_sample, first_sample + N)`` of a clip of ``number_samples_total`` samples
        (for ``execute_extended_range`` on a rank that holds only its own segments' samples)."""
        check(lib().repet_ctx_set_window(self._h, int(number_samples_total), int(first_sample)))

    def upload_wav(self, audio_file):
        """A WAVE file becomes the resident clip: its raw PCM bytes cross PCIe and are decoded and normalised on the
        device exactly as ``repet.wavread`` would (repet.py:914-931). Returns the sampling frequency."""
        image = np.fromfile(audio_file, dtype=np.uint8)
        info = WavInfo()
        check(lib().repet_ctx_upload_wav(self._h, ptr(image), image.size, C.byref(info)))
        self.shape = (info.n_samples, info.n_channels)
        return info.sampling_frequency

    def write_wav(self, audio_file, which="background", dtype=np.float64, sampling_frequency=None):
        """Write the background (or ``audio - background``) of the last run as the file ``repet.wavwrite`` writes for
        an array of ``dtype`` float64 / float32 (repet.py:934-946), straight from the device result."""
        if sampling_frequency is not None:
            check(lib().repet_ctx_set_sampling_frequency(self._h, int(sampling_frequency)))
        code = {"background": 1, "foreground": 2}[which]
        item = np.dtype(dtype).itemsize
        image = np.empty(58 + int(np.prod(self.shape)) * item, dtype=np.uint8)
        written = C.c_int64()
        check(lib().repet_ctx_result_wav(self._h, code, F64 if item == 8 else F32, ptr(image), image.size, C.byref(written)))
        image[:written.value].tofile(audio_file)

    def execute(self, algo, params, timing=False):
        t = Timing() if timing else None
        check(lib().repet_ctx_execute(self._h, ALGO_IDS[algo] if isinstance(algo, str) else algo,
                                      C.byref(params), C.byref(t) if timing else None))
        return t.as_dict() if timing else None

    def execute_async(self, algo, params):
        check(lib().repet_ctx_execute_async(self._h, ALGO_IDS[algo] if isinstance(algo, str) else algo, C.byref(params)))

    def synchronize(self):
        check(lib().repet_ctx_synchronize(self._h))

    def timing_series_begin(self, n_steps):
        """Every ``execute_async`` up to ``n_steps`` records its own per-stage events; no host wait between the runs."""
        check(lib().repet_ctx_timing_series_begin(self._h, int(n_steps)))

    def timing_series_end(self):
        """Wait for the stream; mean per-stage device times over the runs since ``timing_series_begin``."""
        t, n = Timing(), C.c_int32()
        check(lib().repet_ctx_timing_series_end(self._h, C.byref(t), C.byref(n)))
        d = t.as_dict()
        d["steps"] = int(n.value)
        return d

    def execute_extended_range(self, params, first, n_segments, timing=False):
        t = Timing() if timing else None
        check(lib().repet_ctx_execute_extended_range(self._h, C.byref(params), int(first), int(n_segments), C.byref(t) if timing else None))
        return t.as_dict() if timing else None

    def download(self):
        out = result_array(self.shape)
        check(lib().repet_ctx_download(self._h, ptr(out)))
        return out

    def foreground(self):
        """audio_signal - background_signal of the last run (float64, computed on the device)."""
        out = np.empty(self.shape, dtype=np.float64)
        check(lib().repet_ctx_download_foreground(self._h, ptr(out)))
        return out

    def spectrogram(self, which, window_length):
        """(F, T) magnitude spectrogram of the channel-mean mixture / background / foreground signal."""
        code = {"mixture": 0, "background": 1, "foreground": 2}[which] if isinstance(which, str) else int(which)
        t = lib().repet_frame_count(self.shape[0], window_length, window_length // 2, 1)
        spec = np.empty((t, window_length // 2 + 1), dtype=np.float32)
        check(lib().repet_ctx_spectrogram(self._h, code, window_length, ptr(spec), t))
        return spec.T.astype(np.float64)

    def last_periods(self, capacity):
        out = np.empty(max(capacity, 1), dtype=np.int32)
        n = C.c_int32()
        check(lib().repet_ctx_last_periods(self._h, ptr(out), capacity, C.byref(n)))
        return out[:n.value].copy()

    def last_median_path(self):
        """'f32', 'rank' or 'bits': the form of sim's median the last run took (see repet_ctx_last_median_path)."""
        v = C.c_int32()
        check(lib().repet_ctx_last_median_path(self._h, C.byref(v)))
        return ("f32", "rank", "bits")[v.value]

    def last_median_codes(self, number_bins):
        """uint32[channels][frames][number_bins] of the last `sim` run on the bit-sliced path (repet_ctx_last_median_codes)."""
        t = self.last_frame_count()
        ch = self.shape[-1]
        out = np.empty((ch, t, number_bins), dtype=np.uint32)
        check(lib().repet_ctx_last_median_codes(self._h, ptr(out), t, number_bins))
        return out

    def last_frame_count(self):
        t = C.c_int64()
        check(lib().repet_ctx_last_frame_count(self._h, C.byref(t)))
        return t.value

    def last_refine_stats(self):
        """Near-tie refinement counters of the last sim/simonline run (see repet_ctx_last_refine_stats)."""
        out = (C.c_int64 * 4)()
        check(lib().repet_ctx_last_refine_stats(self._h, out))
        return {"rows_refined": out[0], "elements_refined": out[1], "decisions_changed": out[2], "flat_rows": out[3]}

    def last_exact_stats(self):
        """Second level of the peak picking (float64 spectra; see repet_ctx_last_exact_stats)."""
        out = (C.c_int64 * 8)()
        check(lib().repet_ctx_last_exact_stats(self._h, out))
        return {"rows_exact": out[0], "elements_exact": out[1], "rows_changed": out[2], "level2_max_diff": out[3] * 1e-12,
                "unit_rows_f64": out[4], "input_has_remainders": bool(out[5]), "rows_fast_path": out[6], "rows_handed_on": out[7]}

    def last_sim_indices(self, n_rows, number):
        idx = np.empty((max(n_rows, 1), number), dtype=np.int32)
        cnt = np.empty(max(n_rows, 1), dtype=np.int32)
        check(lib().repet_ctx_last_sim_indices(self._h, ptr(idx), ptr(cnt), n_rows, number))
        return idx[:n_rows], cnt[:n_rows]


_default_ctx = {}


def default_context(device=0):
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = _default_ctx[device] = Context(device)
    return ctx


class OnlineSeparator:
    """Streaming online REPET-SIM: ``push(chunk)`` returns the background samples that became final,
    ``finish()`` the rest; the concatenation equals ``repet.simonline`` of the whole signal."""

    def __init__(self, params, n_channels, device=0):
        self._h = C.c_void_p()
        self._channels = int(n_channels)
        self._window = int(params.window_length)
        if lib().repet_device_count() < 1:
            raise RuntimeError("no HIP device visible: the REPET engine has no CPU fallback")
        check(lib().repet_online_open(int(device), self._channels, C.byref(params), C.byref(self._h)))

    def push(self, audio_chunk):
        n, c = np.shape(audio_chunk)
        if c != self._channels:
            raise ValueError("chunk has %d channels, the stream %d" % (c, self._channels))
        a, code = as_input(audio_chunk)
        cap = n + self._window
        out = np.empty((cap, c), dtype=np.float64)
        written = C.c_int64()
        check(lib().repet_online_push(self._h, ptr(a), code, n, ptr(out), cap, C.byref(written)))
        return out[:written.value].copy()

    def finish(self):
        cap = 4 * self._window + 16
        while True:
            out = np.empty((cap, self._channels), dtype=np.float64)
            written = C.c_int64()
            rc = lib().repet_online_finish(self._h, ptr(out), cap, C.byref(written))
            if rc == ERR_BAD_ARG and b"capacity" in lib().repet_last_error():
                cap *= 4
                continue
            check(rc)
            return out[:written.value].copy()

    def close(self):
        if self._h:
            lib().repet_online_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
