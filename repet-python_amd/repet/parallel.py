"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on
the MI355X node, "gloo" in the CPU tests).

The REPET path shards without any exchange during compute: a batch of clips is a set of independent
units, and the segments of ``extended`` are independent ``original`` problems whose cross-faded outputs
add up (repet.py:380-414 is linear in the segments). The only communication is the scatter of
waveforms from the root and the gather of results back -- point-to-point sends of fp32 samples. The
engine computes in fp32, but `sim` / `simonline` take the few float64 decisions of their peak picking from
the 48 bits of sample + remainder (``x - float64(float32(x))``): a float64 clip with such remainders
therefore travels as TWO fp32 planes (``split_float64``), so that a worker rank separates it exactly as
``repet.sim`` on the root would (fp32 or PCM-exact clips send one plane: their remainders are zero).

On the RCCL backend the received samples stay on the device: ``dist.recv`` fills a device tensor, the engine
ingests its pointer (``repet_ctx_upload_device``), the result leaves through ``repet_ctx_download_device``
into the tensor that is sent back -- no host bounce on the worker ranks.

``separate_fn`` / ``range_fn`` default to the HIP engine on this rank's device; the CPU tests inject
stand-ins so the sharding and merge logic runs under gloo without a GPU.
"""
import numpy as np


def deal_clips(lengths, world_size):
    """Clip ids per rank: longest first, dealt round-robin (same rule as repet_run_batch in the C ABI)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world_size] for r in range(world_size)]


def segment_ranges(n_segments, world_size):
    """Contiguous [first, first+count) segment ranges per rank, sizes differing by at most one."""
    base, extra = divmod(n_segments, world_size)
    out, first = [], 0
    for r in range(world_size):
        count = base + (1 if r < extra else 0)
        out.append((first, count))
        first += count
    return out


def extended_plan(number_samples, segment_length, segment_step):
    """(number of segments, list of (start, length)) of ``extended`` for sizes in samples (repet.py:266-281,306-322):
    one segment when the clip is shorter than a segment plus a step; else segments every ``segment_step`` samples, the
    last one taking everything that is left."""
    n, length, step = int(number_samples), int(segment_length), int(segment_step)
    if n < length + step:
        return 1, [(0, n)]
    count = 1 + (n - length) // step
    segments = [(j * step, length) for j in range(count - 1)]
    segments.append(((count - 1) * step, n - (count - 1) * step))
    return count, segments


def segment_window(segments, first, count):
    """Samples [lo, hi) that segments ``first .. first+count-1`` read and write: all a rank needs of the clip."""
    if count <= 0:
        return 0, 0
    lo = segments[first][0]
    start, length = segments[first + count - 1]
    return lo, start + length


def split_float64(x):
    """(fp32 samples, fp32 remainders or None) of a float array: what the engine's own upload makes of a float64 array
    (hostio.hip): ``hi = float32(x)``, ``lo = float32(x - float64(hi))``; None when every remainder is zero."""
    x = np.asarray(x)
    hi = np.ascontiguousarray(x, dtype=np.float32)
    if x.dtype != np.float64:
        return hi, None
    lo = (x - hi.astype(np.float64)).astype(np.float32)
    return hi, (lo if np.any(lo) else None)


# ---- the HIP engine on this rank's device ---------------------------------------------------------------------------
_contexts = {}


def _context(key):
    """One engine context per device (``key`` = device index) and role (``key`` = (role, device index))."""
    import repet
    ctx = _contexts.get(key)
    if ctx is None:
        ctx = _contexts[key] = repet.Context(key[1] if isinstance(key, tuple) else key)
    return ctx


def _engine_separate(algo, device):
    """clip -> background, for host arrays (float64 in, float64 out) and for fp32 device tensors (tensor in, tensor out:
    device-resident ingest and egress, no host bounce)."""
    import repet

    def run(x, fs, remainders=None):
        import torch
        if isinstance(x, torch.Tensor) and x.is_cuda:
            ctx = _context(x.device.index)
            # the engine reads interleaved fp32: a float64 wire (or a strided view) is narrowed / packed first, and the result
            # goes back in the wire's dtype. `x32` stays referenced until upload_device has returned (it copies).
            x32 = x.to(torch.float32).contiguous()
            lo32 = remainders.to(torch.float32).contiguous() if remainders is not None else None
            torch.cuda.current_stream(x.device).synchronize()          # the recv that filled x (and the cast) have completed
            ctx.upload_device(x32.data_ptr(), x32.shape[0], x32.shape[1], remainder_ptr=lo32.data_ptr() if lo32 is not None else None)
            ctx.execute(algo, repet.derive_params(fs))
            out = torch.empty_like(x32)
            ctx.download_device(out.data_ptr())
            return out.to(x.dtype)
        repet.set_device(device)
        if remainders is not None:                                     # (host arrays: the float64 waveform, rebuilt exactly)
            x = np.asarray(x, dtype=np.float64) + np.asarray(remainders, dtype=np.float64)
        return getattr(repet, algo)(x, fs)
    return run


def _engine_extended_range(device):
    """(window of the clip, fs, first, count, total samples, first sample of the window) -> the window's share of
    ``extended``: fp32 device tensor in/out on the RCCL path, host arrays otherwise."""
    import repet

    def run(window, fs, first, count, number_samples_total, first_sample):
        import torch
        ctx = _context(window.device.index if isinstance(window, torch.Tensor) and window.is_cuda else device)
        params = repet.derive_params(fs)
        if isinstance(window, torch.Tensor) and window.is_cuda:
            w32 = window.to(torch.float32).contiguous()                # (see _engine_separate: fp32, packed, kept alive)
            torch.cuda.current_stream(window.device).synchronize()
            ctx.upload_device(w32.data_ptr(), w32.shape[0], w32.shape[1])
            ctx.set_window(number_samples_total, first_sample)
            ctx.execute_extended_range(params, first, count)
            out = torch.empty_like(w32)
            ctx.download_device(out.data_ptr())
            return out.to(window.dtype)
        ctx.upload(np.asarray(window))
        ctx.set_window(number_samples_total, first_sample)
        ctx.execute_extended_range(params, first, count)
        return ctx.download()
    return run


# ---- device memory the engine owns, as tensors ----------------------------------------------------------------------
class _DeviceSpan:
    """Borrowed fp32 device memory behind ``__cuda_array_interface__`` (``torch.as_tensor`` wraps it without a copy)."""

    def __init__(self, pointer, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(d) for d in shape), "typestr": "<f4", "data": (int(pointer), False),
                                         "version": 2, "strides": None}


def tensor_view(pointer, shape, device):
    """fp32 tensor over ``prod(shape)`` floats at ``pointer`` on ``device`` (no copy, no ownership: the context that
    owns the memory must outlive the tensor)."""
    import torch
    return torch.as_tensor(_DeviceSpan(pointer, shape), device=torch.device("cuda", device))


def _accepts_remainders(fn):
    """Whether ``fn(x, fs, remainders)`` can be called: a third positional parameter, ``*args`` or a ``remainders`` keyword."""
    import inspect
    try:
        params = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return True
    positional = [q for q in params if q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD)]
    return len(positional) >= 3 or any(q.kind == q.VAR_POSITIONAL for q in params) or any(q.name == "remainders" for q in params)


def _call_separate(fn, x, fs, lo):
    """``fn`` on a received clip. A two-argument ``separate_fn`` (the documented signature before float64 clips travelled as
    two planes) gets the planes folded back into one float64 waveform -- the same rebuild ``_engine_separate`` does for host
    arrays -- instead of a TypeError on the worker ranks only."""
    if lo is None:
        return fn(x, fs)
    if _accepts_remainders(fn):
        return fn(x, fs, lo)
    import torch
    if isinstance(x, torch.Tensor):
        return fn(x.double() + lo.double(), fs)
    return fn(np.asarray(x, dtype=np.float64) + np.asarray(lo, dtype=np.float64), fs)


# ---- transport ------------------------------------------------------------------------------------------------------
def _wire(array, dtype, dev):
    """A host array as a tensor of the wire dtype on the communication device."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(array, dtype=dtype)).to(dev)


def _host(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def separate_clips(algo, clips, sampling_frequency, separate_fn=None, device=None, root=0, wire_dtype=np.float32,
                   stage_device=None, timings=None):
    """Collective over the default process group. ``clips`` (list of (N_i, C_i) float arrays) is read on
    ``root`` only; every rank separates its share; ``root`` returns the list of background signals (float64) in
    the original order, the other ranks return None. Samples travel as ``wire_dtype`` (fp32: what the engine computes
    in; a float64 clip whose fp32 remainders are not all zero sends them as a second plane and ``separate_fn`` is called
    with them as a third argument -- or, when it takes two, with the planes folded back into one float64 waveform); on the
    RCCL backend a worker rank's clips never leave the device, and the ROOT's side of the wire goes through the engine's
    pinned ring (``repet_ctx_upload`` narrows and splits with the host worker threads, ``repet_ctx_download_from`` widens what
    comes back) instead of through pageable NumPy copies. ``stage_device``: under gloo, move what a worker received onto that
    GPU before calling the engine (the device-resident ingest of the RCCL path on a box whose ranks share one GPU).
    ``timings``: a dict that receives this rank's wall times (``compute_ms``: its own separations, ``total_ms``)."""
    import time
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() != "gloo"
    dev = torch.device("cuda", device if device is not None else rank) if on_gpu else torch.device("cpu")
    stage = torch.device("cuda", stage_device) if (stage_device is not None and not on_gpu) else None
    if on_gpu:
        engine_device = dev.index or 0
    else:
        engine_device = stage_device if stage_device is not None else (device or 0)
    fn = separate_fn or _engine_separate(algo, engine_device)
    tdtype = torch.float32 if np.dtype(wire_dtype) == np.float32 else torch.float64
    t_start = time.perf_counter()
    compute_s = 0.0
    # the root's end of an fp32 RCCL wire: a staging context of the engine (pinned ring + host worker threads)
    # (only the root stages anything: a context is a stream, a pinned ring and workspaces)
    io = _context(("io", dev.index)) if (rank == root and on_gpu and separate_fn is None and np.dtype(wire_dtype) == np.float32) else None

    shares_of = lambda shapes_: deal_clips([s[0] for s in shapes_], world)
    meta = [None]
    if rank == root:
        meta = [[(int(c.shape[0]), int(c.shape[1])) for c in clips]]
    dist.broadcast_object_list(meta, src=root)
    shapes = meta[0]
    shares = shares_of(shapes)
    split_wire = np.dtype(wire_dtype) == np.float32             # (a float64 wire carries everything in one plane)

    if rank == root:
        # The root is a PIPELINE (round 6; before, it split every travelling clip, then sent them all, then separated its own,
        # then fetched the results one by one: 8.8 ms per clip of a batch against 5.6 ms for the same clip through the drop-in).
        # A worker's clip leaves as soon as IT has been narrowed -- the worker separates while the root narrows the next one;
        # a one-word header in front of the planes says whether a remainder plane follows, so nobody waits for a table of the
        # whole batch. The receives of all results are posted before the root turns to its own share, which goes through
        # repet_run_stream (upload, kernels and download of neighbouring clips side by side); the results are widened in the
        # order the workers finish their first clips.
        pending, keep = [], []
        order = [(r, i) for k in range(max((len(ids) for ids in shares), default=0)) for r, ids in enumerate(shares) if r != root and k < len(ids) for i in [ids[k]]]
        for r, i in order:                                      # every worker's first clip, then every worker's second, ...
            if split_wire:
                hi, lo = _split_on_device(io, clips[i], dev) if io is not None else split_float64(clips[i])
            else:
                hi, lo = clips[i], None
            hi = hi if isinstance(hi, torch.Tensor) else _wire(hi, wire_dtype, dev)
            lo = None if lo is None else (lo if isinstance(lo, torch.Tensor) else _wire(lo, wire_dtype, dev))
            header = torch.tensor([1 if lo is not None else 0], dtype=torch.int32, device=dev)
            keep += [header, hi, lo]
            pending.append(dist.isend(header, dst=r))
            pending.append(dist.isend(hi, dst=r))
            if lo is not None:
                pending.append(dist.isend(lo, dst=r))
        out = [None] * len(shapes)
        boxes = {i: torch.empty(shapes[i], dtype=tdtype, device=dev) for r, i in order}
        arrivals = [(i, dist.irecv(boxes[i], src=r)) for r, i in order]
        t_c = time.perf_counter()
        own = list(shares[root])
        if separate_fn is None and len(own) > 1 and engine_device is not None:
            import repet
            for i, y in zip(own, repet.run_batch(algo, [np.asarray(clips[i]) for i in own], sampling_frequency, device=engine_device, depth=2)):
                out[i] = y
        else:
            for i in own:
                out[i] = np.ascontiguousarray(_host(fn(np.asarray(clips[i]), sampling_frequency)), dtype=np.float64)
        compute_s = time.perf_counter() - t_c
        for i, req in arrivals:
            req.wait()
            if io is not None:
                torch.cuda.current_stream(dev).synchronize()       # the receive has landed
                out[i] = io.download_from(boxes[i].data_ptr(), shapes[i])
            else:
                out[i] = _host(boxes[i]).astype(np.float64)
            boxes[i] = None
        for req in pending:
            req.wait()
        if timings is not None:
            timings.update(compute_ms=compute_s * 1e3, total_ms=(time.perf_counter() - t_start) * 1e3, clips=len(shares[root]))
        return out

    def receive(i):
        header = torch.empty(1, dtype=torch.int32, device=dev)
        dist.recv(header, src=root)
        t = torch.empty(shapes[i], dtype=tdtype, device=dev)
        dist.recv(t, src=root)
        lo = None
        if int(header.item()):
            lo = torch.empty(shapes[i], dtype=tdtype, device=dev)
            dist.recv(lo, src=root)
        return t, lo

    for i in shares[rank]:                                      # (a clip is separated as soon as it is here: the next one arrives meanwhile)
        t, lo = receive(i)
        if stage is not None:                                  # (gloo wire, engine on a GPU: the worker's device-resident path)
            t, lo = t.to(stage), (lo.to(stage) if lo is not None else None)
        as_arg = (lambda v: v) if (on_gpu or stage is not None) else (lambda v: v.numpy())
        t_c = time.perf_counter()
        y = _call_separate(fn, as_arg(t), sampling_frequency, as_arg(lo) if lo is not None else None)
        compute_s += time.perf_counter() - t_c
        if isinstance(y, torch.Tensor) and not on_gpu:
            y = y.cpu()
        if not isinstance(y, torch.Tensor):
            y = _wire(y, wire_dtype, dev)
        dist.send(y.to(tdtype), dst=root)
    if timings is not None:
        timings.update(compute_ms=compute_s * 1e3, total_ms=(time.perf_counter() - t_start) * 1e3, clips=len(shares[rank]))
    return None


def _split_on_device(io, clip, dev):
    """(fp32 samples, fp32 remainders or None) of a host clip as device tensors, made by the engine's own upload (hostio.hip:
    worker threads narrow into the pinned ring, the remainder plane follows on a copy stream) and copied out of the staging
    context's buffers on its stream -- the next upload may overwrite them."""
    import torch
    io.upload(np.asarray(clip))
    hi_ptr, lo_ptr, count = io.input_view()
    shape = tuple(int(d) for d in np.shape(clip))
    with torch.cuda.stream(torch.cuda.ExternalStream(io.stream(), device=dev)):
        hi = tensor_view(hi_ptr, shape, dev.index).clone()
        lo = tensor_view(lo_ptr, shape, dev.index).clone() if lo_ptr else None
    io.synchronize()
    return hi, lo


def extended_sharded(audio_signal, sampling_frequency, segment_length, segment_step, range_fn=None, device=None, root=0,
                     wire_dtype=np.float32):
    """``repet.extended`` of one long clip with its segments split over the ranks.

    ``audio_signal`` is read on ``root`` only; ``segment_length`` / ``segment_step`` are in SAMPLES
    (``round(repet.segment_length * fs)``, repet.py:266-267). Rank r gets a contiguous range of segments and is sent
    only the samples those segments cover -- (count + 1) segment steps with the default 50 % overlap, about 1/world of
    the clip -- runs them with the whole clip's cross-fade weights (``repet_ctx_set_window``) and returns
    its window; the root adds the windows into place, which touches each of the world-1 shard borders' overlap once.
    Returns the float64 background on ``root``, None elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() != "gloo"
    dev = torch.device("cuda", device if device is not None else rank) if on_gpu else torch.device("cpu")
    fn = range_fn or _engine_extended_range(dev.index or 0)
    tdtype = torch.float32 if np.dtype(wire_dtype) == np.float32 else torch.float64

    meta = [None]
    if rank == root:
        meta = [(int(np.shape(audio_signal)[0]), int(np.shape(audio_signal)[1]))]
    dist.broadcast_object_list(meta, src=root)
    n, channels = meta[0]
    n_segments, segments = extended_plan(n, segment_length, segment_step)
    ranges = segment_ranges(n_segments, world)
    windows = [segment_window(segments, first, count) for first, count in ranges]

    def run_mine(window):
        first, count = ranges[rank]
        return fn(window, sampling_frequency, first, count, n, windows[rank][0])

    if rank == root:
        x = np.asarray(audio_signal)
        pending = []
        for r in range(world):
            lo, hi = windows[r]
            if r != root and hi > lo:
                pending.append(dist.isend(_wire(x[lo:hi], wire_dtype, dev), dst=r))
        # the windows overlap by one cross-fade at every shard border: added up in float64, so the only rounding the sharded
        # result has on top of a single-GPU run is the fp32 rounding of each rank's own window (a single GPU folds the second
        # segment of a border into the first with one fused multiply-add; bit-identity would need the neighbour's border
        # samples on every rank before its last accumulate, i.e. a chain of world - 1 dependent exchanges)
        total = torch.zeros((n, channels), dtype=torch.float64, device=dev)
        lo, hi = windows[root]
        if hi > lo:
            mine = run_mine(_wire(x[lo:hi], wire_dtype, dev) if on_gpu else np.ascontiguousarray(x[lo:hi], dtype=wire_dtype))
            total[lo:hi] += (mine if isinstance(mine, torch.Tensor) else _wire(mine, wire_dtype, dev)).to(torch.float64)
        for req in pending:
            req.wait()
        for r in range(world):                                  # fixed order: the sums do not depend on arrival times
            lo, hi = windows[r]
            if r != root and hi > lo:
                part = torch.empty((hi - lo, channels), dtype=tdtype, device=dev)
                dist.recv(part, src=r)
                total[lo:hi] += part.to(torch.float64)
        return _host(total).astype(np.float64)

    lo, hi = windows[rank]
    if hi > lo:
        window = torch.empty((hi - lo, channels), dtype=tdtype, device=dev)
        dist.recv(window, src=root)
        part = run_mine(window if on_gpu else window.numpy())
        if not isinstance(part, torch.Tensor):
            part = _wire(part, wire_dtype, dev)
        dist.send(part.to(tdtype), dst=root)
    return None


# ---- `extended` with the clip RESIDENT across the GPUs ---------------------------------------------------------------
def halo_plan(windows):
    """Who owns which samples, and which partial sums have to move, when rank r holds the window ``windows[r] = (lo, hi)``
    of its contiguous segment range: rank r OWNS ``[lo_r, lo_next)`` (``lo_next`` = the next non-empty window's start, the
    clip's end for the last one); whatever its segments wrote beyond that belongs to later ranks and is sent to them.
    Returns ``(owned, moves)``: ``owned[r] = (lo, hi)`` and ``moves = [(src, dst, lo, hi), ...]`` (absolute sample ranges,
    ordered by destination, then source -- the order the destination adds them in). With the default 50 % overlap there is
    exactly one move per shard border: one segment step from rank r to rank r + 1."""
    live = [r for r, (lo, hi) in enumerate(windows) if hi > lo]
    owned = [(0, 0)] * len(windows)
    for k, r in enumerate(live):
        lo, hi = windows[r]
        owned[r] = (lo, windows[live[k + 1]][0] if k + 1 < len(live) else hi)
    moves = []
    for q in live:
        for r in live:
            if r >= q:
                break
            lo = max(owned[q][0], windows[r][0])
            hi = min(owned[q][1], windows[r][1])
            if hi > lo:
                moves.append((r, q, lo, hi))
    return owned, moves


class ExtendedShard:
    """This rank's share of ``repet.extended`` on ONE long clip that stays resident across the GPUs of the job (BASELINE
    configs[2]: 10-min clip, 10-s segments sharded over 8 MI355X; repet.py:306-414).

    Rank r keeps the samples its contiguous range of segments covers in its own HBM. ``step()`` enqueues -- on the engine's
    stream, nothing waits on the host -- the rank's segments (``repet_ctx_execute_extended_range_async`` with the whole
    clip's cross-fade weights) and then the only exchange the path has: at every shard border the earlier rank's partial sums
    for the samples the later rank owns travel there (point-to-point over RCCL/xGMI; one segment step = 1.76 MB per border at
    44.1 kHz stereo) and are added in place. Afterwards ``owned`` of every rank is final and the union is the clip.
    ``gather(root)`` assembles it on the root (verification / the drop-in's return value), outside any timed region.

    ``window`` is the rank's own window of the clip (host array, float32 or float64) or None for an empty range; use
    ``ExtendedShard.plan`` to find it. Under gloo (ranks sharing a GPU, CPU wire) the same code runs with the wire bounced
    through host memory."""

    @staticmethod
    def plan(number_samples, segment_length, segment_step, world):
        n_segments, segments = extended_plan(number_samples, segment_length, segment_step)
        ranges = segment_ranges(n_segments, world)
        windows = [segment_window(segments, first, count) for first, count in ranges]
        return n_segments, ranges, windows

    def __init__(self, window, sampling_frequency, number_samples, channels, device=0, group=None, range_fn=None,
                 segment_length=None, segment_step=None):
        """``range_fn(window, fs, first, count, number_samples, first_sample) -> window's share`` replaces the HIP engine
        (CPU tests of the plan and the exchange under gloo); ``segment_length`` / ``segment_step`` in samples default to the
        module parameters (``repet.segment_length * fs``, repet.py:266-267)."""
        import contextlib
        import torch
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.on_gpu = dist.get_backend(group) != "gloo"
        self.device = int(device)
        self.fs = sampling_frequency
        self.n, self.channels = int(number_samples), int(channels)
        self.range_fn = range_fn
        if range_fn is None:
            import repet
            self.params = repet.derive_params(sampling_frequency)
            segment_length, segment_step = self.params.seg_len_samples, self.params.seg_step_samples
        self.n_segments, self.ranges, self.windows = self.plan(self.n, segment_length, segment_step, self.world)
        self.owned, self.moves = halo_plan(self.windows)
        self.first, self.count = self.ranges[self.rank]
        self.lo, self.hi = self.windows[self.rank]
        self.ctx = None
        self.view = None
        self.inbox = {}
        # ranks as the point-to-point calls want them: dst / src of send / recv / P2POp are GLOBAL ranks, the plan's are
        # group-local
        self.peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
        # what travels: fp32 from the engine, float64 from a range_fn (the dtype a root with an empty window must expect)
        self.wire_dtype = torch.float32 if range_fn is None else torch.float64
        if self.hi <= self.lo:
            return
        if window is None or np.shape(window) != (self.hi - self.lo, self.channels):
            raise ValueError(f"rank {self.rank} needs samples [{self.lo}, {self.hi}) of the clip as its window")
        if range_fn is None:
            import repet
            self.ctx = repet.Context(self.device)
            self.ctx.upload(np.asarray(window))
            self.ctx.set_window(self.n, self.lo)
            pointer, count = self.ctx.result_view()
            assert count == (self.hi - self.lo) * self.channels
            self.view = tensor_view(pointer, (self.hi - self.lo, self.channels), self.device)
            self.stream = torch.cuda.ExternalStream(self.ctx.stream(), device=torch.device("cuda", self.device))
            self.stream_scope = lambda: torch.cuda.stream(self.stream)
        else:
            self.host_window = np.array(window)
            self.view = torch.zeros((self.hi - self.lo, self.channels), dtype=torch.float64)
            self.stream_scope = contextlib.nullcontext
        # receive buffers of the partial sums this rank is owed, one per incoming move (allocated once)
        self.inbox = {(src, lo, hi): torch.empty((hi - lo, self.channels), dtype=self.view.dtype, device=self.view.device)
                      for src, dst, lo, hi in self.moves if dst == self.rank}

    def _piece(self, lo, hi):
        return self.view[lo - self.lo:hi - self.lo]

    def step(self):
        """One pass: this rank's segments, then the border exchange. Returns at once (RCCL backend); ``synchronize()`` waits."""
        import torch
        if self.view is None:
            return
        dist = self.dist
        with self.stream_scope():
            if self.ctx is not None:
                self.ctx.execute_extended_range_async(self.params, self.first, self.count)
            else:
                self.view.copy_(torch.from_numpy(np.asarray(self.range_fn(self.host_window, self.fs, self.first, self.count, self.n, self.lo),
                                                            dtype=np.float64)))
            mine = [m for m in self.moves if self.rank in (m[0], m[1])]
            if not mine:
                return
            if self.on_gpu:
                ops = []
                for src, dst, lo, hi in mine:
                    if src == self.rank:
                        ops.append(dist.P2POp(dist.isend, self._piece(lo, hi), self.peer(dst), self.group))
                    else:
                        ops.append(dist.P2POp(dist.irecv, self.inbox[(src, lo, hi)], self.peer(src), self.group))
                for req in dist.batch_isend_irecv(ops):
                    req.wait()                                  # (orders the stream behind the transfer; no host wait)
            else:
                # gloo: CPU wire. Sends first (non-blocking), then the receives in the plan's order.
                self.synchronize()
                pending = [dist.isend(self._piece(lo, hi).cpu().contiguous(), self.peer(dst), self.group) for src, dst, lo, hi in mine if src == self.rank]
                for src, dst, lo, hi in mine:
                    if dst == self.rank:
                        box = torch.empty((hi - lo, self.channels), dtype=self.view.dtype)
                        dist.recv(box, self.peer(src), self.group)
                        self.inbox[(src, lo, hi)].copy_(box)
                for req in pending:
                    req.wait()
            for src, dst, lo, hi in mine:                       # the plan's order: the sums do not depend on arrival times
                if dst == self.rank:
                    self._piece(lo, hi).add_(self.inbox[(src, lo, hi)])

    def synchronize(self):
        if self.ctx is not None:
            self.ctx.synchronize()

    def owned_result(self):
        """fp32 tensor (view) of the samples this rank owns, final after ``step()`` + ``synchronize()``."""
        if self.view is None:
            return None
        lo, hi = self.owned[self.rank]
        return self._piece(lo, hi)

    def gather(self, root=0):
        """The whole background on ``root`` as a float64 array (None elsewhere): every rank's owned samples, in place."""
        import torch
        dist = self.dist
        self.synchronize()
        mine = self.owned_result()
        wire = (lambda t: t) if self.on_gpu else (lambda t: t.cpu())
        if self.rank != root:
            if mine is not None and mine.shape[0] > 0:
                dist.send(wire(mine).contiguous(), self.peer(root), self.group)
            return None
        out = np.zeros((self.n, self.channels), dtype=np.float64)
        for r in range(self.world):
            lo, hi = self.owned[r]
            if hi <= lo:
                continue
            if r == root:
                part = mine
            else:
                part = torch.empty((hi - lo, self.channels), dtype=self.wire_dtype,
                                   device=torch.device("cuda", self.device) if self.on_gpu else "cpu")
                dist.recv(part, self.peer(r), self.group)
            out[lo:hi] = part.cpu().numpy()
        return out

    def close(self):
        self.view = None
        self.inbox = {}
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None
