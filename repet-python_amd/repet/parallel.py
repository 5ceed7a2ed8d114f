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


def _context(device):
    import repet
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = _contexts[device] = repet.Context(device)
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


# ---- transport ------------------------------------------------------------------------------------------------------
def _wire(array, dtype, dev):
    """A host array as a tensor of the wire dtype on the communication device."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(array, dtype=dtype)).to(dev)


def _host(t):
    return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)


def separate_clips(algo, clips, sampling_frequency, separate_fn=None, device=None, root=0, wire_dtype=np.float32):
    """Collective over the default process group. ``clips`` (list of (N_i, C_i) float arrays) is read on
    ``root`` only; every rank separates its share; ``root`` returns the list of background signals (float64) in
    the original order, the other ranks return None. Samples travel as ``wire_dtype`` (fp32: what the engine computes
    in; a float64 clip whose fp32 remainders are not all zero sends them as a second plane and ``separate_fn`` is called
    with them as a third argument); on the RCCL backend a worker rank's clips never leave the device."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    on_gpu = dist.get_backend() != "gloo"
    dev = torch.device("cuda", device if device is not None else rank) if on_gpu else torch.device("cpu")
    fn = separate_fn or _engine_separate(algo, dev.index or 0)
    tdtype = torch.float32 if np.dtype(wire_dtype) == np.float32 else torch.float64

    shares_of = lambda shapes_: deal_clips([s[0] for s in shapes_], world)
    planes = {}
    meta = [None]
    if rank == root:
        shapes = [(int(c.shape[0]), int(c.shape[1])) for c in clips]
        with_lo = [False] * len(shapes)
        if np.dtype(wire_dtype) == np.float32:                 # (a float64 wire carries everything in one plane)
            for r, ids in enumerate(shares_of(shapes)):
                if r != root:
                    for i in ids:
                        planes[i] = split_float64(clips[i])
                        with_lo[i] = planes[i][1] is not None
        meta = [(shapes, with_lo)]
    dist.broadcast_object_list(meta, src=root)
    shapes, with_lo = meta[0]
    shares = shares_of(shapes)

    if rank == root:
        pending = []
        for r, ids in enumerate(shares):                       # the workers' clips leave first, then the root computes
            if r != root:
                for i in ids:
                    if i in planes:
                        hi, lo = planes.pop(i)
                        pending.append(dist.isend(_wire(hi, wire_dtype, dev), dst=r))
                        if lo is not None:
                            pending.append(dist.isend(_wire(lo, wire_dtype, dev), dst=r))
                    else:
                        pending.append(dist.isend(_wire(clips[i], wire_dtype, dev), dst=r))
        out = [None] * len(shapes)
        for i in shares[root]:
            out[i] = np.ascontiguousarray(_host(fn(np.asarray(clips[i]), sampling_frequency)), dtype=np.float64)
        for req in pending:
            req.wait()
        for r, ids in enumerate(shares):
            if r != root:
                for i in ids:
                    t = torch.empty(shapes[i], dtype=tdtype, device=dev)
                    dist.recv(t, src=r)
                    out[i] = _host(t).astype(np.float64)
        return out

    received = []
    for i in shares[rank]:
        t = torch.empty(shapes[i], dtype=tdtype, device=dev)
        dist.recv(t, src=root)
        lo = None
        if with_lo[i]:
            lo = torch.empty(shapes[i], dtype=tdtype, device=dev)
            dist.recv(lo, src=root)
        received.append((t, lo))
    for t, lo in received:
        if lo is None:
            y = fn(t if on_gpu else t.numpy(), sampling_frequency)
        else:
            y = fn(t if on_gpu else t.numpy(), sampling_frequency, lo if on_gpu else lo.numpy())
        if not isinstance(y, torch.Tensor):
            y = _wire(y, wire_dtype, dev)
        dist.send(y.to(tdtype), dst=root)
    return None


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
