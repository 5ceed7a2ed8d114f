"""Multi-GPU host logic: one process per GPU over torch.distributed (backend "nccl" = RCCL over xGMI on
the MI355X node, "gloo" in the CPU tests).

The REPET path shards without any exchange during compute: a batch of clips is a set of independent
units, and the segments of ``extended`` are independent ``original`` problems whose cross-faded outputs
add up (repet.py:380-414 is linear in the segments). The only communication is the scatter of
waveforms from the root and the gather of results back -- point-to-point sends of a few MB each.

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


def _engine_separate(algo, device):
    import repet

    def run(x, fs):
        repet.set_device(device)
        return getattr(repet, algo)(x, fs)
    return run


def _send(t, dst, dist, device):
    dist.send(t.to(device), dst=dst)


def _recv(shape, dtype, src, dist, device):
    import torch
    t = torch.empty(shape, dtype=dtype, device=device)
    dist.recv(t, src=src)
    return t.cpu()


def separate_clips(algo, clips, sampling_frequency, separate_fn=None, device=None, root=0):
    """Collective over the default process group. ``clips`` (list of (N_i, C_i) float arrays) is read on
    ``root`` only; every rank separates its share; ``root`` returns the list of background signals in
    the original order, the other ranks return None."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device("cuda", device if device is not None else rank)
    fn = separate_fn or _engine_separate(algo, dev.index or 0)

    meta = [None]
    if rank == root:
        meta = [[(int(c.shape[0]), int(c.shape[1])) for c in clips]]
    dist.broadcast_object_list(meta, src=root)
    shapes = meta[0]
    shares = deal_clips([s[0] for s in shapes], world)

    mine = {}
    if rank == root:
        for r, ids in enumerate(shares):
            for i in ids:
                x = torch.from_numpy(np.ascontiguousarray(clips[i], dtype=np.float64))
                if r == root:
                    mine[i] = x.numpy()
                else:
                    _send(x, r, dist, dev)
    else:
        for i in shares[rank]:
            mine[i] = _recv(shapes[i], torch.float64, root, dist, dev).numpy()

    done = {i: np.ascontiguousarray(fn(x, sampling_frequency), dtype=np.float64) for i, x in mine.items()}

    if rank == root:
        out = [None] * len(shapes)
        for r, ids in enumerate(shares):
            for i in ids:
                out[i] = done[i] if r == root else _recv(shapes[i], torch.float64, r, dist, dev).numpy()
        return out
    for i in shares[rank]:
        _send(torch.from_numpy(done[i]), root, dist, dev)
    return None


def _engine_extended_range(device):
    import repet

    def run(x, fs, first, count):
        ctx = repet.Context(device)
        try:
            ctx.upload(x)
            ctx.execute_extended_range(repet.derive_params(fs), first, count)
            return ctx.download()
        finally:
            ctx.close()
    return run


def extended_sharded(audio_signal, sampling_frequency, n_segments, range_fn=None, device=None, root=0):
    """``repet.extended`` of one long clip with its segments split over the ranks.

    Every rank holds ``audio_signal`` (broadcast it first if only the root has it), runs its contiguous
    range of segments and returns a full-length array that is zero outside its range; the partial results
    are summed onto ``root`` (one reduce of the waveform, the only collective on this path)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device("cuda", device if device is not None else rank)
    fn = range_fn or _engine_extended_range(dev.index or 0)
    first, count = segment_ranges(n_segments, world)[rank]
    part = fn(audio_signal, sampling_frequency, first, count) if count > 0 else np.zeros(np.shape(audio_signal))
    t = torch.from_numpy(np.ascontiguousarray(part, dtype=np.float64)).to(dev)
    dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
    return t.cpu().numpy() if rank == root else None
