"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: clip dealing + scatter/gather, and the
segment-range sharding of ``extended``. The separation itself is injected (the oracle stands in for the
HIP engine, which needs a GPU); what is under test is the sharding, ordering and merge."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from repet import parallel
from repet_synth import synth
from oracle import repet_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, result_file):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = globals()[fn_name](rank, world)
        if rank == 0:
            np.savez(result_file, **out)
    finally:
        dist.destroy_process_group()


def _run(fn_name, tmp_path, world=2):
    result = str(tmp_path / "result.npz")
    mp.spawn(_worker, args=(world, _free_port(), fn_name, result), nprocs=world, join=True)
    with np.load(result) as z:
        return {k: z[k] for k in z.files}


FS = 8000
CLIP_SPECS = [(4.0, 1), (7.5, 2), (3.2, 2), (6.0, 1), (5.0, 2)]


def _clips():
    return [synth(d, FS, c, 20 + i) for i, (d, c) in enumerate(CLIP_SPECS)]


def _case_clips(rank, world):
    clips = _clips() if rank == 0 else None
    got = parallel.separate_clips("original", clips, FS, separate_fn=lambda x, fs: orc.original(x, fs))
    if rank != 0:
        return None
    return {f"clip{i}": y for i, y in enumerate(got)}


def _case_extended(rank, world):
    x = synth(27.0, FS, 2, 31)
    n_seg = len(orc.extended_plan(len(x), FS, orc.Params())[0])
    got = parallel.extended_sharded(x, FS, n_seg, range_fn=lambda a, fs, first, count: orc.extended_range(a, fs, first, count))
    return {"y": got, "n_seg": np.int64(n_seg)} if rank == 0 else None


def test_deal_clips_longest_first_round_robin():
    shares = parallel.deal_clips([10, 50, 30, 50, 20], 2)
    assert shares == [[1, 2, 0], [3, 4]]
    assert sorted(sum(parallel.deal_clips(list(range(17)), 8), [])) == list(range(17))
    assert parallel.deal_clips([], 4) == [[], [], [], []]


def test_segment_ranges_are_contiguous_and_balanced():
    assert parallel.segment_ranges(119, 8) == [(0, 15), (15, 15), (30, 15), (45, 15), (60, 15), (75, 15), (90, 15), (105, 14)]
    assert parallel.segment_ranges(3, 8)[:4] == [(0, 1), (1, 1), (2, 1), (3, 0)]


def test_extended_is_the_sum_of_segment_ranges():
    x = synth(27.0, FS, 2, 31)
    n_seg = len(orc.extended_plan(len(x), FS, orc.Params())[0])
    assert n_seg == 4
    full = orc.extended(x, FS)
    parts = sum(orc.extended_range(x, FS, first, count) for first, count in parallel.segment_ranges(n_seg, 3))
    assert np.max(np.abs(parts - full)) < 1e-12
    # step shorter than the overlap: a sample lies under up to four segments and the in-place fades compound
    p = orc.Params(segment_length=8, segment_step=2)
    n_seg = len(orc.extended_plan(len(x), FS, p)[0])
    assert n_seg == 10
    full = orc.extended(x, FS, p)
    parts = sum(orc.extended_range(x, FS, first, count, p) for first, count in parallel.segment_ranges(n_seg, 3))
    assert np.max(np.abs(parts - full)) < 1e-12


def test_scatter_separate_gather_two_ranks(tmp_path):
    got = _run("_case_clips", tmp_path)
    for i, x in enumerate(_clips()):
        assert np.array_equal(got[f"clip{i}"], orc.original(x, FS))


def test_extended_segments_sharded_over_two_ranks(tmp_path):
    got = _run("_case_extended", tmp_path)
    x = synth(27.0, FS, 2, 31)
    assert int(got["n_seg"]) == 4
    assert np.max(np.abs(got["y"] - orc.extended(x, FS))) < 1e-12
