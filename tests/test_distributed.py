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
    out = {}
    for name, wire in (("f64", np.float64), ("f32", np.float32)):
        got = parallel.separate_clips("original", clips, FS, separate_fn=_original_from_planes, wire_dtype=wire)
        if rank == 0:
            out.update({f"{name}_clip{i}": y for i, y in enumerate(got)})
    return out if rank == 0 else None


def _original_from_planes(x, fs, remainders=None):
    x = np.asarray(x, dtype=np.float64)
    if remainders is not None:
        x = x + np.asarray(remainders, dtype=np.float64)
    return orc.original(x, fs)


def _amplified_remainders(x, fs, remainders=None):
    """Stand-in "separation" that shows which bits of the waveform reached the rank: the part of every sample below its fp32
    rounding, scaled up so that it survives the fp32 wire back."""
    x = np.asarray(x, dtype=np.float64)
    if remainders is not None:
        x = x + np.asarray(remainders, dtype=np.float64)
    return (x - x.astype(np.float32).astype(np.float64)) * 2.0 ** 20


def _case_remainders(rank, world):
    clips = None
    if rank == 0:
        clips = _clips()
        clips[3] = clips[3].astype(np.float32).astype(np.float64)      # exact in fp32 (what wavread yields): no second plane
    got = parallel.separate_clips("sim", clips, FS, separate_fn=_amplified_remainders, wire_dtype=np.float32)
    return {f"clip{i}": y for i, y in enumerate(got)} if rank == 0 else None


def _window_range(window, fs, first, count, n_total, first_sample, p=None):
    """Stand-in for the engine on a rank: the rank holds only `window` = samples [first_sample, first_sample + len) of the
    clip; the oracle wants a whole clip, and reads nothing outside the window for these segments."""
    window = np.asarray(window, dtype=np.float64)
    x = np.zeros((n_total, window.shape[1]))
    x[first_sample:first_sample + len(window)] = window
    return orc.extended_range(x, fs, first, count, p)[first_sample:first_sample + len(window)]


def _case_extended(rank, world):
    x = synth(27.0, FS, 2, 31) if rank == 0 else None
    p = orc.Params()
    out = {}
    for name, wire in (("f64", np.float64), ("f32", np.float32)):
        got = parallel.extended_sharded(x, FS, round(p.segment_length * FS), round(p.segment_step * FS), range_fn=_window_range, wire_dtype=wire)
        if rank == 0:
            out[name] = got
    # a step shorter than the overlap: windows of neighbouring ranks overlap by several segments' worth
    p2 = orc.Params(segment_length=8, segment_step=2)
    got = parallel.extended_sharded(x, FS, 8 * FS, 2 * FS, wire_dtype=np.float64,
                                    range_fn=lambda w, fs, first, count, n, s0: _window_range(w, fs, first, count, n, s0, p2))
    if rank == 0:
        out["short_step"] = got
    return out if rank == 0 else None


def _two_argument_original(x, fs):
    """The documented two-argument ``separate_fn``: must keep working when float64 clips travel as two planes."""
    return orc.original(np.asarray(x, dtype=np.float64), fs)


def _case_two_argument_fn(rank, world):
    clips = _clips() if rank == 0 else None
    timings = {}
    got = parallel.separate_clips("original", clips, FS, separate_fn=_two_argument_original, wire_dtype=np.float32, timings=timings)
    assert timings["clips"] >= 2 and timings["total_ms"] >= timings["compute_ms"] > 0
    return {f"clip{i}": y for i, y in enumerate(got)} if rank == 0 else None


def _case_resident_extended(rank, world):
    """ExtendedShard under gloo with the oracle as the engine: every rank holds its window, runs its segments, the borders'
    partial sums move to their owners, the root gathers the owned samples."""
    out = {}
    x = synth(27.0, FS, 2, 31)                       # (every rank synthesises the clip: its window is resident from the start)
    for name, (length, step) in (("default", (10, 5)), ("short_step", (8, 2)), ("gaps", (10, 7.5))):
        p = orc.Params(segment_length=length, segment_step=step)
        fn = lambda w, fs, first, count, n, s0, p=p: _window_range(w, fs, first, count, n, s0, p)
        n_seg, ranges, windows = parallel.ExtendedShard.plan(len(x), round(length * FS), round(step * FS), world)
        lo, hi = windows[rank]
        shard = parallel.ExtendedShard(x[lo:hi] if hi > lo else None, FS, len(x), 2, range_fn=fn,
                                       segment_length=round(length * FS), segment_step=round(step * FS))
        for _ in range(2):                            # a second step must not add the borders twice
            shard.step()
        got = shard.gather(0)
        if rank == 0:
            out[name] = got
        shard.close()
    return out if rank == 0 else None


def test_halo_plan_owners_and_moves():
    """Every sample has exactly one owner; a move carries what an earlier rank's segments wrote into a later rank's samples."""
    n_seg, ranges, windows = parallel.ExtendedShard.plan(26460000, 441000, 220500, 8)         # cfg 3 on 8 GPUs
    owned, moves = parallel.halo_plan(windows)
    assert n_seg == 119 and owned[0] == (0, 15 * 220500) and owned[7] == (105 * 220500, 26460000)
    assert [hi - lo for lo, hi in owned[:7]] == [15 * 220500] * 7
    assert moves == [(r, r + 1, (15 * (r + 1)) * 220500, (15 * (r + 1) + 1) * 220500) for r in range(7)]   # one step per border
    # more ranks than segments: the empty ranks own nothing and move nothing
    n_seg, ranges, windows = parallel.ExtendedShard.plan(27 * FS, 10 * FS, 5 * FS, 6)
    owned, moves = parallel.halo_plan(windows)
    assert n_seg == 4 and windows[4] == (0, 0) and owned[4] == (0, 0) and owned[3][1] == 27 * FS
    assert all(src < 4 and dst < 4 for src, dst, _, _ in moves)
    # a step of a quarter segment: a rank's last segment reaches over several later ranks' samples
    n_seg, ranges, windows = parallel.ExtendedShard.plan(27 * FS, 8 * FS, 2 * FS, 5)
    owned, moves = parallel.halo_plan(windows)
    covered = np.zeros(27 * FS, dtype=int)
    for lo, hi in owned:
        covered[lo:hi] += 1
    assert np.all(covered == 1)
    assert any(dst - src >= 2 for src, dst, _, _ in moves)
    for src, dst, lo, hi in moves:
        assert windows[src][0] <= lo < hi <= windows[src][1] and owned[dst][0] <= lo < hi <= owned[dst][1]


@pytest.mark.parametrize("world", [2, 3])
def test_resident_extended_shards_exchange_their_borders(tmp_path, world):
    got = _run("_case_resident_extended", tmp_path, world=world)
    x = synth(27.0, FS, 2, 31)
    for name, (length, step) in (("default", (10, 5)), ("short_step", (8, 2)), ("gaps", (10, 7.5))):
        want = orc.extended(x, FS, orc.Params(segment_length=length, segment_step=step))
        assert np.max(np.abs(got[name] - want)) < 1e-12, name


def test_two_argument_separate_fn_still_works(tmp_path):
    got = _run("_case_two_argument_fn", tmp_path)
    for i, x in enumerate(_clips()):
        assert np.max(np.abs(got[f"clip{i}"] - orc.original(x, FS))) < 1e-5
    assert parallel._accepts_remainders(_original_from_planes) and not parallel._accepts_remainders(_two_argument_original)
    assert parallel._accepts_remainders(lambda *a: None) and parallel._accepts_remainders(lambda x, fs, remainders=None: None)


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


def test_extended_plan_and_windows_match_the_oracle():
    """parallel.extended_plan restates repet.py:266-281,306-322 for the host logic; a rank's window is what its segments
    touch: (count + 1) steps with the default 50 % overlap -- about 1/world of the clip, not the clip."""
    for seconds, length, step in [(27.0, 10, 5), (14.9, 10, 5), (15.1, 10, 5), (31.7, 8, 2), (36.0, 10, 7.5)]:
        n = round(seconds * FS)
        p = orc.Params(segment_length=length, segment_step=step)
        segs, _ = orc.extended_plan(n, FS, p)
        count, mine = parallel.extended_plan(n, round(length * FS), round(step * FS))
        assert count == len(segs) and mine == [(int(a), int(b)) for a, b in segs]
    count, segs = parallel.extended_plan(26460000, 441000, 220500)             # cfg 3: 600 s at 44.1 kHz
    assert count == 119
    windows = [parallel.segment_window(segs, f, c) for f, c in parallel.segment_ranges(count, 8)]
    assert windows[0] == (0, 16 * 220500) and windows[7] == (105 * 220500, 26460000)
    assert all(hi - lo <= 16 * 220500 + 220500 for lo, hi in windows)          # <= (15 + 1) steps (+ the last segment's remainder)
    assert parallel.segment_window(segs, 5, 0) == (0, 0)


def test_scatter_separate_gather_two_ranks(tmp_path):
    got = _run("_case_clips", tmp_path)
    for i, x in enumerate(_clips()):
        assert np.array_equal(got[f"f64_clip{i}"], orc.original(x, FS))
        # fp32 on the wire (what the engine computes in): a worker's clips are narrowed on the way out and back
        assert np.max(np.abs(got[f"f32_clip{i}"] - orc.original(x, FS))) < 1e-5


def test_float64_remainders_travel_with_the_clips(tmp_path):
    """A float64 clip is sent as two fp32 planes (samples and remainders) so that a worker rank separates the 48 bits a
    single-GPU call sees: every rank's stand-in returns the part of its samples below their fp32 rounding, and that must be
    the original clip's -- on the root's own clips and on the worker's alike; a clip that is exact in fp32 sends one plane."""
    got = _run("_case_remainders", tmp_path)
    clips = _clips()
    clips[3] = clips[3].astype(np.float32).astype(np.float64)
    shares = parallel.deal_clips([len(c) for c in clips], 2)
    assert 3 in shares[1] and len(shares[1]) >= 2                       # the worker rank had a clip of either kind
    for i, x in enumerate(clips):
        want = (x - x.astype(np.float32).astype(np.float64)) * 2.0 ** 20
        assert (np.max(np.abs(want)) == 0) if i == 3 else (np.max(np.abs(want)) > 1e-3)
        # what crossed the wire twice (fp32 remainder out, fp32 result back): 2^-24 relative on values below 2^-4
        assert np.max(np.abs(got[f"clip{i}"] - want)) < 1e-8, i
    hi, lo = parallel.split_float64(clips[0])
    assert lo is not None and np.array_equal(hi.astype(np.float64) + lo.astype(np.float64), clips[0].astype(np.float32).astype(np.float64) + lo)
    assert np.max(np.abs(hi.astype(np.float64) + lo.astype(np.float64) - clips[0])) < 2.0 ** -46
    assert parallel.split_float64(clips[3])[1] is None and parallel.split_float64(clips[0].astype(np.float32))[1] is None


def test_extended_segments_sharded_over_two_ranks(tmp_path):
    got = _run("_case_extended", tmp_path)
    x = synth(27.0, FS, 2, 31)
    assert np.max(np.abs(got["f64"] - orc.extended(x, FS))) < 1e-12
    assert np.max(np.abs(got["f32"] - orc.extended(x, FS))) < 2e-6              # fp32 on the wire and in the root's sums
    assert np.max(np.abs(got["short_step"] - orc.extended(x, FS, orc.Params(segment_length=8, segment_step=2)))) < 1e-12


def test_bench_refuses_a_world_that_is_not_what_gpus_says():
    """`python bench.py --gpus N` without RANK in the environment starts its own N ranks -- after counting the devices, which
    on this CPU box stops it with a clear message -- and inside a job whose WORLD_SIZE differs from --gpus it refuses to run
    (round 3 silently ran one rank and reported n_gpus 1)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    alone = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert alone.returncode != 0 and "this node shows 0 GPU(s)" in alone.stderr, alone.stderr[-500:]
    inside = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"],
                            env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=300)
    assert inside.returncode != 0 and "the two must agree" in inside.stderr, inside.stderr[-500:]
