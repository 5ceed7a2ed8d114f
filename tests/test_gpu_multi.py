"""The multi-GPU paths on real devices. Two kinds of test:

* gated on ``repet_device_count() >= 2`` -- they move bytes between physical MI355Xs (RCCL over xGMI) and compare with the
  single-GPU result; on the one-GPU box they are skipped, on an 8-GPU node they run for every device count 2 .. 8 with no edit;
* always run -- the same host logic with the ranks SHARING device 0 (gloo wire, the worker's device-resident ingest forced with
  ``stage_device``), and ``bench.py --gpus N`` under ``REPET_BENCH_BACKEND=gloo``: partitioning, border exchange, verification
  and the JSON line of the multi-rank bench are executed on the box the driver tests on.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import repet
from repet import _native, parallel
from repet_synth import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    try:
        return int(_native.lib().repet_device_count())
    except Exception:  # noqa: BLE001 -- collection on a box without the library
        return 0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


FS = 22050
CLIPS = [(21.0, 2, 3), (14.0, 1, 4), (17.0, 2, 5), (15.5, 2, 6), (19.0, 1, 7)]


def _clips():
    return [synth(d, FS, c, s) for d, c, s in CLIPS]


def _single(algo, x, fs=FS):
    ctx = repet.Context(0)
    ctx.upload(x)
    ctx.execute(algo, repet.derive_params(fs))
    y = ctx.download()
    ctx.close()
    return y


# ---- workers of the multi-process tests (module level: mp.spawn pickles them by name) ----------------------------------
def _worker(rank, world, port, backend, share_gpu, case, result_file):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    device = 0 if share_gpu else rank
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = globals()[case](rank, world, device, share_gpu)
        if rank == 0:
            np.savez(result_file, **out)
    finally:
        dist.destroy_process_group()


def _spawn(case, tmp_path, world, backend, share_gpu):
    import torch.multiprocessing as mp
    result = str(tmp_path / f"{case}_{world}_{backend}.npz")
    mp.spawn(_worker, args=(world, _free_port(), backend, share_gpu, case, result), nprocs=world, join=True)
    with np.load(result) as z:
        return {k: z[k] for k in z.files}


def _case_separate(rank, world, device, share_gpu):
    out = {}
    clips = _clips() if rank == 0 else None
    for algo in ("sim", "simonline"):
        timings = {}
        got = parallel.separate_clips(algo, clips, FS, device=device, stage_device=device if share_gpu else None, timings=timings)
        assert timings["clips"] == len(parallel.deal_clips([round(d * FS) for d, _, _ in CLIPS], world)[rank])
        if rank == 0:
            out.update({f"{algo}{i}": y for i, y in enumerate(got)})
    return out


def _case_extended(rank, world, device, share_gpu):
    fs = 16000
    x = synth(63.0, fs, 2, 12)
    out = {}
    got = parallel.extended_sharded(x if rank == 0 else None, fs, 10 * fs, 5 * fs, device=device)
    if rank == 0:
        out["scattered"] = got
    _, _, windows = parallel.ExtendedShard.plan(len(x), 10 * fs, 5 * fs, world)
    lo, hi = windows[rank]
    shard = parallel.ExtendedShard(x[lo:hi] if hi > lo else None, fs, len(x), 2, device=device)
    for _ in range(3):                                  # steps are idempotent: the borders are not added twice
        shard.step()
    got = shard.gather(0)
    shard.close()
    if rank == 0:
        out["resident"] = got
    return out


def _check_separate(got):
    for i, x in enumerate(_clips()):
        for algo in ("sim", "simonline"):
            assert np.array_equal(got[f"{algo}{i}"], _single(algo, x)), (algo, i)


def _check_extended(got):
    fs = 16000
    x = synth(63.0, fs, 2, 12)
    want = repet.extended(x, fs)
    for name in ("scattered", "resident"):
        assert got[name].shape == want.shape
        assert np.max(np.abs(got[name] - want)) < 2e-6, name          # one fp32 rounding at the shard borders (parallel.py)


# ---- ranks sharing the one GPU of the box (always run) ------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 3])
def test_scatter_separate_gather_with_ranks_sharing_a_gpu(tmp_path, world):
    """parallel.separate_clips with the HIP engine on every rank: float64 clips leave the root as samples + remainders, a worker
    moves what it received into device memory and runs the device-resident ingest (repet_ctx_upload_device_split), the root ends
    with results BIT-IDENTICAL to single-GPU calls -- `sim` (float64 decisions in its peak picking) and `simonline`."""
    _check_separate(_spawn("_case_separate", tmp_path, world, "gloo", True))


@pytest.mark.parametrize("world", [2, 3])
def test_extended_shards_with_ranks_sharing_a_gpu(tmp_path, world):
    """Both forms of the sharded `extended` on the real engine: scattered from the root (extended_sharded) and resident across
    the ranks with the border exchange (ExtendedShard: repet_ctx_execute_extended_range_async + result views + stream)."""
    _check_extended(_spawn("_case_extended", tmp_path, world, "gloo", True))


def test_result_and_input_views_are_the_engines_buffers():
    """ABI 3: repet_ctx_result_view / repet_ctx_input_view / repet_ctx_stream / repet_ctx_download_from."""
    import torch
    x = synth(9.0, FS, 2, 8)
    ctx = repet.Context(0)
    ctx.upload(x)
    hi_ptr, lo_ptr, count = ctx.input_view()
    assert count == x.size and lo_ptr
    hi, lo = parallel.split_float64(x)
    ctx.synchronize()
    assert np.array_equal(parallel.tensor_view(hi_ptr, x.shape, 0).cpu().numpy(), hi)
    assert np.array_equal(parallel.tensor_view(lo_ptr, x.shape, 0).cpu().numpy(), lo)
    ctx.execute("original", repet.derive_params(FS))
    ptr, count = ctx.result_view()
    view = parallel.tensor_view(ptr, x.shape, 0)
    want = ctx.download()
    assert np.array_equal(view.cpu().numpy().astype(np.float64), want)
    assert np.array_equal(ctx.download_from(view.data_ptr(), x.shape), want)
    # torch work enqueued on the engine's stream is ordered behind the run
    with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", 0))):
        ctx.execute_async("original", repet.derive_params(FS))
        doubled = view * 2
    ctx.synchronize()
    assert np.array_equal(doubled.cpu().numpy().astype(np.float64), 2 * want)
    ctx.upload(x.astype(np.float32))
    assert ctx.input_view()[1] is None                 # fp32 input: no remainder plane
    ctx.close()


def _bench(args, env_extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert run.returncode == 0, (run.returncode, run.stderr[-3000:], run.stdout[-1000:])
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_config3_sharded_over_ranks_dry_run():
    """`bench.py --gpus 3 --config 3` with the ranks sharing this box's GPU: ONE 600-s clip, 119 segments in three contiguous
    ranges, border exchange inside the step, result gathered and compared with the root's single-GPU separation."""
    line = _bench(["--gpus", "3", "--config", "3", "--steps", "2", "--warmup", "1", "--series", "1", "--prewarm-ms", "5"],
                  {"REPET_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 3 and line["scaling"] == "strong" and line["ranks_share_gpus"] is True and line["backend"] == "gloo"
    assert line["verified"]["ok"] is True and line["verified"]["max_abs_difference"] <= 2e-6
    assert len(line["per_rank_ms_per_step"]) == 3 and line["config"]["clips_per_step"] == 1
    assert abs(line["value"] - 600.0 / (line["ms_per_step"] * 1e-3)) / line["value"] < 0.01          # ONE clip per step, whatever N


def test_bench_config5_and_config2_over_ranks_dry_run():
    """config 5: 64 clips in total, 32 per rank at N = 2 (strong scaling), every rank verifies a clip of its batch against the
    single call; config 2: one clip per rank (weak) and the scatter / gather leg, verified bit for bit against the root."""
    line = _bench(["--gpus", "2", "--config", "5", "--steps", "1", "--warmup", "1", "--series", "1", "--prewarm-ms", "5"],
                  {"REPET_BENCH_BACKEND": "gloo"})
    assert line["scaling"] == "strong" and line["config"]["clips_per_step"] == 64 and line["verified"]["ok"] is True
    line = _bench(["--gpus", "2", "--duration", "40", "--steps", "2", "--warmup", "1", "--series", "1", "--prewarm-ms", "5"],
                  {"REPET_BENCH_BACKEND": "gloo"})
    assert line["scaling"] == "weak" and line["verified"]["ok"] is True and line["verified"]["per_rank"] == [True, True]
    sg = line["scatter_gather"]
    assert sg["verified"] is True and sg["clips"] == 8 and len(sg["per_rank"]) == 2 and all(r["clips"] == 4 for r in sg["per_rank"])


# ---- two or more physical MI355Xs (skipped on the one-GPU box) ------------------------------------------------------------
multi = pytest.mark.skipif(_device_count() < 2, reason="needs at least two GPUs")


def _counts():
    return list(range(2, max(_device_count(), 2) + 1))


@multi
@pytest.mark.parametrize("k", _counts())
def test_run_batch_transports_agree_across_physical_devices(k):
    """repet_run_batch on k devices: RCCL transport (grouped ncclSend / ncclRecv over xGMI) == host transport == single calls,
    bit for bit, for `sim` on float64 clips with remainders and for `simonline`."""
    clips = _clips() + [synth(12.0 + i, FS, 2, 40 + i) for i in range(max(0, k - 2))]
    for algo in ("sim", "simonline"):
        want = [_single(algo, x) for x in clips]
        over_rccl = repet.run_batch(algo, clips, FS, n_devices=k, transport="rccl")
        info = repet.last_batch_info()
        assert info["transport"] == "rccl" and info["clips_sent"] == len(clips) - len(range(0, len(clips), k))
        if algo == "sim":
            assert info["clips_with_remainders"] == len(clips)
        over_host = repet.run_batch(algo, clips, FS, n_devices=k, transport="host")
        for a, b, w in zip(over_rccl, over_host, want):
            assert np.array_equal(a, w) and np.array_equal(b, w), algo


@multi
@pytest.mark.parametrize("k", _counts())
def test_parallel_over_rccl_matches_one_gpu(tmp_path, k):
    """One process per GPU under the RCCL backend: scatter / separate / gather bit-identical to single-GPU calls; `extended`
    sharded both ways within one fp32 rounding of the single-GPU result."""
    _check_separate(_spawn("_case_separate", tmp_path, k, "nccl", False))
    _check_extended(_spawn("_case_extended", tmp_path, k, "nccl", False))


@multi
def test_bench_over_rccl_verifies_itself():
    """The driver's command on every device this node has: config 3 (segment ranges + border exchange over xGMI) and config 2
    with the scatter / gather leg; both lines must carry a passed verification and rccl_ranks == N."""
    n = _device_count()
    line = _bench(["--gpus", str(n), "--config", "3", "--steps", "5", "--warmup", "2", "--series", "2"], {})
    assert line["rccl_ranks"] == n and line["backend"] == "nccl" and line["verified"]["ok"] is True and line["scaling"] == "strong"
    line = _bench(["--gpus", str(n), "--steps", "5", "--warmup", "2", "--series", "2"], {})
    assert line["rccl_ranks"] == n and line["verified"]["ok"] is True and line["scatter_gather"]["verified"] is True
