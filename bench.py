#!/usr/bin/env python3
"""bench.py -- headline benchmark: repet.sim throughput on the MI355X, in audio-seconds per second.

A "step" is one full REPET-SIM separation (BASELINE.json config 2: one 180-s 44.1 kHz stereo synthetic
clip) of a clip that is already resident in HBM: STFT -> cosine self-similarity GEMM -> peak picking ->
median mask -> iSTFT, all on the engine's HIP stream (repet_ctx_execute). With N ranks every rank owns
its own clip (seed = rank): independent units, no data-path collective, weak scaling.

The other BASELINE configs keep the partitioning BASELINE.json states for them:
  --config 3  repet.extended on ONE 600-s clip; with N ranks its 119 segments are split into N contiguous ranges, every
              rank keeps only its range's samples resident, runs them, and the partial sums at the N-1 shard borders move to
              their owners over RCCL point-to-point (repet/parallel.py ExtendedShard): total work fixed, "strong" scaling;
              after the timed region the root gathers the result and compares it with its own single-GPU separation.
  --config 5  repet.simonline on 64 30-s clips, 64/N per rank, batched through every stage: total work fixed, "strong".

    python bench.py                       # 1 GPU, finishes in a few minutes (incl. the CPU baseline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `roofline` describes the stage with the largest device time, measured
with HIP events recorded on the engine's stream during the timed steps; `cpu_baseline` is the NumPy
oracle (a port of the reference, validated against it) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "repet-python_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_F16_PEAK_TF = 2500.0 # MI355X_MICROARCH.md: ~2.5 PF dense f16 / bf16
# v_min / v_max / v_pk_min_u16-class VALU instructions: one wave64 instruction per 4 cycles per SIMD (measured 4.2-4.75,
# tools/microbench/cex_rate.hip), 1 024 SIMDs, 2.4 GHz max clock (MI355X_MICROARCH.md) -> 614.4 G wave-instructions/s
VALU_QUARTER_RATE_GINSTR = 1024 * 2.4 / 4.0
# the guide's VALU issue rate for a full-rate instruction (wave64 over two cycles on a SIMD-32): the frame every VALU figure can
# be put in, whatever its instruction class -- profiles/r03_valu_rate.txt holds the control rows (v_add / v_fma / v_and at
# 2.0-2.4 cycles) beside the min / max / packed-min classes (4.2-4.75 cycles) at the measured shader clock
VALU_FULL_RATE_GINSTR = 1024 * 2.4 / 2.0
# rows gathered from a table that an XCD's L2 holds: 16.8-18.8 TB/s chip-wide (MI355X_MICROARCH.md, 'Indexed rows'); the
# vector memory path of the CUs caps the same traffic at 256 CUs x 64 B/clk x 2.4 GHz = 39 TB/s, which it never reaches
L2_GATHER_PEAK_GBS = 17800.0
PMC_FILE = "r06_pmc_traffic.json"
# arithmetic type of the path: fp32 end to end, except that the similarity GEMM of sim / simonline runs on the f16 matrix
# cores as a three-product split of the fp32 operands (22 significant bits per product, fp32 accumulate; gram_f16.hip)
DTYPE_NOTE = {"sim": "f32 (similarity GEMM: f16x3 split of the fp32 unit rows, fp32 accumulate; median: exact selection on bit-sliced rank codes)",
              "simonline": "f32 (similarity band: f16x3 split of the fp32 unit rows, fp32 accumulate)"}


def cpu_baseline(fs, channels, seconds, algo="sim", n_clips=1):
    """Time the float64 NumPy oracle (CPU port of the reference) on a bounded sample of the workload."""
    import numpy as np
    from oracle import repet_oracle as orc
    from repet_synth import synth
    xs = [synth(seconds, fs, channels, k) for k in range(n_clips)]
    t0 = time.perf_counter()
    for x in xs:
        orc.ALGORITHMS[algo](x, fs)
    dt = time.perf_counter() - t0
    try:
        from threadpoolctl import threadpool_info
        blas = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
    except Exception:  # noqa: BLE001
        blas = os.cpu_count() or 1
    return {"value": round(seconds * n_clips / dt, 4), "unit": "audio-seconds/sec", "cores": int(blas), "kind": "port",
            "sample": f"oracle.{algo} (NumPy float64 port of repet.py) on {n_clips} x {seconds:g}-s {fs} Hz {channels}-ch synth clip "
                      f"(the bench workload itself when 180 s), {dt:.1f} s wall; single-threaded except the "
                      f"matrix products ({blas} BLAS threads); host has {os.cpu_count()} logical cores"}


def bind_to_gpu_node(local_rank):
    """Before anything of this process touches the GPU: put the process on the CPUs of the NUMA node its GPU hangs off -- what
    `numactl --cpunodebind` does for a GPU job. The device times do not care; the drop-in call (host threads narrowing the array
    into pinned memory, two PCIe copies) takes 3.8-3.9 ms from there against 4.3-5.9 elsewhere on a two-socket host
    (profiles/r05_dropin_numa.txt), and only a process that STARTS there gets it (the HIP runtime's own threads and pinned
    staging stay where they were created). The node is asked of a child process (the library reads the device's PCI address),
    so that this process is bound before it creates a single thread. REPET_BENCH_NUMA=0: leave the process where it is.
    Returns (all CPUs it was allowed, the CPUs it is bound to) or None."""
    if os.environ.get("REPET_BENCH_NUMA") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    import subprocess
    allowed = sorted(os.sched_getaffinity(0))
    code = ("import sys; sys.path[:0] = [%r]; import repet; from repet import _native; n = _native.lib().repet_device_count(); "
            "print(' '.join(str(c) for c in repet.device_host_cpus(int(sys.argv[1]) %% max(n, 1))) if n > 0 else '')") % os.path.join(ROOT, "repet-python_amd")
    try:
        out = subprocess.run([sys.executable, "-c", code, str(local_rank)], capture_output=True, text=True, timeout=180)
        cpus = sorted(int(tok) for tok in out.stdout.split())
    except Exception:  # noqa: BLE001 -- binding is an optimisation of the host side, never a reason to fail
        return None
    if not cpus or len(cpus) >= len(allowed):
        return None
    os.sched_setaffinity(0, cpus)
    return allowed, cpus


def scatter_gather_leg(dist, rank, world, local_rank, algo, fs, channels, seconds, n_clips):
    """The multi-GPU data path as a user meets it (repet/parallel.py): the root holds `n_clips` clips in host
    RAM as float64, deals them over the ranks (fp32 over RCCL/xGMI; a worker's samples stay on its device), every rank
    separates its share, the root ends with all results in host RAM. Wall time of that whole collective on the root,
    after one warm-up round (RCCL point-to-point channels, contexts, workspaces). Outside the headline's timed region."""
    import numpy as np
    import repet
    from repet import parallel
    from repet_synth import synth
    clips = [synth(seconds, fs, channels, seed=s) for s in range(n_clips)] if rank == 0 else None
    shared_gpu = dist is not None and dist.get_backend() == "gloo"
    timings = {}

    def one_round():
        if dist is None:                     # one process, one GPU: the root's own share is everything -- two clips in flight (repet_run_stream)
            return repet.run_batch(algo, clips, fs, device=local_rank, depth=int(os.environ.get("REPET_BENCH_DEPTH", "2")))
        timings.clear()
        return parallel.separate_clips(algo, clips, fs, device=local_rank, stage_device=local_rank if shared_gpu else None, timings=timings)

    one_round()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    out = one_round()
    dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {k: round(v, 2) if isinstance(v, float) else v for k, v in timings.items()})
    if rank != 0:
        return None
    # every gathered clip against the root's OWN single-GPU separation of the same array: bit for bit (sim's float64 decisions
    # travel with the remainder plane; the worker's device-resident ingest is the same fp32 samples the root would upload)
    repet.set_device(local_rank)
    differing = [i for i, (o, c) in enumerate(zip(out, clips)) if not (o.shape == c.shape and np.array_equal(o, getattr(repet, algo)(c, fs)))]
    entry = {"value": round(seconds * n_clips / dt, 1), "unit": "audio-seconds/sec", "ms": round(dt * 1e3, 2), "clips": n_clips,
             "verified": len(out) == n_clips and not differing,
             "verified_how": "every gathered clip == the root's own single-GPU repet.%s of the same float64 array, bit for bit" % algo,
             "per_rank": per_rank,
             "note": f"{n_clips} x {seconds:g}-s clips in the root's host RAM (float64) -> scattered as fp32 (+ remainder planes) over "
                     f"{('gloo, ranks sharing GPUs (dry run)' if shared_gpu else 'RCCL point-to-point') if dist is not None else 'PCIe only (one GPU)'} -> repet.{algo} on {world} rank(s) -> "
                     "gathered back into the root's host RAM (float64); wall time on the root, second round"}
    if differing:
        entry["clips_that_differ"] = differing
    return entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="steps per timed region (default 400: with 5 regions about two seconds of GPU work at "
                                                           "cfg 2, long enough for a coarse GPU-activity sampler to see it)")
    ap.add_argument("--series", type=int, default=5, help="timed regions of exactly --steps steps each; value / ms_per_step are the MEDIAN series, all of them are listed")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32-GEMM variant (a child run of this script under REPET_GRAM=f32)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--duration", type=float, default=180.0, help="clip length in seconds (config 2: 180)")
    ap.add_argument("--algo", default="sim")
    ap.add_argument("--fs", type=int, default=44100)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--sync-steps", action="store_true", help="one blocking call per step instead of K steps enqueued back to back")
    ap.add_argument("--prewarm-ms", type=float, default=200.0, help="untimed pre-warm before the warm-up steps (device clocks), by wall time")
    ap.add_argument("--scatter-limit", type=float, default=240.0, help="seconds the scatter/gather leg may take before it is given up")
    ap.add_argument("--clips", type=int, default=1, help="independent clips per rank and step (config 5: 64 in total)")
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configs[i-1]: 1 original on the reference's 23-s example clip (--wav, only where the "
                         "reference tree is present), 2 sim 180 s (headline), 3 extended 600 s, 4 adaptive 300 s 48 kHz "
                         "mono, 5 simonline 30-s clips (64 over all ranks)")
    ap.add_argument("--wav", default="/root/reference/audio_file.wav", help="config 1: the reference's example clip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scatter-lenient", action="store_true", help="exit 0 even when the scatter/gather leg fails (default: 1, after the headline line has been printed)")
    ap.add_argument("--no-scatter", action="store_true", help="skip the scatter -> separate -> gather leg (8 clips from the root's host RAM)")
    ap.add_argument("--no-batch", action="store_true", help="config 5: one context and stream per clip instead of one batch context")
    ap.add_argument("--cpu-seconds", type=float, default=180.0,
                    help="length of the CPU-baseline clip (default: the whole config-2 clip, ~25 s of host time)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Started as plain `python bench.py --gpus N`: this process becomes the launcher. It starts the N ranks as a CHILD
        # (python -m torch.distributed.run, one rank per GPU over RCCL) before anything here has touched the GPU -- counting
        # the devices does not -- relays the ranks' output (rank 0 prints the one JSON line) and exits with the child's code.
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not (os.environ.get("REPET_BENCH_BACKEND") == "gloo" and have >= 1):
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s)")
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stderr.write(f"[bench] --gpus {args.gpus} without RANK in the environment: launching {' '.join(cmd)}\n")
        sys.stderr.flush()
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    example_clip = None
    if args.config == 1:
        import scipy.io.wavfile            # the clip is not redistributed with this repository: build container only
        if not os.path.exists(args.wav):
            raise SystemExit(f"--config 1 needs the reference's example clip ({args.wav}); it is not part of this repository")
        args.fs, pcm = scipy.io.wavfile.read(args.wav)
        example_clip = pcm / pow(2, pcm.itemsize * 8 - 1)
        args.algo, args.duration, args.channels = "original", len(pcm) / args.fs, pcm.shape[1]
    elif args.config == 3:
        args.algo, args.duration = "extended", 600.0
    elif args.config == 4:
        args.algo, args.duration, args.fs, args.channels = "adaptive", 300.0, 48000, 1
    elif args.config == 5:
        args.algo, args.duration = "simonline", 30.0
        args.clips = max(1, 64 // max(args.gpus, 1))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the CPU baseline FIRST, while this process may still use every CPU it was given (its BLAS threads are created here, unbound);
    # then the process binds itself to the GPU's NUMA node, before torch / the library create a thread or pin a page
    cpu_line = None
    if world == 1 and not args.no_cpu_baseline and example_clip is None:
        # a bounded sample of the same workload: the whole clip for configs 2, 3 and 4 (10-30 s of host time each), 8 of the 64
        # clips of config 5
        sample_s = args.cpu_seconds if args.config == 2 else args.duration
        cpu_line = cpu_baseline(args.fs, args.channels, min(sample_s, args.duration), args.algo, 8 if args.config == 5 else 1)
    numa = bind_to_gpu_node(local_rank)
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a job of WORLD_SIZE {world}: the two must agree "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}, or plain python bench.py --gpus {args.gpus})")

    import ctypes
    import numpy as np
    import torch
    import repet
    from repet import _native
    from repet_synth import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the REPET engine has no CPU fallback")
    # REPET_BENCH_BACKEND=gloo (test switch): the N ranks talk over gloo and may SHARE GPUs (rank r on device r mod visible),
    # so that the multi-rank partitionings, the border exchange and the verification run on a one-GPU box. The line says so
    # ("backend": "gloo", "ranks_share_gpus": true) and is not a scaling measurement.
    backend = os.environ.get("REPET_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("REPET_BENCH_BACKEND must be nccl or gloo")
    visible = torch.cuda.device_count()
    ranks_share_gpus = backend == "gloo" and visible < world
    if ranks_share_gpus:
        local_rank = local_rank % visible
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or (os.environ.get("REPET_BENCH_DIST") == "1" and "MASTER_ADDR" in os.environ):   # the switch: 1-rank check of the RCCL path
        if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", "INFO"):
            os.environ["NCCL_DEBUG"] = "WARN"        # RCCL prints its version banner on stdout: keep stdout to the one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")    # ... and its warnings (topology, iommu) on stderr
        import torch.distributed as dist
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"{backend} communicator of {dist.get_world_size()} rank(s) for --gpus {args.gpus}")

    fs, channels = args.fs, args.channels
    params = repet.derive_params(fs)
    ctxs = []
    batched = args.clips > 1 and args.algo == "simonline" and not args.no_batch       # equal-shape clips: every stage once over all of them
    clips = []
    # config 3 on N > 1 ranks: ONE clip, its segments in N contiguous ranges (BASELINE.json configs[2]); total work is fixed
    sharded = args.config == 3 and world > 1
    strong = sharded or (args.config == 5 and world > 1)
    shard = None
    if sharded:
        from repet import parallel
        clip = synth(args.duration, fs, channels, seed=0)          # every rank synthesises the clip and keeps only its window
        _, _, windows_ = parallel.ExtendedShard.plan(len(clip), params.seg_len_samples, params.seg_step_samples, world)
        lo_, hi_ = windows_[rank]
        shard = parallel.ExtendedShard(clip[lo_:hi_] if hi_ > lo_ else None, fs, len(clip), channels, device=local_rank)
        clips = [clip]
        args.clips = 1
    for k in range(args.clips if not sharded else 0):           # inputs resident in HBM (fp32, interleaved) before timing starts
        clip = example_clip if example_clip is not None else synth(args.duration, fs, channels, seed=rank * args.clips + k)
        clips.append(clip)
    if sharded:
        pass
    elif batched:
        ctx = repet.Context(local_rank)
        ctx.upload_batch(np.stack(clips))
        ctxs.append(ctx)
    else:
        for clip in clips:
            ctx = repet.Context(local_rank)
            ctx.upload(clip)
            ctxs.append(ctx)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The device reaches its steady clocks only after some tens of milliseconds of work (a 1.1-ms step measures 1.26 ms
    # right after start, 1.16 after three steps, 1.11 after thirty): an untimed pre-warm by wall time, declared in the
    # line as config.prewarm_ms, precedes the W warm-up steps so that small W / K do not measure the ramp.
    if dist is not None:
        # A fresh RCCL communicator does its real set-up at its first collectives, and the device runs slower for a while
        # afterwards (steps timed right behind the first barrier: 1.26 / 1.18 / 1.14 ms at K = 5 / 10 / 20 against 1.03):
        # the first barriers are spent HERE, in front of the pre-warm, so that the timed region starts on a settled device.
        for _ in range(4):
            barrier()
    t_pre = time.perf_counter()
    if sharded:
        # (the pre-warm's length must not depend on a rank's own clock: the border exchange pairs the ranks' steps one to one)
        n_pre = max(1, int(args.prewarm_ms / 1.2))
        for _ in range(n_pre + args.warmup):
            shard.step()
        shard.synchronize()
    while not sharded and (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for ctx in ctxs:
            ctx.execute(args.algo, params)
    for _ in range(args.warmup if not sharded else 0):
        for ctx in ctxs:
            ctx.execute(args.algo, params)
    # The K steps are enqueued back to back on the context's stream, as a job that separates one clip after another would:
    # no host wait between them (a blocking call per step left the device idle for 25-30 us while the host read its events).
    # Every step still records its own per-stage HIP events on that stream (timing series): the stage times below are
    # the means over exactly the K steps of the series that is reported.
    series = len(ctxs) == 1 and not args.sync_steps and 1 <= args.steps <= 4096      # (the series holds 17 events per step)
    per_rank_ms = []

    def timed_region():
        """Exactly K steps between two barriers; returns (elapsed seconds, per-stage totals, per-stage metadata)."""
        st_ms, st_meta = {}, {}
        barrier()
        t0 = time.perf_counter()
        if series:
            ctxs[0].timing_series_begin(args.steps)
        for _ in range(args.steps):
            if sharded:
                shard.step()                     # this rank's segments + the border exchange, enqueued; nothing waits on the host
            elif series:
                ctxs[0].execute_async(args.algo, params)
            elif len(ctxs) == 1:
                tm = ctxs[0].execute(args.algo, params, timing=True)     # blocks until the stream is idle
                for s in tm["stages"]:
                    st_ms[s["name"]] = st_ms.get(s["name"], 0.0) + s["ms"]
                    st_meta[s["name"]] = s
            else:                                   # independent clips: one stream each, enqueued back to back
                for ctx in ctxs:
                    ctx.execute_async(args.algo, params)
                for ctx in ctxs:
                    ctx.synchronize()
        if series:
            ctxs[0].synchronize()
        if sharded:
            shard.synchronize()
        own = time.perf_counter() - t0           # this rank's own K steps (before the closing barrier)
        barrier()
        dt = time.perf_counter() - t0
        if series:
            tm = ctxs[0].timing_series_end()
            assert tm["steps"] == args.steps, tm["steps"]
            for s in tm["stages"]:
                st_ms[s["name"]] = s["ms"] * args.steps
                st_meta[s["name"]] = s
        if dist is not None:                        # the slowest rank's time; every rank's own time beside it
            t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            each = [None] * world
            dist.all_gather_object(each, round(own / max(args.steps, 1) * 1e3, 4))
            per_rank_ms.append(each)
        return dt, st_ms, st_meta

    regions = [timed_region() for _ in range(max(args.series, 1))]
    all_elapsed = [r[0] for r in regions]
    median_at = sorted(range(len(regions)), key=lambda i: regions[i][0])[(len(regions) - 1) // 2]
    elapsed, stage_ms, stage_meta = regions[median_at]                                               # the median series
    if len(ctxs) > 1:                           # per-stage device times of one clip, outside the timed region
        for _ in range(args.steps):
            tm = ctxs[0].execute(args.algo, params, timing=True)
            for s in tm["stages"]:
                stage_ms[s["name"]] = stage_ms.get(s["name"], 0.0) + s["ms"]
                stage_meta[s["name"]] = s
    verified = None
    if sharded:
        # per-stage device times of rank 0's range (outside the timed region; what the steps add on top is the border exchange)
        if shard.ctx is not None and rank == 0:
            for _ in range(args.steps):
                tm = shard.ctx.execute_extended_range(params, shard.first, shard.count, timing=True)
                for s in tm["stages"]:
                    stage_ms[s["name"]] = stage_ms.get(s["name"], 0.0) + s["ms"]
                    stage_meta[s["name"]] = s
        # the sharded result, gathered on the root, against the root's own single-GPU separation of the whole clip
        shard.step()
        got = shard.gather(0)
        if rank == 0:
            ref = repet.Context(local_rank)
            ref.upload(clip)
            ref.execute("extended", params)
            want = ref.download()
            ref.close()
            worst = float(np.max(np.abs(got - want)))
            verified = {"ok": bool(np.all(np.isfinite(got)) and worst <= 2e-6), "max_abs_difference": worst, "tolerance": 2e-6,
                        "how": "all ranks' owned samples gathered on the root vs the root's own single-GPU repet.extended of the whole clip "
                               "(a border sample is the fp32 sum of two ranks' rounded products where one GPU uses a fused multiply-add)"}
        ctx = shard.ctx
    else:
        out = ctxs[-1].download()
        assert out.shape[-2:] == clip.shape and np.all(np.isfinite(out)), "separation produced non-finite samples"
        if dist is not None:
            # every rank checks one of its own clips against a fresh single call on its device: bit for bit
            probe = ctxs[-1]
            one = out if out.ndim == 2 else out[-1]
            repet.set_device(local_rank)
            same = bool(np.array_equal(one, getattr(repet, args.algo)(clips[-1], fs)))
            flags = [None] * world
            dist.all_gather_object(flags, same)
            verified = {"ok": all(flags), "per_rank": flags,
                        "how": "the last resident clip of every rank == repet.%s of the same array on that rank's GPU, bit for bit" % args.algo}
    want_scatter = not args.no_scatter and args.config == 2 and example_clip is None
    line = None

    if rank == 0:
        steps = max(args.steps, 1)                   # stage figures are per clip
        T, F, C = int(ctx.last_frame_count()) if ctx is not None else 0, params.window_length // 2 + 1, channels
        rank_path = any("rank_columns" in name for name in stage_ms)
        sim_like = args.algo in ("sim", "simonline")
        k_mean = net_size = net_instr = None
        if sim_like:
            rows = T if args.algo == "sim" else max(T - params.buffer_frames + 1, 0)
            _, cnt = ctxs[0].last_sim_indices(rows, params.sim_number)       # outside the timed region
            k_mean = float(np.mean(cnt)) if rows else 0.0
            span = T if args.algo == "sim" else params.buffer_frames
            bound = min(params.sim_number, -(-span // (params.sim_distance_frames + 1)))    # what the engine sizes the network for
            ns, ni = ctypes.c_int32(), ctypes.c_int32()
            _native.lib().repet_median_network_info(int(bound), ctypes.byref(ns), ctypes.byref(ni))
            net_size, net_instr = ns.value, ni.value
        stages = []
        for name, total in stage_ms.items():
            ms = total / steps
            meta = stage_meta[name]
            sec = ms * 1e-3
            entry = {"name": name, "ms": round(ms, 4)}
            if name == "mask_sim_select":
                # the bit-sliced selection (mask_bits.hip): per frame and list entry one 256-byte row of every code plane,
                # gathered through L2 / Infinity Cache -- what bounds it is that gather path, not the ~9 400 boolean wave
                # instructions per frame (tools/microbench/bitslice_select.hip: same time with the lists in a 1 MB window,
                # 20 % more with uniformly random lists)
                planes = max(int(T - 1).bit_length(), 11)
                gathered = 256.0 * planes * (k_mean + 1.0) * rows
                ni = ctypes.c_int32()
                _native.lib().repet_median_network_info(-int(bound), None, ctypes.byref(ni))      # (negative bound: the bit-sliced form)
                instr = ni.value * planes * rows if ni.value > 0 else None
                ach = gathered / sec / 1e9
                entry.update({"bound": "l2", "achieved": round(ach, 1), "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s (gathered plane rows)",
                              "frac": round(ach / L2_GATHER_PEAK_GBS, 4), "algorithmic": gathered,
                              "peak_note": "L2-served row gathers, 16.8-18.8 TB/s chip-wide (MI355X_MICROARCH.md 'Indexed rows: gather into LDS')",
                              "planes": planes, "k_mean": round(k_mean, 2)})
                if instr:
                    gi = instr / sec / 1e9
                    entry["valu_view"] = {"wave_instructions": instr, "achieved": round(gi, 1), "peak": VALU_FULL_RATE_GINSTR, "unit": "G wave-instr/s",
                                          "frac": round(gi / VALU_FULL_RATE_GINSTR, 4),
                                          "note": "v_bitop3_b32 is in the fast VALU class (2.6 cycles at 8 waves per SIMD, profiles/r04_valu_rate.txt)"}
                b8d = 4.0 * F * rows * C * (1.0 + k_mean + 1.0)
                entry["survey_8d_bytes"] = {"algorithmic": b8d, "note": "SURVEY 8d prices K5 as one stage: see mask_sim"}
            elif name == "mask_sim" and "mask_sim_select" in stage_ms:
                # lookups of the two middle values in the sorted columns (L2-resident by construction: one block of columns per
                # XCD at a time) + the mask: V and the code word in, X masked in place
                ach = meta["bytes"] / sec / 1e9
                entry.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                              "algorithmic": meta["bytes"],
                              "note": "streams V and the selected codes, writes the mask plane (or masks X in place); the scattered 4-byte table reads (two per cell that needs them, a 128-byte "
                                      "L2 line each) are what it waits for"})
                both = sec + stage_ms["mask_sim_select"] / steps * 1e-3
                b8d = 4.0 * F * rows * C * (1.0 + k_mean + 1.0)
                entry["survey_8d_bytes"] = {"algorithmic": b8d, "GB/s": round(b8d / both / 1e9, 1), "frac_of_hbm_peak": round(b8d / both / 1e9 / HBM_PEAK_GBS, 4),
                                            "k_mean": round(k_mean, 2), "ms": round(both * 1e3, 4),
                                            "note": "SURVEY 8d's byte view of K5 over BOTH kernels of the median (selection + lookups): cache-served gathers, "
                                                    "and 13 bits per gathered value instead of 32"}
            elif name.startswith("mask_sim") and net_instr:
                # VALU-issue-bound: one selection network per wave and block of bins. A wave covers 64 bins
                # on the float path, 128 (two 16-bit rank codes per lane) on the rank path; v_min/v_max-class
                # instructions issue at one per 4 cycles per SIMD (1 024 SIMDs, 2.4 GHz max clock).
                bins_per_wave = 128 if rank_path else 64
                waves = (rows * C * ((F - 1) // bins_per_wave) + -(-rows // 64) * C) * (args.clips if batched else 1)   # + one-lane-per-frame Nyquist kernel
                ach = net_instr * waves / sec / 1e9
                entry.update({"bound": "valu", "achieved": round(ach, 1), "peak": VALU_QUARTER_RATE_GINSTR, "unit": "G wave-instr/s",
                              "frac": round(ach / VALU_QUARTER_RATE_GINSTR, 4), "algorithmic": net_instr * waves,
                              "peak_note": "measured issue rate of the v_min / v_max / v_pk_min_u16 class: one per 4 cycles per SIMD (profiles/r03_valu_rate.txt)",
                              "frac_vs_full_rate_valu": round(ach / VALU_FULL_RATE_GINSTR, 4), "full_rate_valu_peak": VALU_FULL_RATE_GINSTR,
                              "network": {"wires": net_size, "instructions": net_instr, "waves": waves, "codes": "2 x u16 per lane" if rank_path else "f32"}})
                # the byte view SURVEY 8d defines for K5 (reads 4FTC + gathers 4FTC*Kmean, writes 4FTC), beside it:
                # cache-resident gathers, so this is NOT the roof that binds
                b8d = 4.0 * F * rows * C * (1.0 + k_mean + 1.0) * (args.clips if batched else 1)
                entry["survey_8d_bytes"] = {"algorithmic": b8d, "GB/s": round(b8d / sec / 1e9, 1), "frac_of_hbm_peak": round(b8d / sec / 1e9 / HBM_PEAK_GBS, 4),
                                            "k_mean": round(k_mean, 2), "note": "gathers are served by L2 / Infinity Cache" + ("; the rank path moves 2 bytes per gathered value, not 4" if rank_path else "")}
                # short lists (simonline: about ten similar frames): the network is a few dozen instructions and the stage is
                # bound by what MUST cross HBM -- V read once (the gathers re-read it from cache), X read and written
                # (the engine's figure = V + gathers + what it writes: 16 bytes with X masked in place, 4 with the mask as a plane)
                compulsory = meta["bytes"] - 4.0 * params.sim_number * F * rows * C * (args.clips if batched else 1)
                if compulsory / (HBM_PEAK_GBS * 1e9) > net_instr * waves / (VALU_QUARTER_RATE_GINSTR * 1e9):
                    entry["valu_view"] = {k: entry[k] for k in ("achieved", "peak", "unit", "frac")}
                    ach = compulsory / sec / 1e9
                    entry.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic": compulsory,
                                  "note": "compulsory bytes: V once + the mask plane (or X in place); the network is too short to bind"})
            elif meta["flops"] > 0 and "f16x3" in name:
                # f16-split matrix-core kernel: three f16 MFMA products per fp32 term, fp32 accumulate. Priced on EXECUTED
                # f16 flops against the dense f16 MFMA peak; the fp32-equivalent algorithmic rate is stated beside it.
                ach = meta["flops"] / sec / 1e12 if "similarity_gemm" in name else 3.0 * meta["flops"] / sec / 1e12
                executed = meta["flops"] if "similarity_gemm" in name else 3.0 * meta["flops"]
                alg = 2.0 * F * float(T) * T if "similarity_gemm" in name else meta["flops"]
                entry.update({"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TF,
                              "unit": "TFLOP/s (f16, executed)" if "similarity_gemm" in name else "TFLOP/s (f16, 3 products per algorithmic term)",
                              "frac": round(ach / MFMA_F16_PEAK_TF, 4), "algorithmic": executed,
                              "fp32_equivalent": {"flops": alg, "TFLOP/s": round(alg / sec / 1e12, 1), "note": "2*F*T^2 (symmetric half skipped), not comparable with a peak"}})
            elif meta["flops"] > 0 and meta["flops"] / (MFMA_F32_PEAK_TF * 1e12) > meta["bytes"] / (HBM_PEAK_GBS * 1e9):
                ach = meta["flops"] / sec / 1e12
                entry.update({"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s (fp32)",
                              "frac": round(ach / MFMA_F32_PEAK_TF, 4), "algorithmic": meta["flops"]})
            else:
                ach = meta["bytes"] / sec / 1e9
                entry.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic": meta["bytes"]})
                if name == "rank_columns":
                    entry["note"] = "three kernels (transpose, per-column sort + rank search in LDS, transpose back); the sort is LDS/VALU-bound"
                if name == "peaks+rank_columns":
                    entry["note"] = ("two independent launches side by side on two streams: peak picking (one wavefront per row of S, instruction- and "
                                     "latency-bound) and the column sort of V (transpose, per-column sort + rank search in LDS, transpose back); "
                                     "bytes = S read once + the sort's passes, time = both")
            stages.append(entry)
        # the dominant KERNEL: "peaks+rank_columns" is two chains of seven launches side by side on two streams, not a kernel --
        # the line names both: the longest stage by time (dominant_stage) and the longest single-kernel stage (kernel, priced)
        single = [s for s in stages if s["name"] != "peaks+rank_columns"] or stages
        dom = max(single, key=lambda s: s["ms"])
        longest = max(stages, key=lambda s: s["ms"])
        roof = {"kernel": dom["name"], "dominant_stage": {"name": longest["name"], "ms": longest["ms"],
                                                           "is_a_single_kernel": longest["name"] != "peaks+rank_columns"}}
        roof.update({k: dom[k] for k in ("bound", "achieved", "peak", "unit", "frac") if k in dom})
        roof["traffic"] = None
        # HBM-side bytes per launch of the dominant kernel: NOT measured by this run -- read from the committed PMC pass of
        # this same workload and build (profiles/), if there is one for it
        try:
            if args.config == 2 and args.algo == "sim" and args.duration == 180.0:
                with open(os.path.join(ROOT, "profiles", PMC_FILE)) as fh:
                    doc = json.load(fh)
                pmc = doc["stages"][dom["name"].replace("_f16x3", "")]
                import hashlib
                with open(_native.LIB_PATH, "rb") as fh:
                    lib_sha = hashlib.sha256(fh.read()).hexdigest()
                roof["traffic"] = pmc["hbm_bytes_per_launch"]
                roof["traffic_source"] = f"profiles/{PMC_FILE} (committed rocprofv3 --pmc pass of this workload; static, not measured in this run); " \
                                         "FETCH_SIZE*1024*k + WRITE_SIZE*1024 per launch, fetch correction " + pmc["fetch_correction"]
                roof["traffic_library_sha256"] = doc.get("library_sha256")
                roof["traffic_is_of_this_library"] = doc.get("library_sha256") == lib_sha
        except (OSError, KeyError, ValueError):
            pass
        roof["ms_per_launch"] = dom["ms"]
        roof["algorithmic_per_launch"] = dom.get("algorithmic")
        for extra in ("network", "survey_8d_bytes", "fp32_equivalent", "frac_vs_full_rate_valu", "full_rate_valu_peak", "peak_note", "valu_view", "planes", "k_mean"):
            if extra in dom:
                roof[extra] = dom[extra]
        line = {
            "metric": f"audio-seconds/sec (x real-time) for repet.{args.algo}, {fs / 1000:g} kHz {'stereo' if channels == 2 else str(channels) + '-ch'}",
            "value": round(args.duration * args.clips * args.steps * (1 if sharded else world) / elapsed, 2),
            "unit": "audio-seconds/sec",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None and backend == "nccl" else None), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3),
            "series_ms_per_step": [round(e / max(args.steps, 1) * 1e3, 3) for e in all_elapsed],
            "ms_per_step_min": round(min(all_elapsed) / max(args.steps, 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "backend": (backend if dist is not None else None), "ranks_share_gpus": bool(ranks_share_gpus),
            "per_rank_ms_per_step": (per_rank_ms[median_at] if per_rank_ms else None),
            "dtype": DTYPE_NOTE.get(args.algo, "f32"), "data": "synthetic" if example_clip is None else "the reference's example clip (audio_file.wav, read in place)",
            "config": {"workload": (f"repet.extended on ONE {args.duration:g}-s {fs / 1000:g} kHz {channels}-ch synthetic clip, its {shard.n_segments} segments in {world} contiguous ranges "
                                    f"(BASELINE.json configs[2]), every rank's window resident in HBM" if sharded else
                                    f"repet.{args.algo} on {args.clips} x {args.duration:g}-s {fs / 1000:g} kHz {channels}-ch {'synthetic' if example_clip is None else 'example'} clip(s) per GPU "
                                    f"(BASELINE.json configs[{args.config - 1}]), clips resident in HBM"),
                       "clips_per_step": 1 if sharded else world * args.clips, "samples_per_clip": int(clip.shape[0]), "channels": channels,
                       "frames": T, "prewarm_ms": args.prewarm_ms,
                       "series": f"{len(all_elapsed)} timed regions of {args.steps} steps; value and ms_per_step are the median region",
                       "parallelism": (f"segment ranges {shard.ranges} over {world} ranks; per step {len(shard.moves)} border exchange(s) of "
                                       f"{[hi - lo for _, _, lo, hi in shard.moves][:1]} samples x {channels} ch fp32, point-to-point" if sharded else
                                       f"clip-parallel x{world}, no collective" + (", clips batched through every stage" if batched else ""))},
            "roofline": roof,
            "stages": stages,
            "device_ms_per_step": round(sum(s["ms"] for s in stages), 4),
        }
        if verified is not None:
            line["verified"] = verified
        if sharded:
            line["stages_note"] = "stage times are rank 0's range alone (outside the timed region); ms_per_step minus their sum is the border exchange + enqueue"
        if sim_like:
            ex = ctxs[0].last_exact_stats()
            line["peak_picking_second_level"] = {
                "rows": ex["rows_exact"], "of": rows, "rows_fast_path": ex["rows_fast_path"], "float64_unit_rows": ex["unit_rows_f64"],
                "largest_level1_minus_level2": ex["level2_max_diff"], "input_has_remainders": ex["input_has_remainders"],
                "note": "rows whose float64 verdicts on the fp32 spectra are closer than 2.5e-7 are decided again from float64 spectra "
                        "(inside the peak-picking stage's time); REPET_PEAK_EXACT=0 turns it off"}
        if world == 1:
            # PCIe-inclusive drop-in call (float64 NumPy in host RAM -> float64 NumPy out): reported beside `value`,
            # never as `value`. Includes upload, f64->f32, the run, f32->f64 and download. Twice: the synthetic
            # clip as it is (float64 noise: the fp32 remainders of the samples travel too, for the float64 spectra of the
            # peak picking's second level) and rounded to 16-bit PCM values (what wavread yields: no remainders).
            repet.set_device(local_rank)

            def drop_in(x):
                wall = []
                for _ in range(4):
                    t1 = time.perf_counter()
                    getattr(repet, args.algo)(x, fs)
                    wall.append(time.perf_counter() - t1)
                return {"value": round(args.duration / min(wall[1:]), 1), "unit": "audio-seconds/sec",
                        "ms_min": round(min(wall[1:]) * 1e3, 2), "ms_median": round(sorted(wall[1:])[1] * 1e3, 2)}
            # the floor of such a call: the fp32 samples in and the fp32 result out at PCIe Gen5 x16's 63 GB/s (MI355X_MICROARCH.md)
            # + the device step; the remainder plane of a float64 clip travels beside the computation and is not counted
            floor_ms = 2 * clip.size * 4 / 63e9 * 1e3 + elapsed / max(args.steps, 1) * 1e3

            def with_floor(entry):
                entry["pcie_floor_ms"] = round(floor_ms, 2)
                entry["frac_of_pcie_floor"] = round(floor_ms / entry["ms_min"], 3)
                return entry
            # (where the call's host work runs matters on a two-socket host: started under `taskset -c <the GPU node's CPUs>` the
            # same call took 3.84-3.94 ms PCM-exact / 4.7-5.2 ms float64 on the box of profiles/r05_dropin_numa.txt)
            line["host_affinity"] = ({"bound_to_gpu_numa_node": True, "cpus": len(numa[1]), "of": len(numa[0]),
                                      "note": "bench.py binds itself to the CPUs of its GPU's NUMA node before it touches the GPU (numactl --cpunodebind "
                                              "for a GPU job; REPET_BENCH_NUMA=0 turns it off); the CPU baseline ran before that, on every CPU the process was given"}
                                     if numa else {"bound_to_gpu_numa_node": False, "cpus": len(os.sched_getaffinity(0))})
            line["array_in_array_out"] = with_floor(drop_in(clip))
            line["array_in_array_out"]["note"] = ("repet.%s(audio_signal, fs) wall time: float64 NumPy in host RAM -> float64 NumPy out (host threads "
                                                  "narrow/widen through a pinned ring; fp32 samples over PCIe, their fp32 remainders behind them beside the computation)" % args.algo)
            if example_clip is None:
                pcm_clip = np.round(clip * 32768.0).clip(-32768, 32767) / 32768.0
                line["array_in_array_out_pcm16"] = with_floor(drop_in(pcm_clip))
                line["array_in_array_out_pcm16"]["note"] = "the same call on the clip rounded to 16-bit PCM values (float64 array, exact in fp32: no remainders travel)"
        if world == 1 and args.config == 2 and not args.no_variants and example_clip is None and "peaks+rank_columns" in stage_ms:
            # "peaks+rank_columns" is two chains of kernels side by side on two streams. A child run with REPET_RANK_OVERLAP=0 puts
            # the sort BEHIND the peak picking on the main stream with a timing mark behind every kernel: live per-kernel times,
            # each priced on what binds THAT kernel (SURVEY 8d's 4 T^2 bytes for this stage are not read any more: the peak
            # picking works from the Gram epilogue's segment records).
            import subprocess
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 200)), "--warmup", str(args.warmup),
                                        "--series", "1", "--no-cpu-baseline", "--no-scatter", "--no-variants"],
                                       env=dict(os.environ, REPET_RANK_OVERLAP="0", REPET_BENCH_NUMA="0"), capture_output=True, text=True, timeout=600)
                cj = json.loads(child.stdout.strip().splitlines()[-1])
                alone = {st["name"]: st for st in cj["stages"]}
                n_sort = 1 << max(int(T - 1).bit_length(), 11)
                log_n = n_sort.bit_length() - 1
                columns = (F - 1) * C
                # bitonic sort of 2^L keys: 2^(L-1) L (L+1) / 2 compare-exchanges of two instructions (min, max) each; then T binary
                # searches of L probes (read, compare, select): all on the quarter-rate min / max / compare class
                sort_instr = columns * (n_sort // 2 * log_n * (log_n + 1) // 2 * 2 + T * log_n * 3) / 64.0
                rows_k = []
                for nm, st in alone.items():
                    if nm not in ("local_maxima_pass1", "local_maxima_level2", "columns_from_rows", "rank_columns_sort", "code_planes", "rows_from_codes"):
                        continue
                    sec_k = st["ms"] * 1e-3
                    row = {"name": nm, "ms_alone": st["ms"]}
                    if nm == "rank_columns_sort":
                        gi = sort_instr / sec_k / 1e9
                        row.update({"bound": "valu", "achieved": round(gi, 1), "peak": VALU_QUARTER_RATE_GINSTR, "unit": "G wave-instr/s", "frac": round(gi / VALU_QUARTER_RATE_GINSTR, 4),
                                    "algorithmic": sort_instr, "note": f"{columns} columns x bitonic sort of {n_sort} keys (min + max per compare-exchange) + {T} rank searches of {log_n} probes"})
                    elif nm == "local_maxima_pass1":
                        # what it must move: the segment records, the lists it writes, and the lines of S its cut segments and near-ties
                        # ask for (about 60 x 128 B per row: profiles/ PMC pass) -- a latency-bound kernel (one wavefront per row, its span
                        # is its slowest rows), priced against HBM for the record
                        moved = st["algorithmic"] + 60.0 * 128.0 * T if "algorithmic" in st else None
                        if moved:
                            row.update({"bound": "latency", "achieved": round(moved / sec_k / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(moved / sec_k / 1e9 / HBM_PEAK_GBS, 4),
                                        "algorithmic": moved, "rows_per_us": round(T / (sec_k * 1e6), 1),
                                        "note": "one wavefront per row of S, memory round trips per row; bytes = records + lists + ~60 lines of S per row"})
                    elif nm == "local_maxima_level2":
                        row.update({"bound": "latency", "note": "float64 unit rows of the queued frames (256-thread workgroup each: float64 FFT in LDS) + the recorded rows again + the general kernel"})
                    elif "algorithmic" in st:
                        gb = st["algorithmic"] / sec_k / 1e9
                        row.update({"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBS, 4), "algorithmic": st["algorithmic"]})
                    rows_k.append(row)
                for st in stages:
                    if st["name"] == "peaks+rank_columns":
                        st.pop("bound", None); st.pop("achieved", None); st.pop("peak", None); st.pop("unit", None); st.pop("frac", None); st.pop("algorithmic", None)
                        st["kernels"] = rows_k
                        st["kernels_note"] = ("per-kernel times of a child run with REPET_RANK_OVERLAP=0 (the sort behind the peak picking on ONE stream, a timing mark "
                                              "behind every kernel); in the timed runs the two chains overlap on two streams: ms above is both")
                        st["chains_alone_ms"] = {"peaks": round(sum(r["ms_alone"] for r in rows_k if r["name"].startswith("local_maxima")), 4),
                                                 "sort": round(sum(r["ms_alone"] for r in rows_k if not r["name"].startswith("local_maxima")), 4)}
                        # What bounds the STAGE (round 6): the chip's LDS. Every kernel of both chains holds its workgroups' LDS for
                        # their lifetime, and the two chains together ask for more LDS x time than 256 CUs x 160 KB offer in the
                        # time of the longer chain: the stage's floor is the sum, over its kernels, of (time alone) x (share of the
                        # chip's LDS its resident workgroups hold while it runs alone). The shares from the launch geometry
                        # (rank.hip, peaks_wave.hip, peaks_exact.hip); the first peak pass is a tail of fewer and fewer rows, its
                        # share is taken at the mean concurrency of its rows (profiles/r05_peak_gram_spans.txt: 1 749 of 3 840).
                        cu_lds, n_cu = 160 * 1024, 256
                        ts_ = -(-T // 64) * 64
                        seg_pitch = -(-(-(-ts_ // 32)) // 4) * 4
                        cap = -(-(T // (params.sim_distance_frames + 1) + 2) // 4) * 4
                        pass1_lds = max(324, -(-(5 * seg_pitch * 4) // 16)) * 16 + cap * 8 + 96 * (8 + 8 + 4 + 4 + 4 + 6 + 2) + 16
                        sort_lds = (n_sort + n_sort // 8) * 4 + (n_sort >> 4) * 4
                        shares = {"local_maxima_pass1": min(cu_lds // pass1_lds, 16) * pass1_lds / cu_lds * (1749.0 / 3840.0),
                                  # float64 unit rows: one 40-KB workgroup per queued frame, all resident at once, for two thirds of the
                                  # second level's time; its lite and general kernels (the other third) hold a quarter of the LDS
                                  "local_maxima_level2": (2.0 * min(ex["unit_rows_f64"] * 40960.0 / (n_cu * cu_lds), 1.0) + 0.25) / 3.0,
                                  "columns_from_rows": 8 * 32 * 65 * 4 / cu_lds,              # eight 8.3-KB workgroups per CU (wave slots)
                                  "rank_columns_sort": (cu_lds // sort_lds) * sort_lds / cu_lds,
                                  "code_planes": 2 * 1024 * 34 * 2 / cu_lds}                  # two 70-KB workgroups per CU
                        view = [{"name": r["name"], "ms_alone": r["ms_alone"], "lds_share_when_alone": round(shares[r["name"]], 3),
                                 "lds_ms": round(r["ms_alone"] * shares[r["name"]], 4)} for r in rows_k if r["name"] in shares]
                        floor = sum(v["lds_ms"] for v in view)
                        st.update({"bound": "lds_capacity", "achieved": round(floor, 4), "peak": st["ms"], "unit": "ms of the whole chip's LDS (256 CUs x 160 KB)",
                                   "frac": round(floor / st["ms"], 4), "lds_time": view,
                                   "note_bound": "floor = sum over the stage's kernels of (ms alone) x (share of the chip's LDS held while alone); "
                                                 "frac = floor / stage time = how tightly the two chains pack the LDS (DESIGN.md 8.2)"})
            except Exception as exc:  # noqa: BLE001 -- a breakdown must never cost the headline
                for st in stages:
                    if st["name"] == "peaks+rank_columns":
                        st["kernels_error"] = f"{type(exc).__name__}: {exc}"
            # the longest stage's own bound beside the longest kernel's (roofline.dominant_stage)
            for st in stages:
                if st["name"] == roof["dominant_stage"]["name"] and "frac" in st and not roof["dominant_stage"]["is_a_single_kernel"]:
                    roof["dominant_stage"].update({k: st[k] for k in ("bound", "achieved", "peak", "unit", "frac", "chains_alone_ms") if k in st})
        if world == 1 and args.config == 2 and not args.no_variants and example_clip is None and args.clips == 1:
            # A GPU that serves INDEPENDENT clips can keep several in flight: three contexts (three streams, three resident clips)
            # fill the latency-bound stages of one another. Beside the headline, never instead of it (the metric is quoted on one clip).
            import subprocess
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--clips", "3", "--steps", str(min(args.steps, 100)), "--warmup", str(args.warmup),
                                        "--series", "3", "--no-cpu-baseline", "--no-scatter", "--no-variants"],
                                       env=dict(os.environ, REPET_BENCH_NUMA="0"), capture_output=True, text=True, timeout=600)
                cj = json.loads(child.stdout.strip().splitlines()[-1])
                line["three_clips_in_flight"] = {"value": cj["value"], "unit": "audio-seconds/sec", "ms_per_clip": round(cj["ms_per_step"] / 3.0, 4),
                                                 "note": "three contexts, one clip and stream each, a step = the three separations enqueued together and awaited"}
            except Exception as exc:  # noqa: BLE001
                line["three_clips_in_flight"] = {"error": f"{type(exc).__name__}: {exc}"}
        if world == 1 and args.config == 2 and not args.no_variants and example_clip is None:
            # north_star names an fp32 MFMA GEMM for the similarity matrix; the default is the f16x3 split of the fp32 operands.
            # The exact-fp32 kernel (REPET_GRAM=f32, read once per process) is timed by a child run of this script.
            import subprocess
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup),
                                        "--series", "1", "--no-cpu-baseline", "--no-scatter", "--no-variants"],     # (one timed region: a few seconds)
                                       env=dict(os.environ, REPET_GRAM="f32", REPET_BENCH_NUMA="0"), capture_output=True, text=True, timeout=600)
                cj = json.loads(child.stdout.strip().splitlines()[-1])
                gemm = [st for st in cj["stages"] if st["name"].startswith("similarity_gemm")][0]
                line["fp32_gemm_variant"] = {"ms_per_step": cj["ms_per_step"], "gemm_ms": gemm["ms"], "gemm_TFLOP/s_fp32": gemm.get("achieved"),
                                             "gemm_frac_of_fp32_mfma_peak": gemm.get("frac"), "value": cj["value"],
                                             "note": "REPET_GRAM=f32: v_mfma_f32_32x32x2_f32, exact fp32 k-ordered accumulation (gram.hip)"}
            except Exception as exc:  # noqa: BLE001 -- a variant figure must never cost the headline
                line["fp32_gemm_variant"] = {"error": f"{type(exc).__name__}: {exc}"}
        if cpu_line is not None:
            line["cpu_baseline"] = cpu_line

    # The multi-GPU data path (scatter -> separate -> gather over RCCL point-to-point) runs LAST and under a watchdog: the
    # headline above must reach stdout whatever that leg does on a node it has never run on. A time-out or an exception on
    # any rank ends every rank at once, rank 0 with the line (and the reason under "scatter_gather").
    if want_scatter:
        import threading
        import traceback
        once = threading.Lock()                      # the watchdog thread and the main thread: exactly one of them prints

        def bail(reason, trace=None):
            # every rank says why on stderr; rank 0 still prints the headline (with the reason under "scatter_gather"); the
            # exit code is 1 -- a broken multi-GPU data path must not look like success -- unless --scatter-lenient asks for
            # 0 (the headline above is complete either way)
            sys.stderr.write(f"[bench rank {rank}] scatter/gather leg gave up: {reason}\n" + (trace or ""))
            sys.stderr.flush()
            if not once.acquire(blocking=False):
                return
            if rank == 0:
                line["scatter_gather"] = {"error": reason}
                sys.stdout.write("\n" + json.dumps(line) + "\n")
                sys.stdout.flush()
            os._exit(0 if args.scatter_lenient else 1)

        watchdog = threading.Timer(args.scatter_limit, bail, args=(f"no result within {args.scatter_limit:g} s",))
        watchdog.daemon = True
        watchdog.start()
        try:
            for ctx_ in ctxs[1:]:
                ctx_.close()
            scatter = scatter_gather_leg(dist, rank, world, local_rank, args.algo, fs, channels, args.duration, 8)
            if dist is not None:
                dist.barrier()
        except BaseException as exc:                 # noqa: BLE001 -- the other ranks wait in a collective: end them all
            bail(f"{type(exc).__name__}: {exc}", traceback.format_exc())
        watchdog.cancel()
        if not once.acquire(blocking=False):         # the watchdog fired while the leg was finishing: it owns the output
            time.sleep(30)                           # (it ends the process itself; if it has not by then, this does)
            os._exit(1)
        if rank == 0 and scatter is not None:
            line["scatter_gather"] = scatter
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        sys.stdout.write("\n" + json.dumps(line) + "\n")    # the ONE line, on a line of its own whatever a library left unfinished
        sys.stdout.flush()
        # a multi-rank result that does not equal the single-GPU one must not look like success
        bad = [k for k in ("verified", "scatter_gather") if isinstance(line.get(k), dict) and
               (line[k].get("ok") is False or line[k].get("verified") is False)]
        if bad:
            sys.stderr.write(f"[bench] verification failed: {bad}\n")
            raise SystemExit(1)


if __name__ == "__main__":
    main()
