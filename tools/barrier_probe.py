import os, time, sys
sys.path[:0] = ["repet-python_amd", "."]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import repet
from repet_synth import synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = repet.Context(0); ctx.upload(synth(180, 44100, 2, 0)); p = repet.derive_params(44100)
for _ in range(300): ctx.execute("sim", p)
def bar():
    dist.barrier(); torch.cuda.synchronize()
bar()
for K in (10, 20, 50):
    for mode in ("nccl-barrier", "sync-only"):
        bar(); t0 = time.perf_counter()
        for _ in range(K): ctx.execute_async("sim", p)
        ctx.synchronize()
        if mode == "nccl-barrier": bar()
        else: torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(K, mode, round(t / K * 1e3, 4), "ms/step")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("barrier alone", round((t1 - t0) * 1e3, 3), "ms, + sync", round((t2 - t1) * 1e3, 3))
g = dist.new_group(backend="gloo")
for _ in range(3):
    t0 = time.perf_counter(); dist.barrier(group=g); t1 = time.perf_counter()
    print("gloo barrier", round((t1 - t0) * 1e3, 3), "ms")
