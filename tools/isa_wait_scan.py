#!/usr/bin/env python3
"""Static scan of the device code for loads that wait for each other: per kernel, vector-memory loads against
`s_waitcnt vmcnt(0)`, single-dword scalar loads against `s_waitcnt lgkmcnt(0)`, LDS reads against their waits.

A load inside a branch ("cond ? p[i] : 0") is waited for at the branch's join, and a scalar "k < n ? list[k] : pad" per
slot becomes one s_load_dword + wait per slot: unrolled loops of such loads run one memory round trip per element. This
found the median mask's gather (100 scalar round trips per wave), the two transposes of the column sort, the inverse
STFT's accumulate path and the peak kernel's rival search. Kernels near the top of the list deserve a look at their ISA.
usage: tools/isa_wait_scan.py [file.hip ...]   (default: every .hip of repet-python_amd/csrc)"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "repet-python_amd", "csrc")
files = sys.argv[1:] or sorted(glob.glob(os.path.join(SRC, "*.hip")))
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
         "-mllvm", "-pragma-unroll-threshold=131072", "--cuda-device-only", "-S"]
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for f in files:
        out = os.path.join(tmp, os.path.basename(f) + ".s")
        if subprocess.run(["/opt/rocm/bin/hipcc", *flags, f, "-o", out], cwd=SRC, capture_output=True).returncode != 0:
            print("could not compile", f, file=sys.stderr)
            continue
        txt = open(out).read()
        for name in re.findall(r"^(_Z[\w]+):", txt, re.M):
            i = txt.index("\n" + name + ":")
            j = txt.find(".end_amdhsa_kernel", i)
            if j < 0:
                continue
            body = txt[i:j]
            vm = len(re.findall(r"\t(global_load|buffer_load)", body))
            v0 = len(re.findall(r"s_waitcnt vmcnt\(0\)", body))
            sl = len(re.findall(r"\ts_load_dword\s", body))
            ds = len(re.findall(r"\tds_read", body))
            l0 = len(re.findall(r"lgkmcnt\(0\)", body))
            pretty = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            score = (v0 / vm if vm >= 8 else 0.0) + (1.0 if sl >= 10 else 0.0)
            rows.append((score, os.path.basename(f), pretty[-72:], vm, v0, sl, ds, l0))
print(f"{'file':16s} {'kernel':72s} {'vmem':>5s} {'vmcnt0':>6s} {'s_load1':>7s} {'ds_read':>7s} {'lgkm0':>6s}")
for r in sorted(rows, reverse=True)[:40]:
    if r[0] <= 0.2:
        break
    print(f"{r[1]:16s} {r[2]:72s} {r[3]:5d} {r[4]:6d} {r[5]:7d} {r[6]:7d} {r[7]:6d}")
