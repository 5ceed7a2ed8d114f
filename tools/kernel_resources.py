#!/usr/bin/env python3
"""Registers / LDS of every kernel in librepet_hip.so (from the code objects' metadata notes).
usage: tools/kernel_resources.py [substring]"""
import os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "repet-python_amd", "lib", "librepet_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
want = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as tmp:
    shutil.copy(LIB, tmp)
    subprocess.check_call([OBJDUMP, "--offloading", "librepet_hip.so"], cwd=tmp, stdout=subprocess.DEVNULL)
    rows = []
    for b in sorted(f for f in os.listdir(tmp) if "amdgcn" in f):
        notes = subprocess.check_output([OBJDUMP.replace("objdump", "readelf"), "--notes", os.path.join(tmp, b)], text=True)
        for block in notes.split("- .agpr_count:")[1:]:
            f = dict(re.findall(r"\.(name|vgpr_count|sgpr_count|group_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s+(\S+)", block))
            agpr = block.split()[0]
            name = subprocess.run(["c++filt", f.get("name", "?")], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
            if want in name:
                rows.append((name[-70:], f.get("vgpr_count"), agpr, f.get("sgpr_count"), f.get("group_segment_fixed_size"), f.get("sgpr_spill_count")))
    print(f"{'kernel':70s} vgpr agpr sgpr  lds(static) sgpr_spills")
    for r in sorted(set(rows)):
        print(f"{r[0]:70s} {r[1]:>4} {r[2]:>4} {r[3]:>4} {r[4]:>8} {r[5]:>4}")
