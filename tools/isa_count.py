#!/usr/bin/env python3
"""Static instruction counts (VALU / LDS / vector memory / total) of the kernels of an object file or the library whose names
contain a substring. usage: tools/isa_count.py <substring> [file]"""
import os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "repet-python_amd", "lib", "librepet_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
with tempfile.TemporaryDirectory() as tmp:
    base = os.path.basename(src)
    shutil.copy(src, tmp)
    subprocess.check_call([OBJDUMP, "--offloading", base], cwd=tmp, stdout=subprocess.DEVNULL)
    for b in sorted(f for f in os.listdir(tmp) if "amdgcn" in f):
        txt = subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp, b)], text=True)
        for p in re.split(r"\n(?=[0-9a-f]+ <)", txt):
            m = re.match(r"[0-9a-f]+ <([^>]+)>", p)
            if not m or want not in m.group(1):
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
            ops = [l.split()[0] for l in p.split("\n")[1:] if l.strip() and not l.strip().startswith("//")]
            c = lambda pre: sum(1 for o in ops if o.startswith(pre))
            print(f"{name[-60:]:60s} valu {c('v_'):6d}  lds {c('ds_'):5d}  vmem {c('global_') + c('buffer_') + c('flat_'):5d}  salu {c('s_'):6d}  total {len(ops):6d}")
