#!/bin/bash
# Per-kernel device times of one variant on the GPU box (rocprofv3 --kernel-trace --stats): prints the top kernels.
# usage: tools/kprof.sh [algo] [seconds] [fs] [channels]     (arguments of tools/stage_time.py)
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/kprof"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p" -- python3 tools/stage_time.py "$@" > "$out/run.log" 2>&1
f=$(find "$out/p" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/p"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
