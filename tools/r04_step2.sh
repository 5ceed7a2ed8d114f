#!/bin/bash
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/r04b"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "local_maxima or indices" > "$out/t1.log" 2>&1; echo "t1 rc $?"
tail -3 "$out/t1.log"
timeout 900 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "segment_record or similar_frame or second_level or online" > "$out/t2.log" 2>&1; echo "t2 rc $?"
tail -3 "$out/t2.log"
PEAK_PHASES=1 timeout 600 python3 tools/peak_stamps.py > "$out/stamps.txt" 2>&1; echo "stamps rc $?"
grep -v "^  t = " "$out/stamps.txt" | grep -A8 "fastest half"
timeout 600 python3 bench.py --no-variants --no-scatter --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"
python3 - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], [(s["name"], s["ms"]) for s in d["stages"]])
P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > "$out/prof.log" 2>&1
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv" && head -12 "$out/kernel_stats.csv" | cut -c1-130
rm -rf "$out/prof"
