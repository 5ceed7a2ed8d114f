#!/bin/bash
# round 4, first GPU contact of the segment-record peak picking: targeted parity tests, a bench line, kernel stats
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/r04a"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "local_maxima or indices" > "$out/t1.log" 2>&1; echo "t1 rc $?" 
tail -5 "$out/t1.log"
timeout 1500 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "segment_record or second_level or similar_frame or gram_tile" > "$out/t2.log" 2>&1; echo "t2 rc $?"
tail -5 "$out/t2.log"
timeout 600 python3 bench.py --no-variants --no-scatter --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"
python3 - <<'P'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04a/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], [(s["name"], s["ms"]) for s in d["stages"]])
P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > "$out/prof.log" 2>&1
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv" && head -16 "$out/kernel_stats.csv" | cut -c1-150
rm -rf "$out/prof"
