#!/bin/bash
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/r04d"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "local_maxima or indices" > "$out/t1.log" 2>&1; echo "t1 rc $?"
tail -3 "$out/t1.log"
timeout 1500 python3 -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "segment_record or similar_frame or second_level or online or resident_batch" > "$out/t2.log" 2>&1; echo "t2 rc $?"
tail -3 "$out/t2.log"
for split in 1 0; do
  export REPET_PEAK_SPLIT=$split REPET_RANK_OVERLAP=0
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof$split" -- python3 bench.py --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > "$out/prof$split.log" 2>&1
  f=$(find "$out/prof$split" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_split$split.csv"
  rm -rf "$out/prof$split"
done
unset REPET_RANK_OVERLAP
for split in 1 0 1 0; do
  export REPET_PEAK_SPLIT=$split
  timeout 600 python3 bench.py --no-variants --no-scatter --no-cpu-baseline --series 3 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap, split $split', d['ms_per_step'], [(s['name'], s['ms']) for s in d['stages']])"
done
