#!/bin/bash
# Round-end measurement on the GPU box: bench lines for cfg 2-5 and the rocprofv3 kernel-trace summaries of the
# same commands, written under gpurun_out/round/ (copy what should be judged into profiles/).
# usage: tools/round_profile.sh
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/round"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 bench.py > "$out/bench_cfg2.json" 2> "$out/bench_cfg2.err"
for cfg in 3 4 5; do
  timeout 900 python3 bench.py --config $cfg > "$out/bench_cfg$cfg.json" 2> "$out/bench_cfg$cfg.err"
done
for cfg in 2 3 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_cfg$cfg" -- python3 bench.py --config $cfg --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > "$out/prof_cfg$cfg.log" 2>&1
  f=$(find "$out/prof_cfg$cfg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/cfg${cfg}_kernel_stats.csv"
done
rm -rf "$out"/prof_cfg*/
tail -n 1 "$out"/bench_cfg*.json | cut -c1-300
