#!/bin/bash
# One gpurun call of round 6's inner loop: a test selection, the quick bench lines, an A/B over an environment switch, and
# the kernel statistics of cfg 2.  usage: tools/r6_call.sh "<pytest -k expression or empty>" [ENVVAR a b]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
if [ -n "${1:-}" ]; then
  timeout 900 python3 -m pytest tests -x -q -m gpu -k "$1" 2>&1 | tail -15 | tee gpurun_out/r6/tests.log
fi
bash tools/quick_bench.sh now 2>&1 | tee gpurun_out/r6/quick.log
if [ -n "${2:-}" ]; then bash tools/ab_env.sh "$2" "$3" "$4" 2>&1 | tee gpurun_out/r6/ab.log; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/prof -- python3 bench.py --config 2 --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > gpurun_out/r6/prof.log 2>&1
f=$(find gpurun_out/r6/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r6/cfg2_kernel_stats.csv
rm -rf gpurun_out/r6/prof
python3 - gpurun_out/r6/cfg2_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
