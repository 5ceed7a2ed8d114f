#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
bash tools/ab_lib.sh build_diag/lib_peakw3.so 2>&1 | tee gpurun_out/r6/ab_peakw3.log
bash tools/ab_lib.sh build_diag/lib_peakw2.so 2>&1 | tee gpurun_out/r6/ab_peakw2.log
REPET_HIP_LIB=$PWD/build_diag/lib_peakw3.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/prof -- python3 bench.py --config 2 --steps 20 --warmup 5 --series 1 --no-cpu-baseline --no-scatter --no-variants > gpurun_out/r6/prof.log 2>&1
f=$(find gpurun_out/r6/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r6/cfg2_kernel_stats_w3.csv
rm -rf gpurun_out/r6/prof
python3 - gpurun_out/r6/cfg2_kernel_stats_w3.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>5s}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]:>6s} %')
PY
