#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_variants.py tests/test_gpu_configs.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6/tests6.log
bash tools/ab_env.sh REPET_PEAK_NORMS 0 1 2>&1 | tee gpurun_out/r6/ab_norms.log
timeout 900 python3 bench.py --steps 200 --series 3 --no-cpu-baseline --no-scatter > gpurun_out/r6/bench_full.json 2> gpurun_out/r6/bench_full.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], json.dumps(d['roofline'])[:1500])
for s in d['stages']:
    if s['name']=='peaks+rank_columns': print(json.dumps(s)[:3000])
PY
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
