#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 900 python3 -m pytest tests -x -q -m gpu -k "nyquist or clips_in_flight or scatter_separate or bench_config5_and_config2 or parallel_over" 2>&1 | tail -8 | tee gpurun_out/r6/tests2.log
bash tools/ab_env.sh REPET_GRAM_STAGGER_US 0 8 2>&1 | tee gpurun_out/r6/ab_stagger8.log
REPET_GRAM_STAGGER_GROUPS=4 bash tools/ab_env.sh REPET_GRAM_STAGGER_US 0 5 2>&1 | tee gpurun_out/r6/ab_stagger5x4.log
bash tools/ab_env.sh REPET_NYQUIST lane lists 2>&1 | tee gpurun_out/r6/ab_nyq.log
for d in 1 2 3 4; do
REPET_BENCH_DEPTH=$d timeout 600 python3 bench.py --steps 50 --series 1 --no-variants --no-cpu-baseline 2>gpurun_out/r6/scatter_$d.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('depth $d', json.dumps(d.get('scatter_gather'))[:600]); print({k: d.get(k) for k in ('ms_per_step',)}, json.dumps(d.get('array_in_array_out'))[:400])"
done 2>&1 | tee gpurun_out/r6/scatter.log
