#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6/tests_full.log
for nt in 0 1 0 1 0 1; do
REPET_HOST_NT=$nt timeout 600 python3 bench.py --steps 50 --series 1 --no-variants --no-cpu-baseline 2>gpurun_out/r6/nt_$nt.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('scatter_gather') or {}; a=d.get('array_in_array_out') or {}; p=d.get('array_in_array_out_pcm16') or {}; print('REPET_HOST_NT=$nt', 'scatter ms', s.get('ms'), s.get('verified'), '| float64', a.get('ms_min'), a.get('ms_median'), '| pcm16', p.get('ms_min'), p.get('ms_median'))"
done 2>&1 | tee gpurun_out/r6/nt_ab.log
