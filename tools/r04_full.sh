#!/bin/bash
# the whole GPU suite + stamps of the peak kernel + a bench line
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/r04e"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$out/gpu_tests.log" 2>&1; echo "gpu tests rc $?"
tail -4 "$out/gpu_tests.log"
PEAK_PHASES=1 timeout 600 python3 tools/peak_stamps.py > "$out/stamps.txt" 2>&1; echo "stamps rc $?"
grep -v "^  t = " "$out/stamps.txt" | grep -A5 "fastest half"
timeout 600 python3 bench.py --no-scatter > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc $?"
tail -1 "$out/bench.json" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], [(s['name'], s['ms']) for s in d['stages']], d.get('array_in_array_out'), d.get('array_in_array_out_pcm16'), d.get('fp32_gemm_variant'))"
