#!/bin/bash
# bench lines of cfg 2-5 (stages only), for A/B comparisons inside one gpurun call: tools/quick_bench.sh [label]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 2 3 4 5; do
  timeout 300 python3 bench.py --config $c --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('${1:-} cfg $c', d['ms_per_step'], [(s['name'], s['ms']) for s in d['stages']])"
done
