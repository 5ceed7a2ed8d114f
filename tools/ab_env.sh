#!/bin/bash
# the cfg-2 bench line (stages only) under two settings of one environment switch, alternating inside one gpurun call:
#   tools/ab_env.sh REPET_MEDIAN rank bits
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() {
  env "$1=$2" timeout 300 python3 bench.py --config ${CFG:-2} --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$2', d['ms_per_step'], [(s['name'][:10], s['ms']) for s in d['stages']])"
}
for i in 1 2 3; do line "$1" "$2"; line "$1" "$3"; done
