#!/bin/bash
# bench lines (stages only) of several configs under two settings of one environment switch, alternating inside one gpurun call:
#   CFGS="2 3 5" tools/ab_env_all.sh REPET_FFT_STEREO scalar pairs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() {
  env "$1=$2" timeout 300 python3 bench.py --config $3 --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1=$2 cfg $3', d['ms_per_step'], [(s['name'][:10], s['ms']) for s in d['stages']])"
}
for c in ${CFGS:-2 3 5}; do for i in 1 2; do line "$1" "$2" $c; line "$1" "$3" $c; done; done
