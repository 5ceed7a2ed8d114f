#!/bin/bash
# bench lines (stages only) with another build of the library (REPET_HIP_LIB) and with the shipped one, alternating inside one
# gpurun call:   [CFGS="2 3 4 5"] tools/ab_lib.sh build_diag/lib_variant.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() {
  env REPET_HIP_LIB="$1" timeout 300 python3 bench.py --config $3 --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 cfg $3', d['ms_per_step'], [(s['name'][:10], s['ms']) for s in d['stages']])"
}
for c in ${CFGS:-2}; do for i in 1 2 3; do line "$PWD/$1" other $c; line "$PWD/repet-python_amd/lib/librepet_hip.so" shipped $c; done; done
