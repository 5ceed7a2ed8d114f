#!/bin/bash
# the cfg-2 bench line (stages only) with another build of the library (REPET_HIP_LIB) and with the shipped one, alternating inside
# one gpurun call:   tools/ab_lib.sh build_diag/lib_variant.so
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() {
  env REPET_HIP_LIB="$1" timeout 300 python3 bench.py --config 2 --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], [(s['name'][:10], s['ms']) for s in d['stages']])"
}
for i in 1 2 3; do line "$PWD/$1" other; line "$PWD/repet-python_amd/lib/librepet_hip.so" shipped; done
