#!/bin/bash
# the same bench line (cfg 2, stages only) from an older tree's build and from this one, alternating inside one gpurun call:
#   tools/ab_rounds.sh build_diag/r03tree      (git archive <commit> | tar -x -C build_diag/r03tree; make there)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
line() {
  (cd "$1" && timeout 300 python3 bench.py --config 2 --steps 20 --warmup 3 --series 3 --no-cpu-baseline --no-scatter --no-variants 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], [(s['name'][:10], s['ms']) for s in d['stages']])")
}
for i in 1 2 3; do line "$1" old; line . new; done
