#!/bin/bash
# Board power and shader clock while the bench loop runs (rocm-smi sampled every 0.5 s inside the timed region), one
# run per environment setting given as an argument ("VAR=value", "-" for the default build).
# usage: tools/power_probe.sh [VAR=value ...]        (on the GPU box)      default: REPET_GRAM_PIPE=1 REPET_GRAM_PIPE=0
set -u
cd "$GRAFT_REPO_ROOT"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -2
[ $# -eq 0 ] && set -- REPET_GRAM_PIPE=1 REPET_GRAM_PIPE=0
for setting in "$@"; do
  echo "== $setting"
  if [ "$setting" = "-" ]; then
    timeout 120 python3 bench.py --steps 8000 --warmup 20 --series 3 --no-cpu-baseline --no-variants --no-scatter > /tmp/bench_probe.json 2>/dev/null &
  else
    env "$setting" timeout 120 python3 bench.py --steps 8000 --warmup 20 --series 3 --no-cpu-baseline --no-variants --no-scatter > /tmp/bench_probe.json 2>/dev/null &
  fi
  pid=$!
  sleep 9
  for i in 1 2 3 4 5 6; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Current Socket Graphics Package Power|sclk clock level" | sed -E 's/GPU\[0\]\s*: //; s/clock level: [0-9]+: //; s/Current Socket Graphics Package Power \(W\):/W/' | tr '\n' ' '
    echo
    sleep 0.5
  done
  wait $pid
  tail -1 /tmp/bench_probe.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], [(s['name'][:5], s['ms']) for s in d['stages']])"
done
