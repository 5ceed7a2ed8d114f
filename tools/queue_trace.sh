#!/bin/bash
# Which hardware queue does every kernel of a step go to? rocprofv3 --kernel-trace of three bench steps, once in a plain
# process and once in one that has initialised an RCCL communicator (the 1-rank check of the distributed path); prints
# (kernel, Queue_Id) counts. The two streams of a context must sit on different queues for their kernels to overlap.
# usage: tools/queue_trace.sh [extra bench args]
set -u
out="$GRAFT_REPO_ROOT/gpurun_out/q"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in dist plain; do
  if [ $m = dist ]; then export REPET_BENCH_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0; else unset REPET_BENCH_DIST; fi
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$out/$m" -- python3 bench.py --gpus 1 --steps 3 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-scatter "$@" > "$out/$m.log" 2>&1
  f=$(find "$out/$m" -name "*kernel_trace.csv" | head -1)
  python3 - "$m" "$f" <<'PY'
import csv, collections, sys
m, f = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
print(m, "queue / stream columns:", [k for k in rows[0].keys() if "ueue" in k or "tream" in k])
c = collections.Counter((r["Kernel_Name"].split("(")[0][-44:], r.get("Queue_Id"), r.get("Stream_Id")) for r in rows)
for k, v in sorted(c.items()):
    print("   %-46s queue %-3s stream %-3s x%d" % (k[0], k[1], k[2], v))
PY
done
find "$out" -name "*.csv" -delete
