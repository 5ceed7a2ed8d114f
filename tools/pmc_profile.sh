#!/bin/bash
# Collect rocprofv3 PMC counters for the bench workload, one counter group per pass (gfx950 slot limits:
# 8 SQ, 4 TCC with FETCH_SIZE=3 / WRITE_SIZE=2, see MI355X_MICROARCH.md). Never combined with tracing.
# usage: tools/pmc_profile.sh <outdir> [bench args]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 3 --warmup 1 --series 1 --prewarm-ms 0 --no-cpu-baseline --no-scatter --no-variants "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
  echo "pass $name rc=$?"
}
BENCH_ARGS=("$@")
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE
pass l2 TCC_HIT_sum TCC_MISS_sum
find "$out" -name "*counter_collection.csv" | head
