#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r6/tests_full2.log
CFGS="2 3 5" bash tools/ab_lib.sh build_diag/lib_oldstft.so 2>&1 | tee gpurun_out/r6/ab_stft_split_c.log
