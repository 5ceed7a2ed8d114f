#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
timeout 1500 python3 -m pytest tests/test_gpu_stages.py tests/test_gpu_variants.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r6/tests5.log
CFGS="2 3 5 4" bash tools/ab_lib.sh build_diag/lib_oldstft.so 2>&1 | tee gpurun_out/r6/ab_stft_split_b.log
