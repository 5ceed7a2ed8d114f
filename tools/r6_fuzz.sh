#!/bin/bash
# round 6: randomised parity + concurrency soak of the round's last build (profiles/r06_fuzz_parity.txt)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r6
{
timeout 1500 python3 tools/fuzz_parity.py 400 691 2>/dev/null | tail -1
timeout 1200 python3 tools/fuzz_parity.py 250 692 edge 2>/dev/null | tail -1
timeout 600 python3 tools/fuzz_parity.py 60 693 stream 2>/dev/null | tail -1
timeout 600 python3 tools/thread_stress.py 20000 2>/dev/null | tail -1
timeout 600 python3 tools/overlap_stress.py 20000 sim original 2>/dev/null | tail -1
timeout 600 python3 tools/overlap_stress.py 20000 sim extended 2>/dev/null | tail -1
} | tee gpurun_out/r6/fuzz.txt
