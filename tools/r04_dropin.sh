#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in default 12 16 24 32; do
  if [ $t = default ]; then unset REPET_HOST_THREADS; else export REPET_HOST_THREADS=$t; fi
  timeout 300 python3 tools/dropin_probe.py 2>&1 | grep -v amdgpu.ids
done
