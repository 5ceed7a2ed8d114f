#!/bin/bash
# Everything the round's profiles/ files come from, in one call on the GPU box (rocprofv3 kernel statistics and bench lines
# of cfg 2-5, the PMC passes and their summaries, the stream latency bench, the VALU issue rates, the spans of the peak and
# Gram kernels). A step that fails stops the script: a missing file must not look like evidence.
# usage: tools/evidence_run.sh <round tag, e.g. r04>      -> gpurun_out/round/, gpurun_out/pmc_<tag>/
set -euo pipefail
tag=${1:?usage: tools/evidence_run.sh <round tag>}
root="$GRAFT_REPO_ROOT"
mkdir -p "$root/gpurun_out/round" "$root/gpurun_out/pmc_$tag"
cd /tmp && export TMPDIR=/tmp && cd "$root"
step() { echo "== $*"; "$@"; }
# the diagnostic build travels with the snapshot (make -C repet-python_amd/csrc stamps, in the build container): an older one
# than the library lacks its newest entry points, and the step that loads it would fail after everything else has run
[ build_diag/lib_stamps.so -nt repet-python_amd/lib/librepet_hip.so ] || { echo "build_diag/lib_stamps.so is older than the library: run 'make -C repet-python_amd/csrc stamps' first"; exit 1; }
# the PMC passes FIRST: the bench lines written afterwards carry roofline.traffic from THIS call's counters (the file they read,
# profiles/<tag>_pmc_traffic.json, is written here on the box; it records the sha256 of the library it was measured on, and so
# does the bench line)
step timeout 1200 bash tools/pmc_profile.sh "gpurun_out/pmc_$tag" > "gpurun_out/pmc_$tag/pmc.log" 2>&1
step timeout 120 python3 tools/pmc_traffic.py "gpurun_out/pmc_$tag" gpurun_out/round/pmc_traffic.json > gpurun_out/round/pmc_traffic.log 2>&1
cp gpurun_out/round/pmc_traffic.json "profiles/${tag}_pmc_traffic.json"
step timeout 120 python3 tools/pmc_summary.py "gpurun_out/pmc_$tag" > gpurun_out/round/pmc_summary.txt 2>&1
find "gpurun_out/pmc_$tag" -name "*.csv" -size +2M -delete
step timeout 1800 bash tools/round_profile.sh > gpurun_out/round/round.log 2>&1
step timeout 300 python3 tools/stream_bench.py > gpurun_out/round/stream_latency.json 2> gpurun_out/round/stream.err
step timeout 300 bash tools/valu_rates.sh > gpurun_out/round/valu_rate.log 2>&1
cp gpurun_out/valu_rate.txt gpurun_out/round/valu_rate.txt
step timeout 300 python3 tools/peak_stamps.py > gpurun_out/round/peak_gram_spans.txt 2>&1
# the bit-sliced selection on its own, on the lists of the bench clip: time, and cycles per phase of a wave
step timeout 300 python3 tools/dump_sim_lists.py /tmp/sim_lists.bin > gpurun_out/round/bitslice_select.txt 2>&1
for flags in "" "-DSTAMPS"; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $flags -mllvm -pragma-unroll-threshold=131072 -Irepet-python_amd/csrc tools/microbench/bitslice_select.hip -o /tmp/bitslice_select 2>/dev/null
  { echo "# bitslice_select $flags: real lists, lists inside +-150 frames, uniform lists"; /tmp/bitslice_select 0 /tmp/sim_lists.bin; /tmp/bitslice_select 150; /tmp/bitslice_select 0; } >> gpurun_out/round/bitslice_select.txt 2>&1
done
tail -n 3 gpurun_out/round/round.log | cut -c1-400
tail -3 gpurun_out/round/pmc_traffic.log
ls -la gpurun_out/round
