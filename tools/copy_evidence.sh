#!/bin/bash
# (build container) copy what tools/evidence_run.sh <tag> left under gpurun_out/round/ into profiles/<tag>_*
tag=${1:?usage: tools/copy_evidence.sh <round tag>}
R=gpurun_out/round
cp $R/bench_cfg2.json profiles/${tag}_bench_sim_cfg2.json
for n in 3 4 5; do cp $R/bench_cfg$n.json profiles/${tag}_bench_cfg$n.json; cp $R/cfg${n}_kernel_stats.csv profiles/${tag}_cfg${n}_kernel_stats.csv; done
cp $R/cfg2_kernel_stats.csv profiles/${tag}_sim_kernel_stats.csv
cp $R/pmc_summary.txt profiles/${tag}_pmc_summary.txt
cp $R/pmc_traffic.json profiles/${tag}_pmc_traffic.json
cp $R/valu_rate.txt profiles/${tag}_valu_rate.txt
cp $R/stream_latency.json profiles/${tag}_stream_latency.json
cp $R/peak_gram_spans.txt profiles/${tag}_peak_gram_spans.txt
cp $R/bitslice_select.txt profiles/${tag}_bitslice_select.txt
