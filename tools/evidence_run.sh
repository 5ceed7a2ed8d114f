mkdir -p gpurun_out/round gpurun_out/pmc_r03
timeout 1500 bash tools/round_profile.sh > gpurun_out/round/round.log 2>&1
timeout 900 bash tools/pmc_profile.sh gpurun_out/pmc_r03 > gpurun_out/pmc_r03/pmc.log 2>&1
timeout 120 python3 tools/pmc_traffic.py gpurun_out/pmc_r03 gpurun_out/round/pmc_traffic.json > gpurun_out/round/pmc_traffic.log 2>&1
timeout 120 python3 tools/pmc_summary.py gpurun_out/pmc_r03 > gpurun_out/round/pmc_summary.txt 2>&1
find gpurun_out/pmc_r03 -name "*.csv" -size +2M -delete
timeout 300 python3 tools/stream_bench.py > gpurun_out/round/stream_latency.json 2> gpurun_out/round/stream.err
timeout 120 tools/microbench/cex_rate > gpurun_out/round/cex_rate.txt 2>&1
timeout 200 python3 tools/peak_stamps.py > gpurun_out/round/peak_gram_spans.txt 2>&1
tail -n 3 gpurun_out/round/round.log | cut -c1-400; tail -3 gpurun_out/round/pmc_traffic.log; ls -la gpurun_out/round
