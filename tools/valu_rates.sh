#!/bin/bash
# The VALU issue-rate evidence behind the median mask's roofline (bench.py: VALU_QUARTER_RATE_GINSTR / VALU_FULL_RATE_GINSTR):
# the sustained shader clock under load (clock_probe), the control rows of full-rate instructions (v_add_f32, v_fma_f32,
# v_and_b32 ...) and the min / max / packed classes (valu_rate, cex_rate), and behind the bit-sliced median: what v_bitop3_b32
# needs to reach its rate (issue_rate) and what the CU's vector memory path takes per wave-load (gather_rate) -- ONE run on the GPU box.
# usage: tools/valu_rates.sh   -> gpurun_out/valu_rate.txt (copy to profiles/rNN_valu_rate.txt)
set -u
cd "$GRAFT_REPO_ROOT/tools/microbench"
out="$GRAFT_REPO_ROOT/gpurun_out/valu_rate.txt"
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out"
for t in clock_probe valu_rate cex_rate issue_rate gather_rate scatter_rate; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $t.hip -o /tmp/$t 2>/dev/null || { echo "build of $t failed"; exit 1; }
done
{
  echo "# sustained shader clock under load (s_memtime vs the 100 MHz s_memrealtime)"; /tmp/clock_probe
  echo; echo "# valu_rate: 8 independent chains per lane, 8 waves per SIMD; cycles quoted at 2.4 GHz"; /tmp/valu_rate
  echo; echo "# cex_rate: compare-exchange forms, the packed 16-bit classes, the three-input boolean forms"; /tmp/cex_rate
  echo; echo "# issue_rate: v_bitop3_b32 by waves per SIMD and independent chains per lane"; /tmp/issue_rate
  echo; echo "# gather_rate: whole rows through a buffer resource, 25 per round, by load width and waves per SIMD"; /tmp/gather_rate
  echo; echo "# scatter_rate: a dword wave-load by the number of distinct 128-byte lines its lanes touch (L2-resident table)"; /tmp/scatter_rate
} > "$out" 2>&1
cat "$out"
