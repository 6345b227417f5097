#!/bin/bash
# linear1 work split for small launches (harness: bits + time), then peptide / nba B=64 / md17 B=1 end to end
set -u
mkdir -p gpurun_out
{
for shape in "16000 384 16 4" "7680 512 16 2" "10240 256 16 4" "30720 512 16 2"; do
  echo "== $shape linear split"; timeout 120 tools/_exp/lin1_harness $shape 200 256 0 | grep -E "grid|BITS|DIFF|round 2"
  echo "== $shape aligned split"; LIN1_WPT=1 timeout 120 tools/_exp/lin1_harness $shape 200 256 0 | grep -E "grid|BITS|DIFF|round 2"
done
bash tools/gpu_wl.sh peptide "LSL_LIN1_ALIGN=0" "LSL_LIN1_ALIGN=1" "LSL_GEMM2=16" "LSL_GEMM2=17" "LSL_GEMM2=18"
} > gpurun_out/small_split.log 2>&1
tail -30 gpurun_out/small_split.log
