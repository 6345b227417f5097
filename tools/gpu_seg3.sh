#!/bin/bash
set -u
mkdir -p gpurun_out
{
for b in lin1_harness lin1_x2; do
for shape in "245760 512 16 2" "61440 384 16 4" "30720 512 16 2" "7680 512 16 2"; do
  echo "== $b $shape"; LIN1_WPT=1 timeout 120 tools/_exp/$b $shape 50 256 0 | grep -E "BITS|DIFF|round [12]"
done; done
timeout 120 tools/_exp/lin1_stamp_x2 245760 512 16 2 50 256 0 2>&1 | grep -E "segments|wg   0 wave [04]:|round 2"
} > gpurun_out/seg3.log 2>&1
cat gpurun_out/seg3.log
