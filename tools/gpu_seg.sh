#!/bin/bash
set -u
mkdir -p gpurun_out
{
for shape in "245760 512 16 2" "163840 256 16 4" "15104 128 4 2" "61440 384 16 4" "30720 512 16 2"; do
  echo "== $shape"; LIN1_WPT=1 timeout 120 tools/_exp/lin1_harness $shape 50 256 0 | grep -E "BITS|DIFF|round [12]"
done
timeout 120 tools/_exp/lin1_stamp 245760 512 16 2 50 256 0 2>&1 | grep -E "segments|wg   0 wave [04]:|round 2"
} > gpurun_out/seg.log 2>&1
cat gpurun_out/seg.log
