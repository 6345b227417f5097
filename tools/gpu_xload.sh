#!/bin/bash
# linear1 activation load: whole cache lines + LDS transposition (lin1_harness) vs fragment-shaped loads (lin1_x0)
set -u
mkdir -p gpurun_out
{
for shape in "245760 512 16 2" "16000 384 16 4" "7680 512 16 2" "10240 256 16 4" "163840 256 16 4" "15104 128 4 2"; do
  for bin in lin1_x0 lin1_harness; do
    echo "== $shape $bin"; LIN1_WPT=1 timeout 120 tools/_exp/$bin $shape 100 256 0 | grep -E "BITS|DIFF|round [12]"
  done
done
} > gpurun_out/xload.log 2>&1
cat gpurun_out/xload.log
