#!/bin/bash
# linear1 token-stationary kernel A/B: tools/_exp/lin1_<variant> for each argument (bit-compare against the tile kernel + timing)
set -u
mkdir -p gpurun_out
{
for v in "$@"; do
for shape in "245760 512 16 2" "163840 256 16 4" "15104 128 4 2" "61440 384 16 4" "30720 512 16 2" "7680 512 16 2"; do
  echo "== $v: $shape"; LIN1_WPT=1 timeout 120 tools/_exp/lin1_$v $shape 30 256 0 | grep -E "BITS|DIFF|round [12]"
done
done
} > gpurun_out/lin1ab.log 2>&1
cat gpurun_out/lin1ab.log
