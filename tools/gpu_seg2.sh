#!/bin/bash
set -u
mkdir -p gpurun_out
{
for b in lin1_harness lin1_dplain lin1_aplain; do
for shape in "245760 512 16 2" "163840 256 16 4"; do
  echo "== $b $shape"; timeout 120 tools/_exp/$b $shape 50 256 0 | grep -E "BITS|DIFF|round [12]"
done; done
} > gpurun_out/seg2.log 2>&1
cat gpurun_out/seg2.log
