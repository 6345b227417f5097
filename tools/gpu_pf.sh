#!/bin/bash
set -u
mkdir -p gpurun_out
{
for pf in 0 1; do
  for shape in "245760 512 16 2" "163840 256 16 4"; do
    echo "== pf $pf $shape"; timeout 120 tools/_exp/lin1_pf$pf $shape 50 256 0 | grep -E "BITS|DIFF|round [12]"
  done
done
timeout 120 tools/_exp/lin1_stamp 245760 512 16 2 50 256 0 2>&1 | grep -E "wg   0|round 2"
} > gpurun_out/pf.log 2>&1
cat gpurun_out/pf.log
