#!/bin/bash
# linear2 weight-stationary kernel: bit-compare + timing against the tile kernel (tools/lin2_harness.hip); $1 = binary suffix
set -u
mkdir -p gpurun_out
v=${1:-harness}
{
for shape in "245760 512 1536 7680 1" "245760 512 1536 7680 0" "7680 512 1536 7680 1" "30000 512 1536 100 0" "61440 512 1536 7680 1" "1000 512 1536 40 0" \
             "163840 256 1280 160 0" "10240 256 1280 160 0" "368640 256 768 5760 1" "23040 256 768 5760 1" "23000 256 768 5760 0" "51200 128 384 40 0" "333 128 384 40 0"; do
  echo "== $v: $shape"; timeout 120 tools/_exp/lin2_$v $shape 20 | grep -E "lds|mismatch|by feature|BITS|DIFF|round [12]|error|rror|gate"
done
} > gpurun_out/lin2_$v.log 2>&1
cat gpurun_out/lin2_$v.log
