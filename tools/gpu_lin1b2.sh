#!/bin/bash
# K <= 256 linear1 variants (LIN1_B2): bit-compare + timing on the 256- and 128-wide shapes; "$1..." = binary suffixes
for v in "$@"; do
for shape in "163840 256 16 4" "368640 256 16 2" "23040 256 16 2" "10240 256 16 4" "15104 128 4 2" "245760 512 16 2" "163840 256 8 4" "30000 256 8 2" "777 256 16 2" "100000 128 4 4" "2560 128 4 2"; do
  echo "== $v: $shape"; LIN1_WPT=1 timeout 120 tools/_exp/lin1_$v $shape 30 256 0 | grep -E "BITS|DIFF|round 2|unsupported"
  echo "== $v: $shape (no wpt, grid 256, pos mode 1)"; timeout 120 tools/_exp/lin1_$v $shape 10 256 1 | grep -E "BITS|DIFF|unsupported"
done
done
