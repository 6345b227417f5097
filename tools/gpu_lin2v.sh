#!/bin/bash
# linear2 weight-stationary kernel: timing of probe / variant builds (tools/_exp/lin2_*), full cfg-2 launch
set -u
mkdir -p gpurun_out
{
for v in "$@"; do
  echo "== $v"; timeout 120 tools/_exp/lin2_$v 245760 512 1536 7680 1 20 | grep -E "BITS|DIFF|round [12]|rror|wave"
done
} > gpurun_out/lin2v.log 2>&1
cat gpurun_out/lin2v.log
