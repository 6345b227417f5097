#!/bin/bash
# linear1 launch time vs whole tiles per workgroup (256 workgroups): slope = one 80-step segment, intercept = launch + prologue + tail
set -u
mkdir -p gpurun_out
{
for n in 65536 131072 196608 262144 245760 524288; do
  echo "== $n"; timeout 120 tools/_exp/lin1_harness $n 512 16 2 50 256 0 | grep -E "round 2"
done
} > gpurun_out/lin1_scale.log 2>&1
cat gpurun_out/lin1_scale.log
