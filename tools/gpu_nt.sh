#!/bin/bash
# does a streaming-store epilogue keep the operand tiles in L2?  FETCH_SIZE of linear1 with LSL_PROBE=32 (nt stores) vs 0
cd /tmp && export TMPDIR=/tmp
for p in 0 32; do
  out=$GRAFT_REPO_ROOT/gpurun_out/nt_$p
  LSL_PROBE=$p rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/bench_exp.py --steps 1 --warmup 0 --no-cpu > $out.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out | grep -A2 "EpiLinear1" | head -4
done
cd $GRAFT_REPO_ROOT
tools/gpu_ab.sh "LSL_PROBE=0" "LSL_PROBE=32" "LSL_PROBE=0" "LSL_PROBE=32"
