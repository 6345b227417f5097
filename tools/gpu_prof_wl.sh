#!/bin/bash
# rocprofv3 kernel-trace stats of one bench workload: tools/gpu_prof_wl.sh <tag> <bench args...>
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras "$@" > $out/run.log 2>&1
tail -1 $out/run.log | cut -c1-400
f=$(find $out -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -d, -f1-8
find $out -name "*.csv" -size +2M -delete
