#!/bin/bash
# rocprofv3 kernel trace (no counters) of one bench call with the given bench arguments: tools/gpu_trace_wl.sh <tag> --workload w --batch B ...
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-extras --no-roofline "$@" > $out/log.txt 2>&1
cd $GRAFT_REPO_ROOT
echo "== $tag: $*"
grep "^{" $out/log.txt | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms')"
python3 tools/pmc_summary.py $out | head -24
find $out -name "*.csv" -size +1M -delete
