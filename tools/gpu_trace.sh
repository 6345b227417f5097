#!/bin/bash
# rocprofv3 kernel trace of one bench step with the given environment arms: tools/gpu_trace.sh "VAR=val ..." ["VAR=val ..."]
set -u
cd /tmp && export TMPDIR=/tmp
for arm in "$@"; do
  tag=$(echo "$arm" | tr -c 'A-Za-z0-9' '_')
  out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
  rm -rf $out; mkdir -p $out
  export $arm
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-roofline > $out/log.txt 2>&1
  unset ${arm%%=*}
  echo "== $arm"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out | head -12
  find $out -name "*.csv" -size +1M -delete
done
