#!/bin/bash
# kernel trace of one bench step for each library variant tools/_exp/lib_<name>.so (copied over the in-tree library on the GPU box only)
set -u
cd /tmp && export TMPDIR=/tmp
cp $GRAFT_REPO_ROOT/lam_slide_amd/liblamslide_hip.so /tmp/lib_orig.so
for n in "$@"; do
  cp $GRAFT_REPO_ROOT/tools/_exp/lib_$n.so $GRAFT_REPO_ROOT/lam_slide_amd/liblamslide_hip.so
  out=$GRAFT_REPO_ROOT/gpurun_out/trace_lib_$n
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras --no-roofline > $out/log.txt 2>&1
  echo "== $n"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out | head -8
  grep "^{" $out/log.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   traj/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],1))"
  find $out -name "*.csv" -size +1M -delete
done
cp /tmp/lib_orig.so $GRAFT_REPO_ROOT/lam_slide_amd/liblamslide_hip.so
