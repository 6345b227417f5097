#!/bin/bash
# epilogue decomposition of linear1 (timing probes; results are wrong on purpose).  Usage: tools/gpu_probe.sh "0 3 11 19 27"
set -u
mkdir -p gpurun_out
for v in ${1:-0 3 11 19 27}; do
  echo "== LSL_PROBE=$v"
  LSL_PROBE=$v python tools/bench_exp.py --steps 1 --warmup 1 --batch 32 --no-cpu --breakdown 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown']
print('traj/s %.2f | ms: '%(d['value']) + ' '.join('%s %.1f'%(k,v['ms']) for k,v in b.items()))"
done
