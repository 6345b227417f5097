#!/bin/bash
# A/B of environment knobs on the GPU box: each argument is one "VAR=val VAR=val" arm, run as a whole bench.
set -u
mkdir -p gpurun_out
for arm in "$@"; do
  echo "== $arm"
  env $arm python tools/bench_exp.py --steps 1 --warmup 1 --batch 32 --no-cpu --breakdown 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown']
print('traj/s %.2f | ms: '%(d['value']) + ' '.join('%s %.1f'%(k,v['ms']) for k,v in b.items()))"
done
