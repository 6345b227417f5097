#!/bin/bash
# per-workload A/B of environment knobs: tools/gpu_wl.sh <workload> "VAR=val ..." ...
set -u
w=$1; shift
for arm in "$@"; do
  env $arm python tools/bench_exp.py --workload $w --steps 2 --warmup 1 --no-cpu --breakdown 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown']
print('$w [$arm] traj/s %.2f | ms: '%(d['value']) + ' '.join('%s %.1f'%(k,v['ms']) for k,v in b.items()))"
done
