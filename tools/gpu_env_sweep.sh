#!/bin/bash
# one bench workload under several environment settings (tools build).  Usage: tools/gpu_env_sweep.sh "<bench args>" "ENV=.." ...
args=$1; shift
for cfg in "$@"; do
  env $cfg python tools/bench_exp.py $args --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-34s' % '$cfg', ' ms/call %.3f  traj/s %.2f  linear1 %.1f us' % (d['ms_per_step'], d['value'], 1000*d['roofline']['avg_launch_ms']))"
done
