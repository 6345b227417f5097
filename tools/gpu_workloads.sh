#!/bin/bash
# Throughput of every BASELINE.json configuration family with the product library (one bench run each, no CPU leg):
#   tools/gpu_workloads.sh > gpurun_out/workloads.txt
set -u
run() {
  python bench.py --no-cpu --no-extras "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
print('%-18s B=%-5d updates=%-4d  %10.2f traj/s  %9.3f ms/call  whole-path %6.1f TFLOP/s (%.1f %% of bf16 peak)' % (c['workload'], c['batch_per_gpu'], c['state_updates'], d['value'], d['ms_per_step'], d['roofline']['whole_path_tflops'], 100*d['roofline']['whole_path_frac']))"
}
run --workload md17_bench --steps 3 --warmup 1
run --workload md17_bench --batch 1 --steps 20 --warmup 5
run --workload md17_bench --batch 8 --steps 3 --warmup 1
run --workload md17_ref --steps 60 --warmup 20
run --workload md17_ref --batch 64 --steps 5 --warmup 2
run --workload pedestrian_scene --steps 50 --warmup 5
run --workload pedestrian --batch 160 --steps 20 --warmup 5
run --workload pedestrian --steps 10 --warmup 3
run --workload nba --steps 3 --warmup 1
run --workload nba --batch 64 --steps 5 --warmup 2
run --workload peptide --steps 1 --warmup 1
