#!/bin/bash
# GEMM-variant / epilogue A-B on the GPU box (one process per arm; arms are whole bench runs, so differences
# below ~3 % are noise).  Usage: tools/gpu_sweep.sh "0 1 2 3 4 5"
set -u
mkdir -p gpurun_out
for v in ${1:-0 1 2 3 4 5}; do
  echo "== LSL_GEMM=$v"
  LSL_GEMM=$v python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  LSL_GEMM=$v python tools/bench_exp.py --steps 1 --warmup 1 --batch 32 --no-cpu --breakdown 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown']
print('traj/s %.2f  path TF %.1f  lin1 TF %.1f | ms: '%(d['value'], d['roofline']['whole_path_tflops'], d['roofline']['achieved']) + ' '.join('%s %.1f'%(k,v['ms']) for k,v in b.items()))"
done
