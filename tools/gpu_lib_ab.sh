#!/bin/bash
# A/B of whole-library builds on the GPU box: tools/gpu_lib_ab.sh <workload args...> -- lib1.so lib2.so ...   (paths relative to the repo root)
# Each arm copies its library over lam_slide_amd/liblamslide_hip.so IN THE BOX'S SCRATCH COPY and runs bench.py --breakdown.
set -u
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
cp lam_slide_amd/liblamslide_hip.so /tmp/_product.so
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/_product.so lam_slide_amd/liblamslide_hip.so; else cp "$lib" lam_slide_amd/liblamslide_hip.so; fi
  python bench.py --no-cpu --no-extras --breakdown "${args[@]}" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown']
print('$lib: traj/s %.2f ms/step %.1f | ms: '%(d['value'], d['ms_per_step']) + ' '.join('%s %.1f'%(k,v['ms']) for k,v in b.items()))"
done
done
cp /tmp/_product.so lam_slide_amd/liblamslide_hip.so
