#!/bin/bash
# pass-size sweep at the headline shape: trajectories per pass (LSL_CHUNK_TRAJ) -> ms per step
for c in 2 4 8 16 32; do
  LSL_CHUNK_TRAJ=$c python bench.py --no-cpu --no-extras --no-roofline --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunk $c: %.2f traj/s  %.1f ms/step' % (d['value'], d['ms_per_step']))"
done
