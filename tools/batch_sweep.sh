#!/bin/bash
# Runs ON THE GPU BOX: bench.py at several batch sizes (frames per step), headline and fused order.  bash tools/batch_sweep.sh
for r in 1 2 4 6; do
  timeout -k 10 200 python3 bench.py --repeat $r --steps 100 --no-cpu-baseline --no-end-to-end --min-seconds 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('frames/step', d['config']['frames_per_step_per_gpu'], 'fps', d['value'], 'ms/step', d['ms_per_step'], 'fused fps', d['fast_mode']['value'])"
done
