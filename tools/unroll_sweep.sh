#!/bin/bash
# decode graph unroll sweep on the engine path:  gpurun -- 'bash tools/unroll_sweep.sh 1 4 8 16 32 1'
for u in "$@"; do
  VOX_DECODE_UNROLL=$u python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch 0 --no-f32 --fleurs-clips 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('unroll $u: value', d['value'], 'ms/step', d['ms_per_step'], d['stage_ms'], 'engine us', r['avg_launch_us'])"
done
