#!/bin/bash
# A/B of the decode loop forms on the engine path (same box):  gpurun -- 'bash tools/chain_ab.sh'
run() { env "$@" python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch 0 --no-f32 --fleurs-clips 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$*: value', d['value'], 'ms/step', d['ms_per_step'], d['stage_ms'], 'engine us', r['avg_launch_us'])"; }
run VOX_ENGINE_ARGMAX_IN=0 VOX_DECODE_UNROLL=1
run VOX_ENGINE_ARGMAX_IN=0 VOX_DECODE_UNROLL=8
run VOX_ENGINE_ARGMAX_IN=1 VOX_DECODE_UNROLL=1
run VOX_ENGINE_ARGMAX_IN=1 VOX_DECODE_UNROLL=8
run VOX_ENGINE_ARGMAX_IN=1 VOX_DECODE_UNROLL=16
run VOX_ENGINE_ARGMAX_IN=0 VOX_DECODE_UNROLL=1
run VOX_ENGINE_ARGMAX_IN=1 VOX_DECODE_UNROLL=8
