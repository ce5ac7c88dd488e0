#!/bin/bash
# K-split / tiles-per-wave sweep of the wide step's layer GEMMs (VOX_WIDE_FORCE="N:ntw:kz" overrides the plan of every weight with N rows):  gpurun -- 'bash tools/wide_plan_sweep.sh [mt]'
MT=${1:-4}
run() { echo "== VOX_WIDE_FORCE=$1"; VOX_WIDE_FORCE=$1 timeout 300 python tools/wide_bench.py $MT 52 2>&1 | grep -E "^mt $MT ($2)" | cut -c1-100; }
echo "== default plan"; timeout 300 python tools/wide_bench.py $MT 52 2>&1 | grep -E "^mt $MT (q|wo|w1|w2)" | cut -c1-100
for f in 6144:2:4 6144:2:6 6144:2:12 6144:1:4 6144:1:8; do run $f "q"; done
for f in 18432:2:2 18432:2:4 18432:2:6 18432:1:2 18432:1:3; do run $f "w1"; done
for f in 3072:1:4 3072:1:16 3072:2:8 3072:2:16 3072:1:12 3072:1:24 3072:2:12 3072:2:24 3072:2:18; do run $f "wo|w2"; done
