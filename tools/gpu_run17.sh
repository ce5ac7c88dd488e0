#!/bin/bash
# batched-decode sweep in XF mode: n-tiles per wave / split-K of the skinny kernels (env knobs apply to every decode GEMM)
for cfg in "0 0" "1 4" "1 8" "2 4" "2 8" "4 4" "4 8"; do
  set -- $cfg
  if [ "$1" = "0" ]; then echo -n "auto: "; timeout 300 python tools/batch_prof.py 16 2>&1 | grep batch | tail -1
  else echo -n "ntw$1 ks$2: "; VOX_SKINNY_NTW=$1 VOX_SKINNY_KS=$2 timeout 300 python tools/batch_prof.py 16 2>&1 | grep batch | tail -1; fi
done
