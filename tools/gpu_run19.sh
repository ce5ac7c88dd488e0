#!/bin/bash
# per-shape sweep of the skinny kernel in XF mode (batch-16 decode): VOX_SKINNY_FORCE=N:ntw:ks overrides one weight shape
echo -n "auto: "; timeout 300 python tools/batch_prof.py 16 2>&1 | grep batch | tail -1
for f in 18432:4:8 18432:2:4 18432:2:8 6144:1:8 6144:2:4 6144:4:4 6144:4:8 131072:4:8 131072:2:4 3072:1:4; do
  echo -n "$f: "; VOX_SKINNY_FORCE=$f timeout 300 python tools/batch_prof.py 16 2>&1 | grep batch | tail -1
done
