#!/bin/bash
REPO=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/prof_fixed -o fx -- python $REPO/tools/gemv_fixed_cost.py > $REPO/gpurun_out/prof_fixed.log 2>&1
cd $REPO
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/prof_fixed/fx_kernel_trace.csv')))
agg = collections.defaultdict(list)
for r in rows:
    if 'gemv' in r['Kernel_Name']: agg[(r['Kernel_Name'][:40], r['Grid_Size'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in agg.items():
    v = sorted(v); print(k, 'n', len(v), 'median us', v[len(v)//2] / 1e3, 'min', v[0] / 1e3)
PY
