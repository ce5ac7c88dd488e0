#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_batch -o b16 -- env VOX_BATCH_NO_GRAPH=1 python $REPO/tools/batch_prof.py 16 > $REPO/gpurun_out/prof_batch.log 2>&1
cd $REPO; grep batch gpurun_out/prof_batch.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_batch/b16_kernel_stats.csv')))
for r in rows[:22]:
    print(r['Name'][:95].replace('void vox::',''), r['Calls'], f"{float(r['TotalDurationNs'])/1e6:.1f}ms", f"{float(r['AverageNs'])/1e3:.1f}us", r['Percentage'])
PY
