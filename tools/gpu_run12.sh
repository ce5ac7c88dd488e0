#!/bin/bash
# full GPU suite + kernel-stats profile of the bench (encoder attention on MFMA)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
true
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_bench -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 0 > $REPO/gpurun_out/bench_prof.log 2>&1
cd $REPO; tail -2 gpurun_out/bench_prof.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_bench/bench_kernel_stats.csv')))
for r in rows[:24]:
    print(r['Name'][:95].replace('void vox::',''), r['Calls'], f"{float(r['TotalDurationNs'])/1e6:.1f}ms", f"{float(r['AverageNs'])/1e3:.1f}us", r['Percentage'])
PY
