#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
tail -22 gpurun_out/pytest_gpu.log
timeout 600 python tools/gemv_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemv_sweep.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench3.json 2> gpurun_out/bench3.err
tail -3 gpurun_out/bench3.err; cat gpurun_out/bench3.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof3 -o r01 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof3_bench.json 2> $REPO/gpurun_out/prof3.err
cd $REPO
python - <<'PY'
import sqlite3, glob, re
db = glob.glob('gpurun_out/prof3/*.db')[0]
cur = sqlite3.connect(db).cursor()
with open('gpurun_out/prof3_top_kernels.txt', 'w') as f:
    f.write("name | calls | total_us | avg_us | pct\n")
    for r in cur.execute("select * from top_kernels"):
        f.write(f"{re.sub(r'void |vox::', '', str(r[0]))[:110]} | {r[1]} | {r[2]:.1f} | {r[3]:.3f} | {r[4]:.2f}\n")
print(open('gpurun_out/prof3_top_kernels.txt').read()[:3000])
PY
