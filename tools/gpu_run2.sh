#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
REPO=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -120 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout 600 python tools/gemv_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemv_sweep.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err
tail -3 gpurun_out/bench2.err; cat gpurun_out/bench2.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof2 -o r01 -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/prof2_bench.json 2> $REPO/gpurun_out/prof2.err
cd $REPO; ls -R gpurun_out/prof2 | head -20
