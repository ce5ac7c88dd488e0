#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 600 python tools/gemv_sweep.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/gemv_sweep.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench6.json 2> gpurun_out/bench6.err
tail -3 gpurun_out/bench6.err; cat gpurun_out/bench6.json
