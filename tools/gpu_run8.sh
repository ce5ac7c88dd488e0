#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 > gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout 600 python tools/gemm_sweep.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/gemm_sweep.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench8.json 2> gpurun_out/bench8.err
tail -3 gpurun_out/bench8.err; python -c "
import json; d = json.load(open('gpurun_out/bench8.json')); print({k: d[k] for k in ('value','rtf','decode_tok_per_s_ref_def','stage_ms')})"
