#!/bin/bash
# tests + single-clip bench + gemm sweep + batch throughput (no profiler)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/bench_quick.json
python - <<'PY'
import json; d = json.load(open('gpurun_out/bench_quick.json'))
print('bench', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['stage_ms'], 'ref-def', d['decode_tok_per_s_ref_def'], 'rtf', d['rtf'])
print({k: v['avg_us'] for k, v in d['roofline']['all_decode_gemvs'].items()})
PY
timeout 600 python tools/gemm_sweep.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/gemm_sweep.txt
timeout 600 python tools/batch_prof.py 16 2>&1 | grep batch | tee gpurun_out/batch16.txt
