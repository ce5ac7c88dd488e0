#!/bin/bash
# first GPU session: parity tests (all, not -x), smoke, short bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python - <<'PY' 2>&1 | tee gpurun_out/probe.log
import sys; sys.path.insert(0,'.')
from __graft_entry__ import load_package
pkg = load_package(); import ctypes as C
n = C.c_int32(); pkg.lib().vox_device_count(C.byref(n)); print("devices", n.value)
PY
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -6 | tee -a gpurun_out/probe.log
nproc | tee -a gpurun_out/probe.log; free -g | head -2 | tee -a gpurun_out/probe.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -20 | tee gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench1.json 2> gpurun_out/bench1.err
tail -5 gpurun_out/bench1.err; cat gpurun_out/bench1.json
