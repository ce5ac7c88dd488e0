#!/bin/bash
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/batch_bench.log
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
model = pkg.Q4ModelLoader.from_file(path).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0)
for B in (1, 16, 32):
    clips = [pkg.synth.synth_audio(16.0, seed=1234 + i) for i in range(B)]
    ptrs = [ctx.upload(c) for c in clips]; lens = [c.size for c in clips]
    model.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens)
    t0 = time.perf_counter(); outs = model.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens); dt = time.perf_counter() - t0
    tm = model.timings(); ntok = sum(len(o) for o in outs)
    print(f"batch {B}: {dt*1e3:.1f} ms total, {ntok/dt:.0f} tok/s e2e, decode {tm['decode_ms']:.1f} ms ({ntok/(tm['decode_ms']/1e3):.0f} tok/s), encode {tm['encode_ms']:.1f} ms, rtf {dt/(16.0*B):.5f}", flush=True)
    for p in ptrs: ctx.free(p)
PY
