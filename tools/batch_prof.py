#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
model = pkg.Q4ModelLoader.from_file(path).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0      # clip length (a short clip keeps rocprofv3 --pmc targets short: long ones time out / crash)
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
clips = [pkg.synth.synth_audio(SEC, seed=1234 + i) for i in range(B)]
ptrs = [ctx.upload(c) for c in clips]; lens = [c.size for c in clips]
for _ in range(REPS):
    t0 = time.perf_counter(); outs = model.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens); dt = time.perf_counter() - t0
    tm = model.timings(); print(f"batch {B}: {dt*1e3:.1f} ms, decode {tm['decode_ms']:.1f}, encode {tm['encode_ms']:.1f}", flush=True)
