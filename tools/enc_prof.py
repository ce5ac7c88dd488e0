#!/usr/bin/env python3
"""Single-clip encoder only (PMC / kernel-trace target): python tools/enc_prof.py [seconds=16] [reps=3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
model = pkg.Q4ModelLoader.from_file(bench.full_gguf_path(pkg, 42, 0, lambda: None)).load(ctx)
SEC = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0; REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = pkg.synth.synth_audio(SEC, seed=1234)
mel = np.ascontiguousarray(pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]
for _ in range(REPS):
    t0 = time.perf_counter(); out = model.encode_audio(mel); print(f"encode {1e3 * (time.perf_counter() - t0):.2f} ms, {out.shape}", flush=True)
