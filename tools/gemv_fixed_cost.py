#!/usr/bin/env python3
"""Measurement helper (run under rocprofv3 --kernel-trace --stats): Q4 GEMV durations for shrinking N at K = 3072 -> the fixed in-kernel
cost (activation staging + first weight round trip + reduction) once streaming time vanishes."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0); k = 3072
for n in (64, 512, 3072, 6144, 18432):
    t = pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx)
    x = rng.standard_normal((1, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(n * 4)
    for _ in range(50): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, 1, C.c_void_p(dy), 1)
    ctx.synchronize(); print("N", n, flush=True)
