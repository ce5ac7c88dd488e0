#!/usr/bin/env python3
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
for m, k, n in [(16, 3072, 18432), (16, 9216, 3072), (16, 3072, 131072)]:
    t = pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx)
    x = rng.standard_normal((m, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
    for _ in range(3): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(20): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
    ctx.synchronize(); print(m, k, n, f"{(time.perf_counter() - t0) / 20 * 1e6:.1f} us", flush=True)
