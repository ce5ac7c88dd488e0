#!/usr/bin/env python3
"""PMC target: the large-M Q4 GEMM at two batched-encoder shapes (M = 16 clips x 586 frames)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
for m, k, n in [(9376, 1280, 10240), (9376, 5120, 1280)]:
    t = pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx)
    x = rng.standard_normal((m, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
    for _ in range(2): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
    ctx.synchronize(); t0 = time.perf_counter()
    for _ in range(5): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
    ctx.synchronize(); us = (time.perf_counter() - t0) / 5 * 1e6
    print(m, k, n, f"{us:.1f} us {2 * m * k * n / us / 1e6:.0f} TF/s", flush=True)
