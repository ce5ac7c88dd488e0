#!/usr/bin/env python3
"""A/B of the large-M Q4 GEMM (q4_gemm_big_kernel) across library builds: run once per VOX_LIB.  Encoder shapes, M = one clip (800) and a 16-clip batch (12800)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
KN = [(1280, 3840), (1280, 10240), (5120, 1280), (1280, 1280)]
Ms = [800, 12800]
if os.environ.get("VOX_AB_SHAPES") == "prefill":      # the batched decoder prefill: 16 x 38 rows
    KN = [(3072, 6144), (4096, 3072), (3072, 18432), (9216, 3072)]; Ms = [608, 1216, 2432]
tens = {kn: pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, kn[0] * kn[1], 0.02), [kn[1], kn[0]], ctx) for kn in KN}
for m in Ms:
    row = []
    for k, n in KN:
        x = np.random.default_rng(m + k).standard_normal((m, k)).astype(np.float32)
        if os.environ.get('VOX_AB_ZERO'): x[:] = 0.0      # (DVFS probe: same instruction stream, no toggling operands)
        dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
        t = tens[(k, n)]
        for _ in range(2):
            L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); t0 = time.perf_counter(); it = 10
        for _ in range(it):
            L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); us = (time.perf_counter() - t0) / it * 1e6
        y = ctx.download(dy, (min(m, 64), n))
        row.append(f"K{k} N{n}: {us:7.0f} us {2 * m * k * n / us / 1e6:5.0f} TF/s (sum {float(np.abs(y).sum()):.6e})")
        ctx.free(dx); ctx.free(dy)
    print(os.path.basename(os.environ.get("VOX_LIB", "product")), "M=%d" % m, " | ".join(row), flush=True)
