#!/usr/bin/env python3
"""Measurement helper: q4_matmul (MFMA GEMM path) at the encoder / prefill / batch shapes for several tile settings."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
shapes = [(586, 1280, 6144), (586, 2048, 1280), (586, 1280, 10240), (586, 5120, 1280), (146, 5120, 3072), (38, 3072, 6144),
          (38, 3072, 18432), (38, 9216, 3072), (16, 3072, 18432), (16, 9216, 3072)]
tens = {}
for m, k, n in shapes:
    if (k, n) not in tens:
        tens[(k, n)] = pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx)
for tag, env in [("auto", {}), ("mt4nt2", {"VOX_GEMM_MT": "4", "VOX_GEMM_NT": "2"}), ("mt4nt1", {"VOX_GEMM_MT": "4", "VOX_GEMM_NT": "1"}),
                 ("mt2nt2", {"VOX_GEMM_MT": "2", "VOX_GEMM_NT": "2"}), ("mt2nt1", {"VOX_GEMM_MT": "2", "VOX_GEMM_NT": "1"}),
                 ("mt1nt2", {"VOX_GEMM_MT": "1", "VOX_GEMM_NT": "2"}), ("mt1nt1", {"VOX_GEMM_MT": "1", "VOX_GEMM_NT": "1"})]:
    for kk in ("VOX_GEMM_MT", "VOX_GEMM_NT"):
        os.environ.pop(kk, None)
    os.environ.update(env)
    row = []
    for m, k, n in shapes:
        if tag.startswith("mt1") and m > 64: row.append("-"); continue
        x = rng.standard_normal((m, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
        t = tens[(k, n)]
        for _ in range(3):
            L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); t0 = time.perf_counter(); it = 30
        for _ in range(it):
            L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); us = (time.perf_counter() - t0) / it * 1e6
        row.append(f"{us:.0f}us/{2 * m * k * n / us / 1e6:.0f}TF")
        ctx.free(dx); ctx.free(dy)
    print(tag, " ".join(row), flush=True)
print("shapes (M,K,N):", shapes)
