#!/usr/bin/env python3
"""Measurement helper: large-M Q4 GEMM (batched encoder shapes): old 32x128 kernel vs the 64x64-per-wave kernels; also cross-checks results."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
KN = [(1280, 6144), (2048, 1280), (1280, 10240), (5120, 1280)]
Ms = [586, 2344, 9376]
tens = {kn: pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, kn[0] * kn[1], 0.02), [kn[1], kn[0]], ctx) for kn in KN}
ref = {}
for tag, big in [("old", "-1"), ("1x4", "1"), ("auto", "0")]:
    os.environ["VOX_GEMM_BIG"] = big
    for m in Ms:
        row = []
        for k, n in KN:
            x = np.random.default_rng(m + k).standard_normal((m, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
            t = tens[(k, n)]
            for _ in range(2):
                L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
            ctx.synchronize(); t0 = time.perf_counter(); it = 10
            for _ in range(it):
                L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
            ctx.synchronize(); us = (time.perf_counter() - t0) / it * 1e6
            y = ctx.download(dy, (m, n))
            if tag == "old": ref[(m, k, n)] = y; err = 0.0
            else: err = float(np.abs(y - ref[(m, k, n)]).max() / np.abs(ref[(m, k, n)]).max())
            row.append(f"{us:.0f}us/{2 * m * k * n / us / 1e6:.0f}TF/e{err:.0e}")
            ctx.free(dx); ctx.free(dy)
        print(tag, "M=%d" % m, " ".join(row), flush=True)
print("(K,N):", KN)
