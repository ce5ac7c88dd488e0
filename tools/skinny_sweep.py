#!/usr/bin/env python3
"""Measurement helper: batched-decode (M = 16) skinny kernel over (n-tiles per wave, split-K) at the decoder shapes."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0); L = pkg.lib()
rng = np.random.default_rng(0)
shapes = [("qkv", 3072, 6144), ("wo", 4096, 3072), ("w13", 3072, 18432), ("w2", 9216, 3072), ("lm_head", 3072, 131072)]
tens = {nm: pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx) for nm, k, n in shapes}
m = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for ntw, ks in [(0, 0), (1, 4), (1, 8), (2, 4), (2, 8), (4, 4), (4, 8)]:
    for kk in ("VOX_SKINNY_NTW", "VOX_SKINNY_KS"): os.environ.pop(kk, None)
    if ntw: os.environ["VOX_SKINNY_NTW"] = str(ntw); os.environ["VOX_SKINNY_KS"] = str(ks)
    row = []
    for nm, k, n in shapes:
        x = rng.standard_normal((m, k)).astype(np.float32); dx = ctx.upload(x); dy = ctx.alloc(m * n * 4); t = tens[nm]
        for _ in range(3): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); t0 = time.perf_counter(); it = 50
        for _ in range(it): L.vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1)
        ctx.synchronize(); us = (time.perf_counter() - t0) / it * 1e6
        row.append(f"{nm} {us:.1f}us/{n * k * 18 / 32 / us / 1e6:.2f}TB/s")
        ctx.free(dx); ctx.free(dy)
    print("auto" if not ntw else f"ntw{ntw}ks{ks}", " | ".join(row), flush=True)
