#!/usr/bin/env python3
"""Device-resident operator chains from two host threads on two contexts of ONE GPU (no allocation / synchronisation between the launches: the kernels of the two streams
really overlap), against the same chain run one context at a time.
    python tools/concurrency_bisect3.py [chain=60]"""
import ctypes as C, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from importlib import import_module
NCH = int(sys.argv[1]) if len(sys.argv) > 1 else 60
pkg = load_package(); L = pkg.lib(); ctxs = [pkg.Context(0), pkg.Context(0)]
rng = np.random.default_rng(5)
def chk(r):
    if r != 0: raise RuntimeError(L.vox_last_error().decode())
SHAPES = [("qkv 1280 -> 3840", 3840, 1280), ("wo 2048 -> 1280", 1280, 2048), ("w1|w3 1280 -> 10240", 10240, 1280), ("w2 5120 -> 1280", 1280, 5120)]
for M in (600, 3000):
    for nm, N, K in SHAPES:
        raw = rng.integers(0, 256, size=N * (K // 32) * 18, dtype=np.uint8).reshape(-1, 18); raw[:, 1] = (raw[:, 1] & 0x03) | 0x28
        Wt = [pkg.Q4Tensor.from_q4_bytes(raw.reshape(-1).copy(), [N, K], c) for c in ctxs]
        xs = [rng.standard_normal((M, K), dtype=np.float32) for _ in range(2)]
        dx = [c.upload(x) for c, x in zip(ctxs, xs)]; dy = [c.alloc(M * N * 4) for c in ctxs]
        def run(k, out):
            for _ in range(NCH): chk(L.vox_q4_linear_forward(ctxs[k].h, Wt[k].h, None, dx[k], 1, M, dy[k], 1))
            ctxs[k].synchronize(); out.append(np.ascontiguousarray(ctxs[k].download(dy[k], (M, N))).view(np.uint32).copy())
        ref = [[], []]
        for k in range(2): run(k, ref[k])
        bad = 0; worst = 0.0
        for attempt in range(4):
            got = [[], []]
            th = [threading.Thread(target=run, args=(k, got[k])) for k in range(2)]
            for x in th: x.start()
            for x in th: x.join()
            for k in range(2):
                if not np.array_equal(ref[k][0], got[k][0]): bad += 1; worst = max(worst, float(np.abs(ref[k][0].view(np.float32) - got[k][0].view(np.float32)).max()))
        print(f"[M {M}, {nm}] chains of {NCH} launches, 4 attempts x 2 contexts: {bad} outputs not bit-identical (largest difference {worst:.3g})", flush=True)
        for k in range(2): ctxs[k].free(dx[k]); ctxs[k].free(dy[k]); Wt[k].close()
