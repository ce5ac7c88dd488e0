#!/usr/bin/env python3
"""Is one of the heavy kernels sensitive to a LIGHT kernel of another stream sharing its SIMDs?  Victim thread: one device-resident operator per launch (Q4 GEMMs of the encoder's
shapes, the encoder's attention), output compared with a quiet run; noise thread: an endless chain of element-wise adds (18-VGPR waves that fit next to anything) on another context.
    python tools/concurrency_bisect5.py [R=40]"""
import ctypes as C, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pkg = load_package(); L = pkg.lib(); ctxs = [pkg.Context(0), pkg.Context(0)]
rng = np.random.default_rng(5)
def chk(r):
    if r != 0: raise RuntimeError(L.vox_last_error().decode())
stop = threading.Event()
NA = 484 * 6144
na = ctxs[1].upload(rng.standard_normal(NA, dtype=np.float32)); nb = ctxs[1].upload(rng.standard_normal(NA, dtype=np.float32)); no = ctxs[1].alloc(NA * 4)
def noise():
    while not stop.is_set():
        for _ in range(50): chk(L.vox_tensor_add(ctxs[1].h, na, nb, NA, no, 1))
        ctxs[1].synchronize()
def victim_case(name, launch, out_ptr, shape):
    def once():
        launch(); ctxs[0].synchronize(); return np.ascontiguousarray(ctxs[0].download(out_ptr, shape)).view(np.uint32).copy()
    ref = once(); assert np.array_equal(ref, once())
    stop.clear(); th = threading.Thread(target=noise); th.start()
    bad = 0; worst = 0.0; nel = 0
    for r in range(R):
        g = once()
        if not np.array_equal(ref, g): bad += 1; worst = max(worst, float(np.abs(ref.view(np.float32) - g.view(np.float32)).max())); nel = max(nel, int((ref != g).sum()))
    stop.set(); th.join()
    print(f"[{name}] {R} launches next to the noise stream: {bad} not bit-identical (largest difference {worst:.3g}, most differing elements {nel} of {ref.size})", flush=True)
SHAPES = [("qkv 1280 -> 6144", 6144, 1280), ("wo 2048 -> 1280", 1280, 2048), ("w1|w3 1280 -> 10240", 10240, 1280), ("w2 5120 -> 1280", 1280, 5120)]
for M in (484, 3000):
    for nm, N, K in SHAPES:
        raw = rng.integers(0, 256, size=N * (K // 32) * 18, dtype=np.uint8).reshape(-1, 18); raw[:, 1] = (raw[:, 1] & 0x03) | 0x28
        Wt = pkg.Q4Tensor.from_q4_bytes(raw.reshape(-1).copy(), [N, K], ctxs[0])
        dx = ctxs[0].upload(rng.standard_normal((M, K), dtype=np.float32)); dy = ctxs[0].alloc(M * N * 4)
        victim_case(f"Q4 GEMM M {M}, {nm}", lambda: chk(L.vox_q4_linear_forward(ctxs[0].h, Wt.h, None, dx, 1, M, dy, 1)), dy, (M, N))
        ctxs[0].free(dx); ctxs[0].free(dy); Wt.close()
for M in (484, 1500):
    q, k, v = (ctxs[0].upload(rng.standard_normal((M, 2048), dtype=np.float32)) for _ in range(3)); o = ctxs[0].alloc(M * 2048 * 4)
    victim_case(f"attention {M} rows, 32 heads of 64, window 750", lambda: chk(L.vox_attention(ctxs[0].h, q, k, v, M, M, 32, 32, 64, 0, 750, o, 1)), o, (M, 2048))
