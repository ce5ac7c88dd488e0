#!/usr/bin/env python3
"""Cold (cycling 26 layers, 828 MB > Infinity Cache) vs warm (same layer, resident in the 256 MB Infinity Cache) GEMV launch time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
model = pkg.Q4ModelLoader.from_file(path).load(ctx)
for w, nm in enumerate(["qkv", "wo", "w1w3", "w2", "lm_head"]):
    cold = model.bench_decode_gemv(w, 260 if w != 4 else 40)
    warm = model.bench_decode_gemv(w | 0x100, 260 if w != 4 else 40)
    print(f"{nm:8s} cold {cold[0]:7.2f} us ({cold[1] / cold[0] / 1e3:6.0f} GB/s)   warm {warm[0]:7.2f} us ({warm[1] / warm[0] / 1e3:6.0f} GB/s)", flush=True)
model.close()
