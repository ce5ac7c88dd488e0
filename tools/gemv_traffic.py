#!/usr/bin/env python3
"""PMC target: the decode-engine launch (the product's decode step) plus the w1|w3 GEMV and lm_head of the per-operator path, so FETCH_SIZE / WRITE_SIZE per launch can be read off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
model = pkg.Q4ModelLoader.from_file(path).load(ctx)
for which, iters in ((2, 104), (4, 16), (5, 24)):      # 5 = the whole decode step as one decode-engine launch
    us, nbytes, kn = model.bench_decode_gemv(which, iters)
    print(which, kn, round(us, 2), "us", int(nbytes), "B", flush=True)
model.close()
