#!/usr/bin/env python3
"""Measurement helper: decode-step GEMV launch durations for several rows-per-wave / grid settings (HIP events)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
names = ["qkv", "wo", "w1w3", "w2", "lm_head"]
KEYS = ("VOX_GEMV_R", "VOX_GEMV_R_PAIR", "VOX_GEMV_R_ARGMAX", "VOX_GEMV_WGS")
for tag, env in [("default", {}), ("R1", {"VOX_GEMV_R": "1", "VOX_GEMV_R_ARGMAX": "1"}), ("R2", {"VOX_GEMV_R": "2"}),
                 ("R4", {"VOX_GEMV_R": "4", "VOX_GEMV_R_PAIR": "4", "VOX_GEMV_R_ARGMAX": "4"}),
                 ("wgs512", {"VOX_GEMV_WGS": "512"}), ("wgs1024", {"VOX_GEMV_WGS": "1024"}), ("wgs1536", {"VOX_GEMV_WGS": "1536"})]:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    model = pkg.Q4ModelLoader.from_file(path).load(ctx)
    row = {}
    for w, nm in enumerate(names):
        us, nbytes, kn = model.bench_decode_gemv(w, 260 if w != 4 else 40)
        row[nm] = (round(us, 2), round(nbytes / us / 1e3), kn.split("<")[1][:-1])
    model.close()
    print(tag, json.dumps(row), flush=True)
