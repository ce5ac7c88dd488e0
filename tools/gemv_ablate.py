#!/usr/bin/env python3
"""Times the five decode-step GEMVs (26 layers cycled, HBM-cold) with the library VOX_LIB points at -- run once per measurement build
(voxtral-mini-realtime-rs_amd/build.py abl_*) to see what each part of q4_gemv_kernel costs.  Prints one line."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
model = pkg.Q4ModelLoader.from_file(path).load(ctx)
out = []
for rep in range(2):
    out = []
    for which, nm in enumerate(["qkv", "wo", "w1w3", "w2", "lm_head"]):
        us, nbytes, kn = model.bench_decode_gemv(which, 260 if which != 4 else 40)
        out.append(f"{nm} {us:6.2f}")
tag = os.path.basename(os.environ.get("VOX_LIB", "product")).replace("libvoxtral_hip_", "").replace(".so", "")
print(f"{tag:14s} " + "  ".join(out) + "   us per launch", flush=True)
model.close()
