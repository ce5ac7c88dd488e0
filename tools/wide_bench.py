#!/usr/bin/env python3
"""Per-operator timing of the wide decode step (vox_bench_wide): GEMM launch, finishing launch, both, and the same operator as `mt` 16-row skinny launches.
    python tools/wide_bench.py [mt=4] [iters=104]        (VOX_WIDE_FORCE="N:ntw:kz" tries another plan for one weight shape)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
mts = [int(sys.argv[1])] if len(sys.argv) > 1 else [4, 3, 2]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 104
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx)
lib = pkg.lib()
names = ["q|k|v", "wo", "w1|w3", "w2", "lm_head"]
for mt in mts:
    tot = [0.0, 0.0]
    for which in range(5):
        out = (C.c_double * 4)()
        rc = lib.vox_bench_wide(m.h, which, mt, iters if which < 4 else max(iters // 4, 8), out)
        if rc != 0:
            print(names[which], "error", lib.vox_last_error()); continue
        print(f"mt {mt} {names[which]:8s} gemm {out[0]:7.2f} us  finish {out[1]:6.2f} us  both {out[2]:7.2f} us   {mt} x skinny {out[3]:7.2f} us", flush=True)
        if which < 4:
            tot[0] += out[2]; tot[1] += out[3]
    print(f"mt {mt} layer GEMMs: wide {tot[0]:.1f} us, skinny x{mt} serial {tot[1]:.1f} us")
m.close(); ctx.close()
