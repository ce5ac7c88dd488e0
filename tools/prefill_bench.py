#!/usr/bin/env python3
"""38-token decoder prefill (gguf/model.rs:908-923) through the C ABI's forward_hidden_with_cache, full-size synthetic model: wall time per call
(includes the 0.5 MB H2D / D2H of the hidden states) for the GEMM kernels selectable with VOX_PREFILL_KERNEL / VOX_NO_SKINNY_MT."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package(); ctx = pkg.Context(0)
model = pkg.Q4ModelLoader.from_file(bench.full_gguf_path(pkg, 42, 0, lambda: None)).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0); dec = model.decoder()
x = (0.5 * np.random.default_rng(0).standard_normal((1, 38, 3072))).astype(np.float32)
ref = None
for label, env in (("32x128 MFMA kernel, 3 passes over the weights (round 1)", {"VOX_NO_SKINNY_MT": "1"}),
                   ("q4_skinny_mt_kernel<3,*> (rows from L2 per wave)", {"VOX_PREFILL_KERNEL": "1"}),
                   ("q4_gemm_kernel<3,NT,tile-ordered B> (48-row tile)", {"VOX_PREFILL_KERNEL": "2"}),
                   ("round-3 form: q4_skinny_mt2_kernel + MFMA flash attention", {"VOX_PREFILL_WIDE": "0", "VOX_ATTN_NO_SMALL": "1"}),
                   ("round 6, wide GEMMs only (MFMA flash attention)", {"VOX_ATTN_NO_SMALL": "1"}),
                   ("round 6, short-sequence attention only (mt2 GEMMs)", {"VOX_PREFILL_WIDE": "0"}),
                   ("default (round 6): q4_wide_kernel planes + short-sequence attention -> XF", {})) + \
                  tuple((f"2-D kernel, {k} slices forced for every operator", {"VOX_SKINNY_MT2": k}) for k in sys.argv[1:]):
    for k in ("VOX_NO_SKINNY_MT", "VOX_PREFILL_KERNEL", "VOX_SKINNY_MT2", "VOX_PREFILL_WIDE", "VOX_ATTN_NO_SMALL"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = dec.create_cache_preallocated(64)
    out = dec.forward_hidden_with_cache(x, t, c); c.reset()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); out = dec.forward_hidden_with_cache(x, t, c); ts.append(time.perf_counter() - t0); c.reset()
    c.close()
    if ref is None:
        ref = out
    print(f"{label:62s}: {np.median(ts) * 1e3:7.3f} ms per 38-token prefill   max|d| vs first variant {np.abs(out - ref).max() / np.abs(ref).max():.2e}", flush=True)
model.close(); ctx.close()
