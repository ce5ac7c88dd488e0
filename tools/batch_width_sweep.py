#!/usr/bin/env python3
"""Decode-step time of vox_transcribe_batch by batch width, launch-based lock-step groups vs the batched decode-layer engine for every 16-row group
(VOX_BATCH_ENGINE_WIDE=1): decides which widths the engine serves.   python tools/batch_width_sweep.py [seconds=16] [widths=16,32,48,64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
widths = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "16,32,48,64").split(",")]
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
clips = [pkg.synth.synth_audio(seconds, seed=4321 + i) for i in range(max(widths))]
for wide in ("", "1"):
    if wide:
        os.environ["VOX_BATCH_ENGINE_WIDE"] = "1"
    else:
        os.environ.pop("VOX_BATCH_ENGINE_WIDE", None)
    for w in widths:
        ptrs = [ctx.upload(c) for c in clips[:w]]; lens = [c.size for c in clips[:w]]
        m.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens)
        n0 = m.set_batch_engine()[1]; ctx.synchronize(); t0 = time.perf_counter()
        outs = m.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=lens)
        ctx.synchronize(); dt = time.perf_counter() - t0; tm = m.timings(); n1 = m.set_batch_engine()[1]
        steps = len(outs[0])
        print(f"width {w:3d}  engine-for-every-width {bool(wide)!s:5}  engine launches {n1 - n0:5d}  decode {tm['decode_ms']:8.2f} ms = {tm['decode_ms'] / steps:6.3f} ms per step "
              f"= {w * steps / tm['decode_ms'] * 1e3:8.0f} tok/s (decode only)   whole batch {dt * 1e3:8.2f} ms", flush=True)
        for p in ptrs:
            ctx.free(p)
