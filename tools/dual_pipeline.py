#!/usr/bin/env python3
"""Measurement helper: P independent transcription pipelines on ONE GPU (each its own context/stream and model handle, driven by
its own host thread) -- batch k+1's encoder overlaps batch k's latency-bound decode.  Usage: dual_pipeline.py <pipelines> <batch> <batches>"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
pkg = load_package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 4
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
ctxs = [pkg.Context(0) for _ in range(P)]
models = [pkg.Q4ModelLoader.from_file(path).load(c) for c in ctxs]
t = pkg.TimeEmbedding(3072).embed(6.0)
clips = [pkg.synth.synth_audio(16.0, seed=1234 + i) for i in range(B)]
ptrs = [[c.upload(x) for x in clips] for c in ctxs]; lens = [x.size for x in clips]
for m, pp in zip(models, ptrs):
    m.transcribe_batch(None, t, device_ptrs=pp, n_samples=lens)          # warm-up (graphs, pools)
ntok = [0] * P
def work(i):
    for _ in range(NB):
        outs = models[i].transcribe_batch(None, t, device_ptrs=ptrs[i], n_samples=lens)
        ntok[i] += sum(len(o) for o in outs)
th = [threading.Thread(target=work, args=(i,)) for i in range(P)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
print(f"pipelines {P} x batch {B} x {NB} batches: {dt * 1e3:.1f} ms, {sum(ntok) / dt:.0f} tok/s aggregate, {dt * 1e3 / (P * NB):.1f} ms per batch", flush=True)
