#!/usr/bin/env python3
"""Which phase of the path is not reproducible when TWO contexts run on one GPU at the same time?  Each task runs R times on each of two (context, model replica) pairs,
first one pair after the other (reference), then both at once in two host threads; the outputs must be bit-identical.
    python tools/concurrency_bisect.py [R=6]"""
import importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
ctxs = [ctx, pkg.Context(0)]; models = [m, m.replicate(ctxs[1])]
mel = pkg.MelSpectrogram.voxtral(ctx)
clips = [pkg.synth.synth_audio(6.0 + 1.5 * (i % 11), seed=500 + i) for i in range(96)]
mels = [np.ascontiguousarray(mel.compute_log(pkg.pad_audio(pkg.peak_normalize(c) if hasattr(pkg, "peak_normalize") else c)).T)[None] for c in clips[:4]]

def flat(o):
    if isinstance(o, (list, tuple)): return np.concatenate([np.asarray(x).ravel().view(np.uint32) if np.asarray(x).dtype == np.float32 else np.asarray(x).ravel().astype(np.uint32) for x in o] + [np.array([len(o)], np.uint32)])
    a = np.asarray(o); return a.ravel().view(np.uint32) if a.dtype == np.float32 else a.ravel().astype(np.uint32)

TASKS = [
    ("encode_audio, one 12 s clip's mel", lambda mm, k, r: mm.encode_audio(mels[(k + r) % 4])),
    ("transcribe_audio (single-stream engine)", lambda mm, k, r: mm.transcribe_audio(clips[(3 * k + r) % 96], t)),
    ("transcribe_batch, 16 clips (lock-step, batched engine)", lambda mm, k, r: mm.transcribe_batch(clips[16 * k:16 * k + 16], t)),
    ("transcribe_batch, 48 clips (continuous)", lambda mm, k, r: mm.transcribe_batch(clips[48 * k:48 * k + 48], t)),
]
sel = os.environ.get("BISECT_TASKS")
for ti, (name, fn) in enumerate(TASKS):
    if sel and str(ti) not in sel.split(","): continue
    def run(k, out):
        for r in range(R): out.append(flat(fn(models[k], k, r)))
        ctxs[k].synchronize()
    for k in range(2): run(k, [])      # warm-up
    ref = [[], []]
    for k in range(2): run(k, ref[k])
    bad_total = 0
    for attempt in range(2):
        got = [[], []]
        th = [threading.Thread(target=run, args=(k, got[k])) for k in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        bad = sum(int(a.shape != b.shape or not np.array_equal(a, b)) for k in range(2) for a, b in zip(ref[k], got[k]))
        bad_total += bad
    print(f"[{name}] {2 * R} calls x 2 attempts: {bad_total} not bit-identical to the one-at-a-time run", flush=True)
models[1].close(); ctxs[1].close(); m.close(); ctx.close()
