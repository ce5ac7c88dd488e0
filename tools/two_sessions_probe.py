#!/usr/bin/env python3
"""Overlap inside a rank's share: the FLEURS-like corpus as S concurrent sessions on ONE GPU -- S contexts (own stream, workspaces, graphs), S model
replicas (vox_model_replicate), S host threads, the corpus LPT-split S ways -- against one session over the whole corpus.  Prints wall time, tok/s and whether the ids agree.
    python tools/two_sessions_probe.py [n_clips=647] [S=2] [stagger_ms=0]"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 647
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
stagger = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
pkg = load_package(); ctx = pkg.Context(0)
shard = importlib.import_module(pkg.__name__ + ".shard")
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
durs = shard.fleurs_like_durations(n, seed=7)
clips = [pkg.synth.synth_audio(durs[i], seed=9000 + i) for i in range(n)]
ctxs = [ctx] + [pkg.Context(0) for _ in range(S - 1)]
models = [m] + [m.replicate(c) for c in ctxs[1:]]
if os.environ.get("PROBE_SHARED"):      # vox_ctx_set_shared: no batched engines (a second session makes their bounded waits expire: a strike = a wasted re-run), planner on the scaled table
    for c in ctxs: c.set_shared(True)
parts = shard.lpt_partition(durs, S)

def one():
    ctx.synchronize(); t0 = time.perf_counter(); outs = m.transcribe_batch(clips, t); ctx.synchronize(); return time.perf_counter() - t0, outs

def many(concurrent=True):
    res = [None] * S; errs = []
    def work(k):
        try:
            if stagger > 0 and k > 0:
                time.sleep(stagger * k / 1e3)
            res[k] = models[k].transcribe_batch([clips[i] for i in parts[k]], t); ctxs[k].synchronize()
        except Exception as e:
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
    t0 = time.perf_counter()
    if concurrent:
        for x in th: x.start()
        for x in th: x.join()
    else:      # the same S sessions one after the other: the reference for "concurrency changes nothing" (same inputs, same plans)
        for x in th: x.start(); x.join()
    dt = time.perf_counter() - t0
    if errs: raise errs[0]
    outs = [None] * n
    for k in range(S):
        for i, o in zip(parts[k], res[k]): outs[i] = o
    return dt, outs

if os.environ.get("PROBE_C_SESSIONS"):      # the library's own sessions (vox_model_set_sessions: hidden contexts + replicas + library threads) instead of this script's threads
    def many(concurrent=True):
        if not concurrent:
            m.set_sessions(1); ctx.set_shared(True)
            try:
                res = [m.transcribe_batch([clips[i] for i in parts[k]], t) for k in range(S)]
            finally:
                ctx.set_shared(False)
            outs = [None] * n
            for k in range(S):
                for i, o in zip(parts[k], res[k]): outs[i] = o
            return 0.0, outs
        m.set_sessions(S)
        ctx.synchronize(); t0 = time.perf_counter(); outs = m.transcribe_batch(clips, t); ctx.synchronize(); dt = time.perf_counter() - t0
        m.set_sessions(1)
        return dt, outs
one(); many()      # warm-up: workspaces, graphs' kernels, planner calibration
for rep in range(2):
    d1, o1 = one(); dS, oS = many()
    ids = sum(len(o) for o in o1)
    same = sum(int(len(a) == len(b) and (a == b).all()) for a, b in zip(o1, oS))
    dQ, oQ = many(False); race = sum(int(len(a) == len(b) and (a == b).all()) for a, b in zip(oQ, oS))
    print(f"rep {rep}: the {S} sessions in sequence {dQ:.3f} s; ids identical, concurrent vs in sequence: {race}/{n}")
    print(f"rep {rep}: one session {d1:.3f} s ({ids / d1:.0f} tok/s); {S} concurrent sessions {dS:.3f} s ({ids / dS:.0f} tok/s), x{d1 / dS:.3f}; ids identical {same}/{n}; engine state {[mm.set_batch_engine() for mm in models]}", flush=True)
for mm in models[1:]: mm.close()
for c in ctxs[1:]: c.close()
m.close(); ctx.close()
