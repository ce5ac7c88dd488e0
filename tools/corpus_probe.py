#!/usr/bin/env python3
"""Stage breakdown of the FLEURS-like corpus call (bench.py `fleurs_like`): one vox_transcribe_batch over 647 clips with VOX_BATCH_VERBOSE=1.
    python tools/corpus_probe.py [n_clips=647] [reps=2] [sessions=1]      (sessions > 1: vox_model_set_sessions)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 647
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sessions = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pkg = load_package(); ctx = pkg.Context(0)
shard = importlib.import_module(pkg.__name__ + ".shard")
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
durs = shard.fleurs_like_durations(n, seed=7)
clips = [pkg.synth.synth_audio(durs[i], seed=9000 + i) for i in range(n)]
os.environ["VOX_BATCH_VERBOSE"] = "1"
if sessions > 1:
    m.set_sessions(sessions)
for r in range(reps + 1):
    ctx.synchronize(); t0 = time.perf_counter(); outs = m.transcribe_batch(clips, t); ctx.synchronize(); dt = time.perf_counter() - t0
    tm = m.timings(); ids = sum(len(o) for o in outs)
    print(f"rep {r}: {dt:.3f} s, {ids} ids, {ids / dt:.0f} tok/s; stage ms: pre {tm['preprocess_ms']:.0f} enc {tm['encode_ms']:.0f} dec {tm['decode_ms']:.0f}, replays {tm['graph_replays']}", flush=True)
m.close(); ctx.close()
