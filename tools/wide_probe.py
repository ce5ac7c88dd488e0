#!/usr/bin/env python3
"""Wide decode step probe: a batch of short clips on G forced slot groups, wide step on / off; prints ids agreement and the per-step decode time.
    python tools/wide_probe.py [G=4] [clips=70]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16 * G      # default: every slot holds ONE clip of the same length -> every step runs at full width
pkg = load_package(); ctx = pkg.Context(0)
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
clips = [pkg.synth.synth_audio(secs, seed=100 + i) for i in range(n)]
os.environ["VOX_BATCH_SLOT_GROUPS"] = str(G); os.environ["VOX_BATCH_VERBOSE"] = "1"
res = {}
for tag, env in (("wide", {"VOX_BATCH_NO_WIDE_SPLIT": "1"}), ("split2x2", {}), ("chains", {"VOX_BATCH_NO_WIDE": "1"})):      # one 4-group wide chain / two 2-group wide chains on two streams (the default at four groups) / four 16-row launch chains
    for k, v in env.items():
        os.environ[k] = v
    m.transcribe_batch(clips, t)
    ctx.synchronize(); t0 = time.perf_counter(); outs = m.transcribe_batch(clips, t); ctx.synchronize(); dt = time.perf_counter() - t0
    tm = m.timings(); res[tag] = outs
    print(f"{tag:7s} G={G} {n} clips: {dt * 1e3:.1f} ms, decode {tm['decode_ms']:.1f} ms, replays {tm['graph_replays']}, {tm['decode_ms'] / max(tm['graph_replays'], 1):.3f} ms per step (prefill included)", flush=True)
    for k in env:
        del os.environ[k]
for other in ("split2x2", "chains"):
    same = sum(int(len(a) == len(b) and (a == b).all()) for a, b in zip(res["wide"], res[other]))
    print(f"ids identical, wide vs {other}: {same}/{n}")
m.close(); ctx.close()
