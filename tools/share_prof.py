#!/usr/bin/env python3
"""rocprofv3 target: ONE rank's share of the FLEURS-like corpus at world 8 (81 clips) through vox_transcribe_batch (continuous batching), after a warm-up call.
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/share_prof.py [world=8] [rank=0]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8; rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pkg = load_package(); ctx = pkg.Context(0); shard = importlib.import_module(pkg.__name__ + ".shard")
m = pkg.Q4ModelLoader.from_file(bench.full_gguf_path(pkg, 42, 0, lambda: None)).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
durs = shard.fleurs_like_durations(647, seed=7); idx = shard.lpt_partition(durs, world)[rank]
clips = [pkg.synth.synth_audio(durs[i], seed=9000 + i) for i in idx]
os.environ["VOX_BATCH_VERBOSE"] = "1"
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter(); outs = m.transcribe_batch(clips, t); ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"share of rank {rank}/{world}: {len(clips)} clips, {sum(len(o) for o in outs)} ids, {dt * 1e3:.1f} ms, stage {m.timings()}", flush=True)
m.close(); ctx.close()
