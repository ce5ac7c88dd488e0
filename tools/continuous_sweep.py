#!/usr/bin/env python3
"""Continuous batching (round 5) against the round-4 lock-step batches on the FLEURS-like corpus (BASELINE configs[4] stand-in): one rank's share of the
647 clips at world 8 (81 clips) and the whole corpus on one GPU, for every slot-group count the planner may pick (VOX_BATCH_SLOT_GROUPS=1..4), the planner's own
choice, and the lock-step form (VOX_BATCH_NO_CONTINUOUS=1 with 64-clip length buckets).  VOX_BATCH_VERBOSE=1 prints the plan per session on stderr.
    python tools/continuous_sweep.py [world=8] [clips=647]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_clips = int(sys.argv[2]) if len(sys.argv) > 2 else 647
pkg = load_package(); ctx = pkg.Context(0); shard = importlib.import_module(pkg.__name__ + ".shard")
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
durs = shard.fleurs_like_durations(n_clips, seed=7)
parts = shard.lpt_partition(durs, world)
clips = {i: pkg.synth.synth_audio(durs[i], seed=9000 + i) for i in range(n_clips)}
os.environ["VOX_BATCH_VERBOSE"] = "1"


def run(idx, bucket):
    ctx.synchronize(); t0 = time.perf_counter(); ntok = 0; st = {"preprocess_ms": 0.0, "encode_ms": 0.0, "decode_ms": 0.0}
    for grp in shard.length_buckets(idx, durs, bucket):
        outs = m.transcribe_batch([clips[i] for i in grp], t); ntok += sum(len(o) for o in outs)
        tm = m.timings()
        for k in st:
            st[k] += tm[k]
    ctx.synchronize(); return time.perf_counter() - t0, ntok, st


def report(tag, idx, bucket):
    run(idx, bucket)                                  # warm-up (pool, graphs' kernels)
    dt, ntok, st = min((run(idx, bucket) for _ in range(2)), key=lambda r: r[0])
    print(f"{tag:58s} {len(idx):4d} clips  {dt * 1e3:8.1f} ms  {ntok / dt:8.0f} tok/s   pre {st['preprocess_ms']:6.1f}  enc {st['encode_ms']:7.1f}  dec {st['decode_ms']:7.1f} ms", flush=True)
    return dt


share = parts[0]
if os.environ.get("VOX_SWEEP_STEP_COST"):      # the cost model's inputs: a rank share on 1..4 forced slot groups, engine forms (<= 2 active groups) on and off
    for eng in (1, 0):
        if not eng: os.environ["VOX_BATCH_CONT_NO_ENGINE"] = "1"
        for G in (1, 2, 3, 4):
            os.environ["VOX_BATCH_SLOT_GROUPS"] = str(G)
            report(f"rank share, {G} slot group(s) forced, engine forms {'on' if eng else 'off'}", share, 4096)
    m.close(); ctx.close(); sys.exit(0)
quick = bool(os.environ.get("VOX_SWEEP_QUICK"))      # only the planner's-choice lines (A/B of a knob, e.g. VOX_BATCH_CHUNK)
t_lock = t_all_lock = float("nan")
if not quick:
    os.environ["VOX_BATCH_NO_CONTINUOUS"] = "1"
    t_lock = report("rank share, lock-step 64-clip buckets (round 4)", share, 64)
    t_all_lock = report("whole corpus, lock-step 64-clip buckets (round 4)", list(range(n_clips)), 64)
    del os.environ["VOX_BATCH_NO_CONTINUOUS"]
    for G in (1, 2, 3, 4):
        os.environ["VOX_BATCH_SLOT_GROUPS"] = str(G)
        report(f"rank share, continuous, {G} slot group(s) forced", share, 4096)
    del os.environ["VOX_BATCH_SLOT_GROUPS"]
t_cont = report("rank share, continuous, planner's choice", share, 4096)
t_all = report("whole corpus, continuous, planner's choice", list(range(n_clips)), 4096)
per = []
for r in range(world):
    run(parts[r], 4096); per.append(min(run(parts[r], 4096)[0] for _ in range(2)))
print(f"per-rank shares (continuous): {[round(v, 3) for v in per]}  -> predicted {world}-GPU scaling {t_all / max(per):.2f} (lock-step: share {t_lock:.3f} s, corpus {t_all_lock:.3f} s -> {t_all_lock / t_lock:.2f})")
m.close(); ctx.close()
