#!/usr/bin/env python3
"""Soak test of the persistent engines' hand-off protocols (flags 16 bytes apart, partial planes that carry their own validity, tagged granules): the same batch of 16 clips
and the same single clip transcribed REPS times; every repetition must reproduce the first one's ids bit for bit and no hand-off timeout may be reported.  Then the
two-group launches (round 5): a ragged 48-clip batch on 32 slots -- two groups per launch while both are active, one group per launch behind that, slots refilled,
positions up to ~430 -- REPS times against the ids of the launch chains (the interleaved groups time-share LDS regions behind a publish-count guard: a race would show here).
python tools/soak.py [reps=30]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from __graft_entry__ import load_package
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pkg = load_package(); ctx = pkg.Context(0)
m = pkg.Q4ModelLoader.from_file(bench.full_gguf_path(pkg, 42, 0, lambda: None)).load(ctx)
t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
clips = [pkg.synth.synth_audio(6.0 + 1.5 * i, seed=700 + i) for i in range(16)]      # ragged: 6 .. 28.5 s (positions up to ~360: several attention rounds)
ref = m.transcribe_batch(clips, t); one = m.transcribe_audio(clips[-1], t)
eng0 = m.set_batch_engine()[1]; t0 = time.time(); bad = 0
for r in range(reps):
    out = m.transcribe_batch(clips, t)
    bad += sum(not np.array_equal(a, b) for a, b in zip(out, ref))
    bad += not np.array_equal(m.transcribe_audio(clips[-1], t), one)
n_eng = m.set_batch_engine()[1] - eng0
print(f"soak: {reps} x (16 ragged clips through the batched engine + the longest clip single-stream): {n_eng} batched-engine launches, {sum(len(o) for o in ref)} ids per batch, "
      f"mismatching results {bad}, engine still active {m.set_batch_engine()[0]} / {m.set_decode_engine(True)}, {time.time() - t0:.1f} s")
wide = [pkg.synth.synth_audio(3.0 + 0.56 * ((7 * i) % 48), seed=900 + i) for i in range(48)]      # 3 .. 29.3 s
os.environ["VOX_BATCH_SLOT_GROUPS"] = "2"; os.environ["VOX_BATCH_CONT_NO_ENGINE"] = "1"
ref2 = m.transcribe_batch(wide, t)                                   # the forked launch chains
del os.environ["VOX_BATCH_CONT_NO_ENGINE"]
eng1 = m.set_batch_engine()[1]; t1 = time.time(); bad2 = 0
for r in range(reps):
    out = m.transcribe_batch(wide, t)
    bad2 += sum(not np.array_equal(a, b) for a, b in zip(out, ref2))
n_eng2 = m.set_batch_engine()[1] - eng1
del os.environ["VOX_BATCH_SLOT_GROUPS"]
print(f"soak: {reps} x (48 ragged clips, 32 slots, continuous batch on the engine forms) against the launch chains' ids: {n_eng2} engine launches (two groups per launch while both are active), "
      f"{sum(len(o) for o in ref2)} ids per batch, mismatching results {bad2}, engine still active {m.set_batch_engine()[0]}, {time.time() - t1:.1f} s")
sys.exit(1 if bad or bad2 or not m.set_batch_engine()[0] else 0)
