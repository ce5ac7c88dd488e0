#!/usr/bin/env python3
"""Finds the synthetic clip (audio seed) on which the PEAKED full-size Q4 model (synth.write_synthetic_gguf(peaked=True), seed 44) produces a transcript whose smallest top-2
logit margin is largest -- the clip of tests/golden/make_fullsize_peaked_golden.py.  Runs the HIP path only (75 ms per clip); the golden itself comes from the CPU oracle.
    gpurun -- python tools/peaked_seed_search.py [n_seeds=120] [seconds=16]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from model_fixtures import cache_dir
pkg = load_package(); S = pkg.synth
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 120; seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
tag = os.environ.get("VOX_SYNTH_PEAKED_N", "24") + "_" + os.environ.get("VOX_SYNTH_PEAKED_GAIN", "5")
path = os.path.join(cache_dir(), "full_q4_peaked_seed44.gguf" if tag == "24_5" else f"full_q4_peaked_seed44_{tag}.gguf")
if not os.path.exists(path):
    S.write_synthetic_gguf(path + ".tmp", S.ModelDims(), seed=44, peaked=True); os.replace(path + ".tmp", path)
ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx); t = pkg.TimeEmbedding(3072).embed(6.0)
melspec = pkg.MelSpectrogram.voxtral(ctx)
best = []
for seed in range(7000, 7000 + n_seeds):
    x = S.synth_audio(seconds, seed=seed)
    mel = np.ascontiguousarray(melspec.compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]
    ids, lg = m.transcribe_streaming(mel, t, return_logits=True)
    srt = np.sort(lg, axis=1); mg = srt[:, -1] - srt[:, -2]; amax = float(np.abs(lg).max())
    best.append((float(mg.min()) / amax, seed, float(mg.min()), amax, len(set(ids.tolist()))))
best.sort(reverse=True)
for r in best[:8]:
    print("rel min margin %.4g  seed %d  min margin %.4g  |logit| max %.2f  distinct ids %d" % r)
print("config %s: median rel min margin over %d seeds: %.4g; median distinct ids %d" % (tag, n_seeds, float(np.median([b[0] for b in best])), int(np.median([b[4] for b in best]))))
var = sorted(best, key=lambda r: (-min(r[4], 12), -r[0]))
for r in var[:4]:
    print("  most varied: rel min margin %.4g  seed %d  min margin %.4g  |logit| max %.2f  distinct ids %d" % r)
m.close(); ctx.close()
