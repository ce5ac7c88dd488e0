#!/usr/bin/env python3
"""Diagnostic for the heavy-tail golden: per-step top-logit error of the decode engine and of the per-operator path against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from model_fixtures import cache_dir
pkg = load_package()
g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_30s_heavytail_oracle.npz"))
path = os.path.join(cache_dir(), "full_q4_heavytail_seed43.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=43, heavy_tail=True); os.replace(path + ".tmp", path)
x = pkg.synth.synth_audio(30.0, seed=4321)
ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0)
mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
res = {}
for engine in (True, False):
    m.set_decode_engine(engine)
    ids, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)
    e = np.abs(lg.max(axis=1) - top1)
    agree = ids == rids; stop = len(ids) if agree.all() else int(np.argmin(agree))
    print(f"engine={engine}: ids agree {stop}/{len(ids)}; top-logit err: step0 {e[0]:.3e} step1 {e[1]:.3e} step2 {e[2]:.3e} max {e.max():.3e} at step {int(e.argmax())}; rel to |logit|max {e.max()/amax:.2e}; "
          f"per-step relative to own top: max {float((e/np.abs(top1)).max()):.2e}")
    print("   err every 20 steps:", " ".join(f"{v:.2e}" for v in e[::20]))
    res[engine] = lg
d = np.abs(res[True] - res[False])
print(f"engine vs per-operator: max |dlogit| {d.max():.3e}; per step max: first {d[0].max():.2e} {d[1].max():.2e} {d[2].max():.2e} last {d[-1].max():.2e}")
print("logits_step0 vs oracle (first 4096):", float(np.abs(res[False][0, :4096] - g["logits_step0"]).max()))
