#!/usr/bin/env python3
"""Heavy-tail golden, stage split: encoder output vs oracle; decoder prefill on the ORACLE's audio embeddings vs oracle (isolates the decoder)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 128)))
import oracle_lib as orc
from __graft_entry__ import load_package
from model_fixtures import cache_dir, rel_err
pkg = load_package()
path = os.path.join(cache_dir(), "full_q4_heavytail_seed43.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=43, heavy_tail=True); os.replace(path + ".tmp", path)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
x = pkg.synth.synth_audio(secs, seed=4321)
ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx); o = orc.Model(path)
t = pkg.TimeEmbedding(3072).embed(6.0)
xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
t0 = time.time(); ref_audio = o.encode_audio(mel); print(f"oracle encoder {time.time()-t0:.1f}s, S={ref_audio.shape[0]}", flush=True)
out_audio = m.encode_audio(mel[None])[0]
print(f"encoder output: rel_err (max|d|/max|ref|) {rel_err(out_audio, ref_audio):.3e}; max|ref| {np.abs(ref_audio).max():.3f}; rms rel {np.sqrt(((out_audio-ref_audio)**2).mean())/np.sqrt((ref_audio**2).mean()):.3e}")
dec = m.decoder(); ids = np.array([1] + [32] * 37, dtype=np.int32)
x0 = ref_audio[:38] + o.embed_tokens(ids)
oc = o.cache(64); c = dec.create_cache_preallocated(64)
rh = o.forward_hidden_with_cache(x0, t, oc); gh = dec.forward_hidden_with_cache(x0[None], t, c)[0]
print(f"decoder prefill hidden on the oracle's embeddings: rel_err {rel_err(gh, rh):.3e}; max|ref| {np.abs(rh).max():.1f}; rows rms rel {np.sqrt(((gh-rh)**2).mean())/np.sqrt((rh**2).mean()):.3e}")
rl = o.lm_head(rh[-1:]); gl = dec.lm_head(gh[None, -1:])[0]
print(f"logits: rel_err {rel_err(gl, rl):.3e}; max|logit| {np.abs(rl).max():.1f}")
gl2 = dec.lm_head(rh[None, -1:])[0]
print(f"lm_head alone (oracle hidden in): rel_err {rel_err(gl2, rl):.3e}")
# decoder on HIP's own embeddings
x1 = out_audio[:38] + o.embed_tokens(ids); c2 = dec.create_cache_preallocated(64)
gh2 = dec.forward_hidden_with_cache(x1[None], t, c2)[0]
print(f"decoder prefill hidden on HIP's embeddings vs oracle: rel_err {rel_err(gh2, rh):.3e}")
