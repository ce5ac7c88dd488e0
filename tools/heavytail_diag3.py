#!/usr/bin/env python3
"""Heavy-tail golden, 38-token decoder prefill against the oracle under every K decomposition of the prefill GEMMs: how much of the deviation is summation-order
noise of two f32 computations (the oracle sums sequentially in f32) rather than a property of one kernel."""
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
x = pkg.synth.synth_audio(4.0, seed=4321)
ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx); o = orc.Model(path)
t = pkg.TimeEmbedding(3072).embed(6.0)
xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
ref_audio = o.encode_audio(mel)
dec = m.decoder(); ids = np.array([1] + [32] * 37, dtype=np.int32)
x0 = ref_audio[:38] + o.embed_tokens(ids)
oc = o.cache(64); rh = o.forward_hidden_with_cache(x0, t, oc)
KEYS = ("VOX_NO_SKINNY_MT", "VOX_PREFILL_KERNEL", "VOX_SKINNY_MT2", "VOX_PREFILL_NO_FUSED_FIN", "VOX_PREFILL_NO_SUMK")
outs = {}
for label, env in [("32x128 kernel, no K split", {"VOX_NO_SKINNY_MT": "1"}), ("one-dimensional skinny kernel (4 waves split K)", {"VOX_PREFILL_KERNEL": "1"}),
                   ("2-D default (automatic slices, fused finishing kernels)", {}), ("2-D default, separate finishing kernels", {"VOX_PREFILL_NO_FUSED_FIN": "1"})] + \
                  [(f"2-D, {k} slices forced", {"VOX_SKINNY_MT2": str(k)}) for k in (2, 3, 4, 6, 8, 12)]:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = dec.create_cache_preallocated(64)
    gh = dec.forward_hidden_with_cache(x0[None], t, c)[0]; c.close()
    outs[label] = gh
    print(f"{label:62s}: rel_err vs oracle {rel_err(gh, rh):.3e}   rms rel {np.sqrt(((gh - rh) ** 2).mean()) / np.sqrt((rh ** 2).mean()):.3e}", flush=True)
ks = list(outs)
mean = np.mean([outs[k].astype(np.float64) for k in ks], axis=0)
print(f"oracle vs the mean of the {len(ks)} HIP variants: rel_err {rel_err(rh, mean):.3e};  variants vs their mean: " + ", ".join(f"{rel_err(outs[k], mean):.2e}" for k in ks))
