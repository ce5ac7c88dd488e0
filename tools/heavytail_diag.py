#!/usr/bin/env python3
"""Diagnostics for the heavy-tail full-size golden (tests/golden/fullsize_30s_heavytail_oracle.npz), one script, three modes:
  steps    per-step top-logit error of the decode engine and of the per-operator path against the oracle's 30 s golden
  stages   [seconds=4]  stage split: encoder output vs oracle; decoder prefill on the ORACLE's audio embeddings vs oracle (isolates the decoder); lm_head alone
  prefill  the 38-token decoder prefill against the oracle under every K decomposition of the prefill GEMMs: how much of the deviation is summation-order noise
           of two f32 computations (the oracle sums sequentially in f32) rather than a property of one kernel
python tools/heavytail_diag.py <mode> [args]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 128)))
from __graft_entry__ import load_package
from model_fixtures import cache_dir, rel_err
pkg = load_package()
mode = sys.argv[1] if len(sys.argv) > 1 else "steps"
path = os.path.join(cache_dir(), "full_q4_heavytail_seed43.gguf")
if not os.path.exists(path):
    pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=43, heavy_tail=True); os.replace(path + ".tmp", path)
ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
t = pkg.TimeEmbedding(3072).embed(6.0)


def mode_steps():
    g = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_30s_heavytail_oracle.npz"))
    x = pkg.synth.synth_audio(30.0, seed=4321)
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


def _oracle_inputs(secs):
    import oracle_lib as orc
    o = orc.Model(path)
    x = pkg.synth.synth_audio(secs, seed=4321)
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
    return o, mel


def mode_stages():
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    o, mel = _oracle_inputs(secs)
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


def mode_prefill():
    o, mel = _oracle_inputs(4.0)
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


{"steps": mode_steps, "stages": mode_stages, "prefill": mode_prefill}[mode]()
