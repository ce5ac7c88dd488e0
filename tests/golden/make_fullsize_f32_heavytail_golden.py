#!/usr/bin/env python3
"""Generates tests/golden/fullsize_30s_f32_heavytail_oracle.npz: the CPU oracle's greedy ids and per-step top-2 logits of the f32 SafeTensors path at FULL
size for STRESS statistics -- the synthetic dense model of synth.write_fast_dense_checkpoint(heavy_tail=True) (seed 8: power-of-two Student-t(4) block scales,
six outlier channels x 64 in the decoder's residual stream, final norm centred on 5; every value exact in bf16 and in the oracle's f16 copy) on the 30 s clip of the
Q4 stress golden (seed 4321: 234 decoder positions).  Same recipe as make_fullsize_f32_golden.py: run once on the GPU box's host CPU
(`gpurun -- python tests/golden/make_fullsize_f32_heavytail_golden.py`), the small result is committed and replayed by
tests/test_gpu_fullsize.py::test_full_30s_f32_heavytail_vs_oracle_golden.  Nothing here touches the GPU or /root/reference."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 128)))
import oracle_lib as orc
from __graft_entry__ import load_package
from model_fixtures import cache_dir, dense_head_sha
pkg = load_package(); S = pkg.synth
SEED = 8; SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
gg = os.path.join(cache_dir(), f"full_dense_heavytail_seed{SEED}.gguf"); st = os.path.join(cache_dir(), f"full_dense_heavytail_seed{SEED}.safetensors")
t0 = time.time()
if not (os.path.exists(gg) and os.path.exists(st)):
    S.write_fast_dense_checkpoint(st + ".tmp", gg + ".tmp", S.ModelDims(), seed=SEED, heavy_tail=True); os.replace(st + ".tmp", st); os.replace(gg + ".tmp", gg)
print(f"checkpoints ready in {time.time() - t0:.1f} s ({os.path.getsize(st) / 1e9:.2f} GB safetensors, {os.path.getsize(gg) / 1e9:.2f} GB gguf)", flush=True)
x = S.synth_audio(SECONDS, seed=4321)
xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
t = pkg.TimeEmbedding(3072).embed(6.0)
t0 = time.time(); o = orc.Model(gg); print(f"oracle load {time.time() - t0:.1f} s", flush=True)
t0 = time.time(); ids, lg = o.transcribe_streaming(mel, t, want_logits=True); dt = time.time() - t0
srt = np.sort(lg, axis=1)
out = os.path.join(ROOT, "tests", "golden", "fullsize_30s_f32_heavytail_oracle.npz")
np.savez_compressed(out, ids=ids.astype(np.int32), top1=srt[:, -1].astype(np.float32), top2=srt[:, -2].astype(np.float32),
                    logit_absmax=np.float32(np.abs(lg).max()), mel_frames=np.int32(mel.shape[1]), seed=np.int32(SEED), seconds=np.float32(SECONDS),
                    st_head_sha256=np.frombuffer(dense_head_sha(st), dtype=np.uint8), st_size=np.int64(os.path.getsize(st)),
                    logits_step0=lg[0].astype(np.float32)[:4096], audio_sha256=np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8))
print(f"oracle f32 (heavy-tailed, {SECONDS:g} s): {len(ids)} ids in {dt:.1f} s; |logit| max {float(np.abs(lg).max()):.2f}; min top-2 margin {float((srt[:, -1] - srt[:, -2]).min()):.4g}; "
      f"distinct ids {len(set(ids.tolist()))}; wrote {out}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import shutil; shutil.copy(out, os.path.join(ROOT, "gpurun_out", "fullsize_30s_f32_heavytail_oracle.npz"))
o.close()
