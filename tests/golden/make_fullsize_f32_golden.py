#!/usr/bin/env python3
"""Generates tests/golden/fullsize_16s_f32_oracle.npz: the CPU oracle's greedy ids and per-step top-2 logits for BASELINE configs[0-1]
(the f32 SafeTensors path, bin/transcribe.rs:362-438) at FULL size -- the synthetic dense model with the real Voxtral-Mini-4B-Realtime
shapes (voxtral synth.write_fast_dense_checkpoint, seed 7: BF16 on disk like the published checkpoint) on the 16 s bench clip (seed 1234),
un-chunked pipeline.  The oracle reads the same values from a dense GGUF (linears F16, exact) and runs them as f32, sequential-k sums
like the reference's CPU path.  It needs a many-core host with > 40 GB of RAM, so it is run once on the GPU box's host CPU
(`gpurun -- python tests/golden/make_fullsize_f32_golden.py`) and the small result is committed;
`tests/test_gpu_fullsize.py::test_full_16s_clip_f32_vs_oracle_golden` replays it.  Nothing here touches the GPU or /root/reference."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 128)))
import oracle_lib as orc
from __graft_entry__ import load_package
from model_fixtures import cache_dir, dense_head_sha
pkg = load_package(); S = pkg.synth
SEED = 7
gg = os.path.join(cache_dir(), f"full_dense_seed{SEED}.gguf"); st = os.path.join(cache_dir(), f"full_dense_seed{SEED}.safetensors")
t0 = time.time()
if not (os.path.exists(gg) and os.path.exists(st)):
    S.write_fast_dense_checkpoint(st + ".tmp", gg + ".tmp", S.ModelDims(), seed=SEED); os.replace(st + ".tmp", st); os.replace(gg + ".tmp", gg)
print(f"checkpoints ready in {time.time() - t0:.1f} s ({os.path.getsize(st) / 1e9:.2f} GB safetensors, {os.path.getsize(gg) / 1e9:.2f} GB gguf)", flush=True)
x = S.synth_audio(16.0, seed=1234)
xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
t = pkg.TimeEmbedding(3072).embed(6.0)
t0 = time.time(); o = orc.Model(gg); print(f"oracle load {time.time() - t0:.1f} s", flush=True)
t0 = time.time(); ids, lg = o.transcribe_streaming(mel, t, want_logits=True); dt = time.time() - t0
srt = np.sort(lg, axis=1)
out = os.path.join(ROOT, "tests", "golden", "fullsize_16s_f32_oracle.npz")
np.savez_compressed(out, ids=ids.astype(np.int32), top1=srt[:, -1].astype(np.float32), top2=srt[:, -2].astype(np.float32),
                    logit_absmax=np.float32(np.abs(lg).max()), mel_frames=np.int32(mel.shape[1]), seed=np.int32(SEED),
                    st_head_sha256=np.frombuffer(dense_head_sha(st), dtype=np.uint8), st_size=np.int64(os.path.getsize(st)),
                    logits_step0=lg[0].astype(np.float32)[:4096], audio_sha256=np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8))
print(f"oracle f32: {len(ids)} ids in {dt:.1f} s; min top-2 margin {float((srt[:, -1] - srt[:, -2]).min()):.4g}; distinct ids {len(set(ids.tolist()))}; wrote {out}", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
import shutil; shutil.copy(out, os.path.join(ROOT, "gpurun_out", "fullsize_16s_f32_oracle.npz"))
o.close()
