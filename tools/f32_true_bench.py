#!/usr/bin/env python3
"""The f32 path on a checkpoint that is NOT bf16-representable (`dense2_gemm_kernel` -- bf16 hi + lo weight planes, three MFMAs per product -- is what
such a checkpoint's encoder / prefill GEMMs run on; its wide tile spilled 187-212 VGPRs until round 5): a synthetic full-size F16 SafeTensors checkpoint (8.9 GB, ten random
mantissa bits per weight, so no tensor fits one bf16 plane -> every linear is WFMT_F32), one 16 s clip, stage timings.   python tools/f32_true_bench.py"""
import json, os, struct, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); S = pkg.synth
path = "/tmp/vox_bench_full_f16.safetensors"
if not os.path.exists(path):
    t0 = time.time(); d = S.ModelDims(); man = S.tensor_manifest(d)
    hdr = {}; off = 0
    for name, shape, kind, sigma in man:
        n = int(np.prod(shape)) * 2; hdr[name] = {"dtype": "F16", "shape": [int(x) for x in shape], "data_offsets": [off, off + n]}; off += n
    hdr["__metadata__"] = {"format": "pt"}
    hb = json.dumps(hdr, separators=(",", ":")).encode(); hb += b" " * ((8 - len(hb) % 8) % 8)
    rng = np.random.default_rng(7)
    with open(path + ".tmp", "wb") as f:
        f.write(struct.pack("<Q", len(hb))); f.write(hb)
        for name, shape, kind, sigma in man:
            ne = int(np.prod(shape))
            if kind == "norm":
                bits = (1.0 + sigma * rng.standard_normal(ne)).astype(np.float16).view(np.uint16)
            else:   # +-[2^-6, 2^-5) (rms 0.023) or +-[2^-8, 2^-7): sign + 10 random mantissa bits (f16: exponent bias 15)
                r = rng.bit_generator.random_raw((ne + 3) // 4).view(np.uint16)[:ne]
                bits = (r & np.uint16(0x83FF)) | np.uint16((15 - 6) << 10 if sigma >= 0.015 else (15 - 8) << 10)
            f.write(np.ascontiguousarray(bits).tobytes())
    os.replace(path + ".tmp", path)
    print(f"wrote {path} ({off / 1e9:.2f} GB) in {time.time() - t0:.1f}s", flush=True)
ctx = pkg.Context(0)
t0 = time.time(); model = pkg.VoxtralModelLoader.from_file(path).load(ctx); print(f"load {time.time() - t0:.1f}s, device weights {model.weight_bytes() / 1e9:.2f} GB", flush=True)
t = pkg.TimeEmbedding(3072).embed(6.0)
x = pkg.synth.synth_audio(16.0, seed=1234); dx = ctx.upload(x)
for it in range(4):
    t0 = time.perf_counter(); ids = model.transcribe_audio(None, t, device_ptr=dx, n_samples=x.size); dt = time.perf_counter() - t0
    tm = model.timings()
    print(f"f32 path, F16 checkpoint (exact f32 planes + bf16 hi/lo planes): {dt * 1e3:.1f} ms, {len(ids) / dt:.1f} tok/s e2e, encode {tm['encode_ms']:.2f} ms, decode {tm['decode_ms']:.1f} ms "
          f"({len(ids) / (tm['decode_ms'] / 1e3):.1f} tok/s ref-def)", flush=True)
clips = [pkg.synth.synth_audio(16.0, seed=4321 + i) for i in range(16)]
model.transcribe_batch(clips, t)
t0 = time.perf_counter(); outs = model.transcribe_batch(clips, t); dt = time.perf_counter() - t0; tm = model.timings()
print(f"16 x 16 s clips: {dt * 1e3:.1f} ms, {sum(len(o) for o in outs) / dt:.0f} tok/s, encode {tm['encode_ms']:.1f} ms (the stacked encoder's GEMMs: dense2_gemm_kernel<2,*>, 64 x 128 tiles)", flush=True)
model.close(); ctx.close()
