#!/usr/bin/env python3
"""BASELINE configs[1]: single 16 s clip through the f32 SafeTensors path (dense bf16 weights on device, f32 activations) at the real
Voxtral-Mini-4B-Realtime shapes.  Synthetic BF16 checkpoint (8.9 GB) written to /tmp with a fast bit-level generator."""
import json, os, struct, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); S = pkg.synth
path = "/tmp/vox_bench_full_bf16.safetensors"
if not os.path.exists(path):
    t0 = time.time(); d = S.ModelDims(); man = S.tensor_manifest(d)
    hdr = {}; off = 0
    for name, shape, kind, sigma in man:
        n = int(np.prod(shape)) * 2; hdr[name] = {"dtype": "BF16", "shape": [int(x) for x in shape], "data_offsets": [off, off + n]}; off += n
    hdr["__metadata__"] = {"format": "pt"}
    hb = json.dumps(hdr, separators=(",", ":")).encode(); hb += b" " * ((8 - len(hb) % 8) % 8)
    rng = np.random.default_rng(7)
    with open(path + ".tmp", "wb") as f:
        f.write(struct.pack("<Q", len(hb))); f.write(hb)
        for name, shape, kind, sigma in man:
            ne = int(np.prod(shape))
            if kind == "norm":
                bits = S.f32_to_bf16_bits((1.0 + sigma * rng.standard_normal(ne)).astype(np.float32))
            else:   # +-[2^-6, 2^-5) (rms 0.023) or, for biases/convs with a small sigma, +-[2^-8, 2^-7): sign + 7 random mantissa bits
                r = rng.bit_generator.random_raw((ne + 3) // 4).view(np.uint16)[:ne]
                bits = (r & np.uint16(0x807F)) | np.uint16(0x3C80 if sigma >= 0.015 else 0x3B80)
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
    print(f"f32 path: {dt * 1e3:.1f} ms, {len(ids) / dt:.1f} tok/s e2e, decode {tm['decode_ms']:.1f} ms ({len(ids) / (tm['decode_ms'] / 1e3):.1f} tok/s ref-def), encode {tm['encode_ms']:.1f} ms, rtf {dt / 16.0:.5f}", flush=True)
for which, nm in enumerate(["qkv", "wo", "w1w3", "w2", "lm_head"]):
    us, nbytes, kn = model.bench_decode_gemv(which, 52 if which != 4 else 10)
    print(f"  {nm}: {us:.1f} us, {nbytes / 1e6:.1f} MB, {nbytes / us / 1e6:.2f} TB/s  {kn}", flush=True)
model.close(); ctx.close()
