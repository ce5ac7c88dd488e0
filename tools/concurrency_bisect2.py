#!/usr/bin/env python3
"""Operator-level companion of tools/concurrency_bisect.py: single operators through the C ABI from two host threads on two contexts of ONE GPU, against the same calls made
one context at a time.  Outputs must be bit-identical.
    python tools/concurrency_bisect2.py [R=8]"""
import os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
import bench
from importlib import import_module
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pkg = load_package(); ctxs = [pkg.Context(0), pkg.Context(0)]
rng = np.random.default_rng(5)
def q4(n, k, c):
    raw = rng.integers(0, 256, size=n * (k // 32) * 18, dtype=np.uint8).reshape(-1, 18); raw[:, 1] = (raw[:, 1] & 0x03) | 0x28; raw[:, 0] = raw[:, 0]      # f16 scales ~ 2^-5 .. 2^-3
    return pkg.Q4Tensor.from_q4_bytes(raw.reshape(-1), [n, k], c)
W = {}
for nm, (n_, k_) in {"qkv": (3840, 1280), "w2": (1280, 5120), "w13": (10240, 1280)}.items():
    r = rng.integers(0, 256, size=n_ * (k_ // 32) * 18, dtype=np.uint8).reshape(-1, 18); r[:, 1] = (r[:, 1] & 0x03) | 0x28
    W[nm] = [pkg.Q4Tensor.from_q4_bytes(r.reshape(-1).copy(), [n_, k_], c) for c in ctxs]
X = {m_: rng.standard_normal((1, m_, 5120), dtype=np.float32) for m_ in (600, 3000)}
qa, ka, va = (rng.standard_normal((600, 2048), dtype=np.float32) for _ in range(3))
TASKS = [
    ("q4_matmul 600 x 1280 -> 3840", lambda k, r: pkg.q4_matmul(X[600][:, :, :1280], W["qkv"][k])),
    ("q4_matmul 600 x 5120 -> 1280", lambda k, r: pkg.q4_matmul(X[600], W["w2"][k])),
    ("q4_matmul 3000 x 1280 -> 10240", lambda k, r: pkg.q4_matmul(X[3000][:, :, :1280], W["w13"][k])),
    ("q4_matmul 3000 x 5120 -> 1280", lambda k, r: pkg.q4_matmul(X[3000], W["w2"][k])),
    ("attention 600 rows, 32 heads of 64, window 750", lambda k, r: import_module(pkg.__name__ + ".gguf").attention(ctxs[k], qa, ka, va, 32, 32, 0, 750)),
]
def bits(a): return np.ascontiguousarray(a).ravel().view(np.uint32)
for name, fn in TASKS:
    def run(k, out):
        for r in range(R): out.append(bits(fn(k, r)))
    ref = [[], []]
    for k in range(2): run(k, []); run(k, ref[k])
    bad = 0; worst = 0.0
    for attempt in range(2):
        got = [[], []]
        th = [threading.Thread(target=run, args=(k, got[k])) for k in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        for k in range(2):
            for a, b in zip(ref[k], got[k]):
                if not np.array_equal(a, b):
                    bad += 1; d = np.abs(a.view(np.float32) - b.view(np.float32)); worst = max(worst, float(np.nanmax(d)))
    print(f"[{name}] {2 * R} calls x 2 attempts: {bad} not bit-identical (largest difference {worst:.3g}, reference magnitude {float(np.abs(ref[0][0].view(np.float32)).mean()):.3g})", flush=True)
# the encoder as a whole, with and without its split-K operators
path = bench.full_gguf_path(pkg, 42, 0, lambda: None)
m = pkg.Q4ModelLoader.from_file(path).load(ctxs[0]); models = [m, m.replicate(ctxs[1])]
mel = pkg.MelSpectrogram.voxtral(ctxs[0])
mels = [np.ascontiguousarray(mel.compute_log(pkg.pad_audio(pkg.synth.synth_audio(12.0, seed=500 + i))).T)[None] for i in range(4)]
for tag, env in (("default", {}), ("VOX_ENC_SPLITK=0", {"VOX_ENC_SPLITK": "0"}), ("VOX_CONV_VALU=1", {"VOX_CONV_VALU": "1"})):
    for a, b in env.items(): os.environ[a] = b
    def run(k, out):
        for r in range(R): out.append(bits(models[k].encode_audio(mels[(k + r) % 4])))
    ref = [[], []]
    for k in range(2): run(k, []); run(k, ref[k])
    bad = 0; worst = 0.0; nbad_el = 0
    for attempt in range(2):
        got = [[], []]
        th = [threading.Thread(target=run, args=(k, got[k])) for k in range(2)]
        for x in th: x.start()
        for x in th: x.join()
        for k in range(2):
            for a, b in zip(ref[k], got[k]):
                if not np.array_equal(a, b):
                    bad += 1; d = np.abs(a.view(np.float32) - b.view(np.float32)); worst = max(worst, float(np.nanmax(d))); nbad_el = max(nbad_el, int((a != b).sum()))
    print(f"[encode_audio, {tag}] {2 * R} calls x 2 attempts: {bad} not bit-identical (largest difference {worst:.3g}, most differing elements {nbad_el} of {ref[0][0].size}, magnitude {float(np.abs(ref[0][0].view(np.float32)).mean()):.3g})", flush=True)
    for a in env: del os.environ[a]
