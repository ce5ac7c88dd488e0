"""Synthetic weights / audio in the reference's real on-disk layout.

There is no network and no real ``voxtral-q4.gguf`` in the build or GPU containers, so the
parity tests, ``smoke()`` and ``bench.py`` run on synthetic weights written as a genuine
GGUF v3 file with the exact tensor names, reversed dims, dtypes and 32-byte alignment the
reference loader expects (``src/gguf/reader.rs:105-188``, ``src/gguf/loader.rs:191-491``,
name tables ``src/models/weights.rs:219-397``).  A real GGUF drops in
unchanged.

The Q4_0 codec here (``quantize_q4_0`` / ``dequantize_q4_0``) is the numpy twin of the
reference's test quantiser (``src/gguf/tests.rs:24-87``); it is host-side tooling
(SURVEY.md section 8f item 4), not part of the accelerated path.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, asdict

import numpy as np

ENC = "mm_streams_embeddings.embedding_module.whisper_encoder"
EMB = "mm_streams_embeddings.embedding_module"
TOK = EMB + ".tok_embeddings.weight"
ADP = EMB + ".audio_language_projection"

GGML_F32, GGML_F16, GGML_Q4_0 = 0, 1, 2


@dataclass
class ModelDims:
    """Shapes of the model (defaults: ``src/models/config.rs:441-493``)."""
    enc_layers: int = 32
    enc_dim: int = 1280
    enc_heads: int = 32          # head_dim fixed at 64
    enc_ffn: int = 5120
    dec_layers: int = 26
    dec_dim: int = 3072
    dec_heads: int = 32          # head_dim fixed at 128
    dec_kv_heads: int = 8
    dec_ffn: int = 9216
    vocab: int = 131072
    n_mels: int = 128
    t_cond: int = 32
    enc_bias: bool = True        # wq/wv/wo/w2 biases (gguf/loader.rs:226-232,248-249)

    ENC_HD = 64
    DEC_HD = 128

    def as_dict(self):
        return asdict(self)


def tiny_dims(**kw) -> ModelDims:
    """A small model with the real architecture (for parity tests that must finish in seconds)."""
    d = dict(enc_layers=2, enc_dim=128, enc_heads=2, enc_ffn=256, dec_layers=2, dec_dim=256,
             dec_heads=4, dec_kv_heads=2, dec_ffn=512, vocab=512, n_mels=128, t_cond=32)
    d.update(kw)
    return ModelDims(**d)


def tensor_manifest(d: ModelDims):
    """[(name, shape (PyTorch order), kind, sigma)] -- SURVEY.md Appendix A."""
    out = []
    out.append((f"{ENC}.conv_layers.0.conv.weight", (d.enc_dim, d.n_mels, 3), "f32", 0.05))
    out.append((f"{ENC}.conv_layers.0.conv.bias", (d.enc_dim,), "f32", 0.01))
    out.append((f"{ENC}.conv_layers.1.conv.weight", (d.enc_dim, d.enc_dim, 3), "f32", 0.02))
    out.append((f"{ENC}.conv_layers.1.conv.bias", (d.enc_dim,), "f32", 0.01))
    qd = d.enc_heads * d.ENC_HD
    for i in range(d.enc_layers):
        p = f"{ENC}.transformer.layers.{i}"
        out.append((f"{p}.attention_norm.weight", (d.enc_dim,), "norm", 0.02))
        out.append((f"{p}.ffn_norm.weight", (d.enc_dim,), "norm", 0.02))
        for nm, shape, bias in (("attention.wq", (qd, d.enc_dim), True), ("attention.wk", (qd, d.enc_dim), False),
                                ("attention.wv", (qd, d.enc_dim), True), ("attention.wo", (d.enc_dim, qd), True),
                                ("feed_forward.w1", (d.enc_ffn, d.enc_dim), False),
                                ("feed_forward.w2", (d.enc_dim, d.enc_ffn), True),
                                ("feed_forward.w3", (d.enc_ffn, d.enc_dim), False)):
            out.append((f"{p}.{nm}.weight", shape, "q4", 0.03))
            if bias and d.enc_bias:
                out.append((f"{p}.{nm}.bias", (shape[0],), "f32", 0.01))
    out.append((f"{ENC}.transformer.norm.weight", (d.enc_dim,), "norm", 0.02))
    out.append((f"{ADP}.0.weight", (d.dec_dim, d.enc_dim * 4), "q4", 0.02))
    out.append((f"{ADP}.2.weight", (d.dec_dim, d.dec_dim), "q4", 0.02))
    out.append((TOK, (d.vocab, d.dec_dim), "q4", 0.02))
    for i in range(d.dec_layers):
        p = f"layers.{i}"
        out.append((f"{p}.ada_rms_norm_t_cond.0.weight", (d.t_cond, d.dec_dim), "q4", 0.02))
        out.append((f"{p}.ada_rms_norm_t_cond.2.weight", (d.dec_dim, d.t_cond), "q4", 0.05))
        out.append((f"{p}.attention_norm.weight", (d.dec_dim,), "norm", 0.02))
        out.append((f"{p}.ffn_norm.weight", (d.dec_dim,), "norm", 0.02))
        out.append((f"{p}.attention.wq.weight", (d.dec_heads * d.DEC_HD, d.dec_dim), "q4", 0.02))
        out.append((f"{p}.attention.wk.weight", (d.dec_kv_heads * d.DEC_HD, d.dec_dim), "q4", 0.02))
        out.append((f"{p}.attention.wv.weight", (d.dec_kv_heads * d.DEC_HD, d.dec_dim), "q4", 0.02))
        out.append((f"{p}.attention.wo.weight", (d.dec_dim, d.dec_heads * d.DEC_HD), "q4", 0.02))
        out.append((f"{p}.feed_forward.w1.weight", (d.dec_ffn, d.dec_dim), "q4", 0.02))
        out.append((f"{p}.feed_forward.w2.weight", (d.dec_dim, d.dec_ffn), "q4", 0.02))
        out.append((f"{p}.feed_forward.w3.weight", (d.dec_ffn, d.dec_dim), "q4", 0.02))
    out.append(("norm.weight", (d.dec_dim,), "norm", 0.02))
    return out


# --------------------------------------------------------------------------- Q4_0 codec (numpy)

def quantize_q4_0(data: np.ndarray) -> np.ndarray:
    """Reference test quantiser, ``src/gguf/tests.rs:24-57``: d = amax/7, q = min(15, trunc(v/d + 8.5))."""
    x = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / np.float32(7.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = (x * inv[:, None] + np.float32(8.5)).astype(np.float32)
    q = np.clip(np.trunc(q), 0, 15).astype(np.uint8)  # Rust `as u8` saturates; then .min(15)
    out = np.empty((x.shape[0], 18), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def dequantize_q4_0(raw: np.ndarray, n_elems: int) -> np.ndarray:
    """``src/gguf/tensor.rs:88-113``."""
    b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 18)
    d = b[:, 0:2].copy().view(np.float16).astype(np.float32)  # [nb,1]
    qs = b[:, 2:]
    lo = (qs & 0x0F).astype(np.float32) - np.float32(8.0)
    hi = (qs >> 4).astype(np.float32) - np.float32(8.0)
    out = np.concatenate([lo * d, hi * d], axis=1)
    return out.reshape(-1)[:n_elems]


_NIB_LUT = np.array([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10, 11, 12, 13, 14, 15], dtype=np.uint8)   # symmetric about 8 -> zero-mean weights
_BYTE_LUT = (_NIB_LUT[np.arange(256) & 15] | (_NIB_LUT[np.arange(256) >> 4] << 4)).astype(np.uint8)


def synth_q4_blocks(rng: np.random.Generator, n_elems: int, sigma: float) -> np.ndarray:
    """Random Q4_0 blocks generated directly in the quantised domain (fast enough for the 2.5 GB
    full-size model): nibbles in 1..15 with mean exactly 8 (zero-mean weights, the range the GGML
    quantiser produces), f16 scale d ~ sigma/4.18 * U(0.5, 1.5) so the dequantised weights have
    std ~ sigma."""
    nb = n_elems // 32
    out = np.empty((nb, 18), dtype=np.uint8)
    out[:, 2:] = _BYTE_LUT[rng.bit_generator.random_raw(nb * 2).view(np.uint8)].reshape(nb, 16)   # raw PCG64 stream: ~2 GB/s
    d = (np.float32(sigma / 4.18) * (np.float32(0.5) + rng.random(nb, dtype=np.float32))).astype(np.float16)
    out[:, 0:2] = d.view(np.uint8).reshape(nb, 2)
    return out.reshape(-1)


# --------------------------------------------------------------------------- GGUF writer

def _gguf_str(s: str) -> bytes:
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def write_gguf(path: str, tensors, version: int = 3):
    """tensors: iterable of (name, shape_pytorch_order, ggml_dtype, bytes-like or callable->bytes).
    Layout per ``src/gguf/reader.rs:105-188`` (and the in-test builder ``src/gguf/tests.rs:90-168``):
    magic, version, counts, one KV (general.architecture="voxtral"), tensor infos with REVERSED
    dims, data section at the next 32-byte boundary; each tensor 32-byte aligned."""
    tensors = list(tensors)
    sizes = []
    for name, shape, dt, _ in tensors:
        ne = int(np.prod(shape))
        sizes.append(ne * 4 if dt == GGML_F32 else ne * 2 if dt == GGML_F16 else ne // 32 * 18)
    head = bytearray()
    head += struct.pack("<IIQQ", 0x46554747, version, len(tensors), 1)
    head += _gguf_str("general.architecture") + struct.pack("<I", 8) + _gguf_str("voxtral")
    off = 0
    offsets = []
    for (name, shape, dt, _), sz in zip(tensors, sizes):
        head += _gguf_str(name) + struct.pack("<I", len(shape))
        for dim in reversed(shape):
            head += struct.pack("<Q", int(dim))
        head += struct.pack("<IQ", dt, off)
        offsets.append(off)
        off = (off + sz + 31) // 32 * 32
    head += b"\0" * ((32 - len(head) % 32) % 32)
    with open(path, "wb") as f:
        f.write(head)
        pos = 0
        for (name, shape, dt, data), sz, o in zip(tensors, sizes, offsets):
            if pos < o:
                f.write(b"\0" * (o - pos)); pos = o
            buf = data() if callable(data) else data
            buf = memoryview(np.ascontiguousarray(buf)).cast("B")
            assert len(buf) == sz, (name, len(buf), sz)
            f.write(buf); pos += sz


HEAVY_OUTLIER_CHANNELS = (7, 301, 1024, 1777, 2048, 3000)      # residual-stream channels whose producers (decoder wo / w2 rows) are scaled x50
PEAKED_TOKENS = tuple(1021 + 5003 * i for i in range(int(os.environ.get('VOX_SYNTH_PEAKED_N', '24'))))     # `peaked`: the vocabulary rows whose block scales are multiplied by PEAKED_GAIN
PEAKED_GAIN = float(os.environ.get('VOX_SYNTH_PEAKED_GAIN', '5'))      # (the environment overrides exist for tools/peaked_seed_search.py only)


def write_synthetic_gguf(path: str, dims: ModelDims, seed: int = 42, dense_override=None, heavy_tail: bool = False, peaked: bool = False):
    """Write a synthetic Q4_0 GGUF for ``dims``.  Deterministic in (dims, seed, heavy_tail, peaked).
    ``peaked``: a PEAKED logit distribution -- the block scales of twelve rows of the tied embedding / lm_head matrix (PEAKED_TOKENS) are multiplied by PEAKED_GAIN and
    the final norm weight is centred on 3, so every step's argmax is one of twelve loud tokens with |logit| ~ 30 and a top-2 margin that is a sizeable fraction of the
    largest logit (random Gaussian logits over 131 072 rows put a near-tie into almost every 100-step transcript, which caps what an ids-equal test can assert).
    ``dense_override``: optional {name: float32 array} quantised with the reference quantiser
    instead of the random generator.
    ``heavy_tail``: realistic statistics instead of N(0, sigma^2) -- every Q4 block scale is multiplied by 0.3 + |Student-t(nu=4)| (clipped at 20: outlier
    blocks), the decoder's wo / w2 rows that feed HEAVY_OUTLIER_CHANNELS of the residual stream are scaled x50 (outlier channels in h, as real
    checkpoints have), and the final norm weight is centred on 5 instead of 1 (logits an order of magnitude larger)."""
    man = tensor_manifest(dims)

    def gen(idx, name, shape, kind, sigma):
        def make():
            rng = np.random.default_rng([seed, idx])
            ne = int(np.prod(shape))
            if dense_override is not None and name in dense_override:
                a = np.asarray(dense_override[name], dtype=np.float32).reshape(shape)
                return quantize_q4_0(a) if kind == "q4" else a
            if kind == "q4":
                blk = synth_q4_blocks(rng, ne, sigma)
                if heavy_tail:
                    b = blk.reshape(-1, 18)
                    d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(-1)
                    d *= np.minimum(0.3 + np.abs(rng.standard_t(4, d.size)), 20.0).astype(np.float32)
                    if name.endswith("attention.wo.weight") or name.endswith("feed_forward.w2.weight"):
                        if name.startswith("layers."):                     # decoder only: rows = residual-stream channels
                            nbr = int(shape[1]) // 32
                            for ch in HEAVY_OUTLIER_CHANNELS:
                                if ch < int(shape[0]):
                                    d[ch * nbr:(ch + 1) * nbr] *= 50.0
                    b[:, 0:2] = np.minimum(d, 6.0e4).astype(np.float16).view(np.uint8).reshape(-1, 2)
                if peaked and name == TOK:
                    b = blk.reshape(-1, 18); nbr = int(shape[1]) // 32
                    d = b[:, 0:2].copy().view(np.float16).astype(np.float32).reshape(-1)
                    for tk in PEAKED_TOKENS:
                        if tk < int(shape[0]):
                            d[tk * nbr:(tk + 1) * nbr] *= PEAKED_GAIN
                    b[:, 0:2] = np.minimum(d, 6.0e4).astype(np.float16).view(np.uint8).reshape(-1, 2)
                return blk
            if kind == "norm":
                centre = 5.0 if (heavy_tail and name == "norm.weight") else 3.0 if (peaked and name == "norm.weight") else 1.0
                return (centre + sigma * rng.standard_normal(ne)).astype(np.float32)
            return (sigma * rng.standard_normal(ne)).astype(np.float32)
        return make

    tensors = [(name, shape, GGML_Q4_0 if kind == "q4" else GGML_F32, gen(i, name, shape, kind, sigma))
               for i, (name, shape, kind, sigma) in enumerate(man)]
    write_gguf(path, tensors)
    return path


def synth_audio(seconds: float = 16.0, seed: int = 1234, sample_rate: int = 16000) -> np.ndarray:
    """SURVEY.md section 8(d) clip: 0.3 sin(2 pi 220 t) + 0.2 sin(2 pi (440+30t) t) + 0.05 N(0,1),
    0.5 s fade in/out, float32 in [-1, 1]."""
    n = int(round(seconds * sample_rate))
    t = np.arange(n, dtype=np.float64) / sample_rate
    rng = np.random.default_rng(seed)
    x = 0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.2 * np.sin(2 * np.pi * (440.0 + 30.0 * t) * t) + 0.05 * rng.standard_normal(n)
    fade = min(int(0.5 * sample_rate), n // 2)
    if fade > 0:
        ramp = np.linspace(0.0, 1.0, fade)
        x[:fade] *= ramp
        x[-fade:] *= ramp[::-1]
    return x.astype(np.float32)


def read_gguf_tensors(path: str):
    """Minimal numpy GGUF parser for files written by `write_gguf` (tooling/tests only):
    returns {name: (shape_pytorch_order, ggml_dtype, raw uint8 array)}."""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos); pos += struct.calcsize(fmt); return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal pos
        n = rd("<Q"); s = bytes(buf[pos:pos + n]).decode(); pos += n; return s

    magic, ver, nt, nkv = rd("<IIQQ")
    assert magic == 0x46554747
    for _ in range(nkv):
        rstr(); ty = rd("<I"); assert ty == 8; rstr()
    infos = []
    for _ in range(nt):
        name = rstr(); nd = rd("<I"); dims = [rd("<Q") for _ in range(nd)]; dt = rd("<I"); off = rd("<Q")
        infos.append((name, tuple(reversed(dims)), dt, off))
    data0 = (pos + 31) // 32 * 32
    out = {}
    for name, shape, dt, off in infos:
        ne = int(np.prod(shape)); nb = ne * 4 if dt == 0 else ne * 2 if dt == 1 else ne // 32 * 18
        out[name] = (shape, dt, buf[data0 + off:data0 + off + nb])
    return out


def gguf_dense_f32(path: str):
    """{name: float32 array in PyTorch shape} with Q4_0 tensors dequantised (gguf/tensor.rs:88-113)."""
    out = {}
    for name, (shape, dt, raw) in read_gguf_tensors(path).items():
        ne = int(np.prod(shape))
        if dt == GGML_Q4_0:
            out[name] = dequantize_q4_0(np.asarray(raw), ne).reshape(shape)
        elif dt == GGML_F16:
            out[name] = np.asarray(raw).view(np.float16).astype(np.float32).reshape(shape)
        else:
            out[name] = np.asarray(raw).view(np.float32).reshape(shape).copy()
    return out


# --------------------------------------------------------------------------- f32 / SafeTensors path (BF16 on disk)

def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even f32 -> bf16 bit patterns (uint16)."""
    b = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    return ((b + np.uint32(0x7FFF) + ((b >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << np.uint32(16)).view(np.float32)


def synth_dense_tensors(dims: ModelDims, seed: int = 42):
    """Yield (name, shape, bf16 bit pattern array) for the dense (f32-path) model: the same tensor names and
    shapes as the published ``consolidated.safetensors`` (BF16 on disk, ``src/models/weights.rs:219-397``)."""
    for idx, (name, shape, kind, sigma) in enumerate(tensor_manifest(dims)):
        rng = np.random.default_rng([seed, idx, 77])
        ne = int(np.prod(shape))
        if kind == "norm":
            v = (1.0 + sigma * rng.standard_normal(ne)).astype(np.float32)
        else:
            v = (sigma * rng.standard_normal(ne)).astype(np.float32)
        yield name, shape, f32_to_bf16_bits(v)


def write_safetensors(path: str, tensors):
    """tensors: iterable of (name, shape, dtype_str, bytes-like). Minimal SafeTensors writer (u64 header length,
    JSON header, raw little-endian data) -- the layout ``safetensors`` / ``src/models/weights.rs:170-205`` read."""
    import json
    tensors = list(tensors)
    hdr = {}; off = 0
    for name, shape, dt, data in tensors:
        n = memoryview(np.ascontiguousarray(data)).nbytes
        hdr[name] = {"dtype": dt, "shape": [int(s) for s in shape], "data_offsets": [off, off + n]}
        off += n
    hdr["__metadata__"] = {"format": "pt"}
    hb = json.dumps(hdr, separators=(",", ":")).encode()
    hb += b" " * ((8 - len(hb) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(hb))); f.write(hb)
        for _, _, _, data in tensors:
            f.write(memoryview(np.ascontiguousarray(data)).cast("B"))


def write_synthetic_safetensors(path: str, dims: ModelDims, seed: int = 42):
    write_safetensors(path, ((n, s, "BF16", bits) for n, s, bits in synth_dense_tensors(dims, seed)))
    return path


def write_synthetic_dense_gguf(path: str, dims: ModelDims, seed: int = 42):
    """The same values as ``write_synthetic_safetensors`` as an all-F32 GGUF (what the CPU oracle loads for the f32 path)."""
    write_gguf(path, ((n, s, GGML_F32, bf16_bits_to_f32(bits)) for n, s, bits in synth_dense_tensors(dims, seed)))
    return path


# --------------------------------------------------------------------------- full-size dense checkpoints (fast generator)

def _fast_bf16_bits(rng: np.random.Generator, ne: int, sigma: float) -> np.ndarray:
    """bf16 bit patterns sign | exponent | 7 random mantissa bits: +-[2^-6, 2^-5) (rms 0.023) for linear weights, +-[2^-8, 2^-7) for
    tensors with a small sigma (biases).  One raw PCG64 draw per four values: fast enough for the 4.4 G-parameter model (seconds)."""
    r = rng.bit_generator.random_raw((ne + 3) // 4).view(np.uint16)[:ne]
    return (r & np.uint16(0x807F)) | np.uint16(0x3C80 if sigma >= 0.015 else 0x3B80)


def _bf16_bits_to_f16_bits(b: np.ndarray) -> np.ndarray:
    """Exact for the patterns of `_fast_bf16_bits` (exponents 2^-6 / 2^-8, 7 mantissa bits: all inside f16's normal range)."""
    e = ((b >> np.uint16(7)) & np.uint16(0xFF)).astype(np.int32) - 127 + 15
    assert e.min() >= 1 and e.max() <= 30
    return (b & np.uint16(0x8000)) | (e.astype(np.uint16) << np.uint16(10)) | ((b & np.uint16(0x7F)) << np.uint16(3))


def dense_checkpoint_tensors(dims: ModelDims, seed: int = 7, heavy_tail: bool = False):
    """Yield (name, shape, kind, bf16 bit patterns) of the synthetic dense (f32-path) checkpoint with the published tensor names and
    shapes; linear / conv / bias values come from the fast bit-level generator, norm weights are 1 + N(0, sigma^2) rounded to bf16.
    ``heavy_tail``: the stress statistics of write_synthetic_gguf(heavy_tail=True) in bf16-exact form -- every 32-element block of a 2-D linear is scaled by
    a POWER OF TWO 2^round(log2(0.3 + |Student-t(4)|)) clipped to 2^-2 .. 2^4, the decoder's wo / w2 rows that feed HEAVY_OUTLIER_CHANNELS by 2^6 more, the
    final norm is centred on 5: an exponent add on the bit patterns, so the values stay exact in bf16 AND in the oracle's f16 copy (exponents 2^-8 .. 2^4)."""
    for idx, (name, shape, kind, sigma) in enumerate(tensor_manifest(dims)):
        rng = np.random.default_rng([seed, idx, 99])
        ne = int(np.prod(shape))
        if kind == "norm":
            centre = 5.0 if (heavy_tail and name == "norm.weight") else 1.0
            bits = f32_to_bf16_bits((centre + sigma * rng.standard_normal(ne)).astype(np.float32))
        else:
            bits = _fast_bf16_bits(rng, ne, sigma)
            if heavy_tail and kind == "q4" and len(shape) == 2 and int(shape[1]) % 32 == 0:
                rt = np.random.default_rng([seed, idx, 4])
                k = np.clip(np.rint(np.log2(0.3 + np.abs(rt.standard_t(4, ne // 32)))), -2, 4).astype(np.int32)
                if name.startswith("layers.") and (name.endswith("attention.wo.weight") or name.endswith("feed_forward.w2.weight")):
                    nbr = int(shape[1]) // 32
                    for ch in HEAVY_OUTLIER_CHANNELS:
                        if ch < int(shape[0]):
                            k[ch * nbr:(ch + 1) * nbr] += 6
                bits = (bits.astype(np.int32) + (np.repeat(k, 32) << 7)).astype(np.uint16)      # exponent field += k (base exponents 2^-6: no carry into the sign)
        yield name, shape, kind, bits


def write_fast_dense_checkpoint(st_path: str | None, gguf_path: str | None, dims: ModelDims, seed: int = 7, heavy_tail: bool = False):
    """The SAME synthetic dense model twice: `st_path` = BF16 SafeTensors (what VoxtralModelLoader reads, like the published
    consolidated.safetensors), `gguf_path` = a dense GGUF for the CPU oracle (2-D linears as F16 -- exact, see _bf16_bits_to_f16_bits --
    everything else as F32).  Either path may be None.  Full size: 8.9 GB each, a few seconds per file."""
    import json
    man = tensor_manifest(dims)
    if st_path:
        hdr = {}; off = 0
        for name, shape, kind, sigma in man:
            n = int(np.prod(shape)) * 2
            hdr[name] = {"dtype": "BF16", "shape": [int(x) for x in shape], "data_offsets": [off, off + n]}; off += n
        hdr["__metadata__"] = {"format": "pt"}
        hb = json.dumps(hdr, separators=(",", ":")).encode(); hb += b" " * ((8 - len(hb) % 8) % 8)
        with open(st_path, "wb") as f:
            f.write(struct.pack("<Q", len(hb))); f.write(hb)
            for name, shape, kind, bits in dense_checkpoint_tensors(dims, seed, heavy_tail):
                f.write(np.ascontiguousarray(bits).tobytes())
    if gguf_path:
        def gen(name, shape, kind, bits):
            if kind == "q4" and len(shape) == 2:
                return GGML_F16, _bf16_bits_to_f16_bits(bits)
            return GGML_F32, bf16_bits_to_f32(bits)
        # write_gguf wants the dtype up front: linears (kind q4 in the manifest) are the F16 ones
        it = dense_checkpoint_tensors(dims, seed, heavy_tail)
        def lazy(entry):
            name, shape, kind, sigma = entry
            def make():
                n2, s2, k2, bits = next(it)
                assert n2 == name
                return gen(n2, s2, k2, bits)[1]
            return make
        write_gguf(gguf_path, [(name, shape, GGML_F16 if (kind == "q4" and len(shape) == 2) else GGML_F32, lazy((name, shape, kind, sigma)))
                               for name, shape, kind, sigma in man])
