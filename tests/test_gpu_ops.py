"""GPU parity of the operator boundary (Q4Tensor / q4_matmul / Q4Linear / log-mel) against the CPU oracle.
Every call goes through the C ABI of libvoxtral_hip.so.  Inputs are the reference's own deterministic
sin/cos generators where the reference has a test for the case (gguf/tests.rs, tests/gguf_integration.rs)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


def _w(n, k, f=0.0007, a=0.05, fn=np.cos):
    return (fn(np.arange(n * k, dtype=np.float32) * np.float32(f)) * np.float32(a)).astype(np.float32)


def _act(n, f=0.001, a=0.1):
    return (np.sin(np.arange(n, dtype=np.float32) * np.float32(f)) * np.float32(a)).astype(np.float32)


def test_q4_tensor_dequantize_bit_exact(pkg, orc, ctx):
    # gguf/tests.rs:331-365 (tol 1e-5 there; the repack is lossless so we demand bit-exact)
    w = (np.sin(np.arange(256, dtype=np.float32) * np.float32(0.05) - np.float32(6.4)) * np.float32(0.3)).astype(np.float32)
    q = orc.q4_quantize(w)
    t = pkg.Q4Tensor.from_q4_bytes(q, [8, 32], ctx)       # K must be a whole number of blocks
    assert t.shape() == [8, 32] and t.num_blocks() == 8
    assert (t.dequantize().reshape(-1) == orc.q4_dequantize(q, 256)).all()
    rng = np.random.default_rng(0)
    for n, k in ((16, 32), (64, 96), (3, 3072), (130, 1280)):
        raw = pkg.synth.synth_q4_blocks(rng, n * k, 0.02)
        t = pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx)
        assert t.num_blocks() == n * k // 32
        assert (t.dequantize().reshape(-1) == orc.q4_dequantize(raw, n * k)).all()


def test_q4_tensor_validation(pkg, orc, ctx):
    raw = orc.q4_quantize(np.ones(64, np.float32))
    with pytest.raises(pkg.VoxError, match="byte count mismatch"):          # tensor.rs:43-48
        pkg.Q4Tensor.from_q4_bytes(raw[:-1], [2, 32], ctx)
    with pytest.raises(pkg.VoxError, match="divisible by 32"):              # tensor.rs:38-41
        pkg.Q4Tensor.from_q4_bytes(raw, [3, 11], ctx)
    t = pkg.Q4Tensor.from_q4_bytes(raw, [2, 32], ctx)
    with pytest.raises(pkg.VoxError, match="K="):                           # op.rs:98-100 (panic -> error status)
        pkg.q4_matmul(np.zeros((1, 1, 64), np.float32), t)
    with pytest.raises(pkg.VoxError, match="3-D"):                          # op.rs:92
        pkg.q4_matmul(np.zeros((1, 32), np.float32), t)


def test_q4_matmul_small(pkg, orc, ctx):
    # gguf/tests.rs:371-412, tol 1e-3
    k = n = 32
    w = (np.sin(np.arange(n * k, dtype=np.float32) * np.float32(0.1)) * np.float32(0.5)).astype(np.float32)
    q = orc.q4_quantize(w); act = (np.arange(k, dtype=np.float32) * np.float32(0.1)).reshape(1, 1, k)
    exp = orc.reference_matmul(act.reshape(1, k), orc.q4_dequantize(q, n * k).reshape(n, k))
    out = pkg.q4_matmul(act, pkg.Q4Tensor.from_q4_bytes(q, [n, k], ctx))
    assert out.shape == (1, 1, n) and np.abs(out.reshape(1, n) - exp).max() < 1e-3


@pytest.mark.parametrize("b,s,k,n,tol", [(1, 1, 128, 64, 1e-2), (1, 10, 3072, 3072, 1e-2), (1, 1, 3072, 9216, 1e-2),   # tests.rs:414-478
                                         (4, 10, 128, 64, 1e-3),                                                      # tests.rs:642-694
                                         (1, 5, 64, 64, 1e-2), (1, 8, 128, 64, 1e-2),                                 # gguf_integration.rs:73-148
                                         (1, 1, 3072, 3072, 1e-2), (1, 10, 3072, 9216, 1e-2), (1, 1, 9216, 3072, 1e-2)])  # :150-224
def test_q4_matmul_reference_shapes(pkg, orc, ctx, b, s, k, n, tol):
    act = _act(b * s * k).reshape(b, s, k); w = _w(n, k)
    q = orc.q4_quantize(w)
    exp = orc.q4_matmul(q, n, k, act)
    out = pkg.q4_matmul(act, pkg.Q4Tensor.from_q4_bytes(q, [n, k], ctx))
    assert out.shape == (b, s, n)
    err = np.abs(out - exp).max()
    assert err < tol, err
    assert err < 2e-5 * max(1.0, np.abs(exp).max()), err     # our own, tighter bar: f32-class accuracy


def test_q4_linear_bias(pkg, orc, ctx):
    # gguf/tests.rs:484-562
    i, o = 64, 32
    w = (np.sin(np.arange(o * i, dtype=np.float32) * np.float32(0.001)) * np.float32(0.1)).astype(np.float32)
    q = orc.q4_quantize(w); bias = np.arange(o, dtype=np.float32) * np.float32(0.01)
    act = (np.arange(i, dtype=np.float32) * np.float32(0.1)).reshape(1, 1, i)
    lin = pkg.Q4Linear.new(pkg.Q4Tensor.from_q4_bytes(q, [o, i], ctx), bias)
    out = lin.forward(act)
    assert out.shape == (1, 1, o) and np.abs(out - orc.q4_matmul(q, o, i, act, bias)).max() < 1e-3
    z = pkg.Q4Linear.new(pkg.Q4Tensor.from_q4_bytes(orc.q4_quantize(_w(64, 128, 0.001, 0.1, np.sin)), [64, 128], ctx)).forward(np.zeros((2, 5, 128), np.float32))
    assert z.shape == (2, 5, 64) and (z == 0).all()
    out = lin.forward(np.tile(act, (1, 9, 1)))                # M > 4 -> MFMA path with bias
    assert np.abs(out - orc.q4_matmul(q, o, i, np.tile(act, (1, 9, 1)), bias)).max() < 1e-3


MODEL_SHAPES = [(3072, 6144), (4096, 3072), (3072, 18432), (9216, 3072), (3072, 32), (32, 3072),     # decoder GEMVs
                (1280, 6144), (2048, 1280), (1280, 10240), (5120, 1280), (5120, 3072), (3072, 3072)]  # encoder / adapter


@pytest.mark.parametrize("k,n", MODEL_SHAPES)
@pytest.mark.parametrize("m", [1, 3, 38])
def test_q4_matmul_model_shapes_random(pkg, orc, ctx, k, n, m):
    """Random (not slowly-varying) activations at every (K, N) the model uses; GEMV (m<=4) and MFMA GEMM (m=38)."""
    rng = np.random.default_rng(k * 7 + n + m)
    nn = min(n, 2048)                                            # oracle time; row coverage still spans many workgroups
    raw = pkg.synth.synth_q4_blocks(rng, nn * k, 0.02)
    x = rng.standard_normal((1, m, k)).astype(np.float32)
    exp = orc.q4_matmul(raw, nn, k, x)
    out = pkg.q4_matmul(x, pkg.Q4Tensor.from_q4_bytes(raw, [nn, k], ctx))
    err = np.abs(out - exp).max() / np.abs(exp).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("m,k,n", [(17, 128, 16), (32, 3072, 528), (38, 4096, 3072), (48, 9216, 200), (33, 1280, 2050), (47, 3072, 18432 // 8)])
def test_q4_skinny_mt_prefill_rows(pkg, orc, ctx, m, k, n, monkeypatch):
    """17..48 rows (the 38-token decoder prefill): q4_skinny_mt_kernel -- 2 or 3 m-tiles sharing one weight fetch, split-K over 4 waves;
    ragged N and M, with bias (Q4Linear), against the oracle and against the 32 x 128 MFMA kernel it replaces (VOX_NO_SKINNY_MT=1)."""
    rng = np.random.default_rng(m * 5 + k + n)
    raw = pkg.synth.synth_q4_blocks(rng, n * k, 0.04)
    x = (rng.standard_normal((1, m, k)) * (1 + np.arange(k) / k)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32)
    w = pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx)
    exp = orc.q4_matmul(raw, n, k, x, bias=bias)
    out = pkg.Q4Linear(w, bias).forward(x)
    assert np.abs(out - exp).max() / np.abs(exp).max() < 2e-5
    monkeypatch.setenv("VOX_NO_SKINNY_MT", "1")
    old = pkg.Q4Linear(w, bias).forward(x)
    assert np.abs(out - old).max() / np.abs(exp).max() < 2e-5
    w.close()


@pytest.mark.parametrize("m,k,n", [(5, 32, 16), (17, 64, 48), (64, 128, 64), (65, 96, 80), (146, 5120, 256), (586, 1280, 192)])
def test_q4_gemm_ragged(pkg, orc, ctx, m, k, n):
    """MFMA GEMM edge tiles: M, N not multiples of the 64x64 workgroup tile; transposition-detecting (asymmetric) data."""
    rng = np.random.default_rng(m + k + n)
    raw = pkg.synth.synth_q4_blocks(rng, n * k, 0.05)
    x = (rng.standard_normal((1, m, k)) * (1 + np.arange(k) / k)).astype(np.float32)
    exp = orc.q4_matmul(raw, n, k, x)
    out = pkg.q4_matmul(x, pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx))
    assert np.abs(out - exp).max() / np.abs(exp).max() < 2e-5


def test_q4_matmul_device_pointers(pkg, orc, ctx):
    import ctypes as C
    rng = np.random.default_rng(1); n, k, m = 256, 512, 2
    raw = pkg.synth.synth_q4_blocks(rng, n * k, 0.02); x = rng.standard_normal((m, k)).astype(np.float32)
    t = pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx)
    dx = ctx.upload(x); dy = ctx.alloc(m * n * 4)
    pkg._lib.check(pkg.lib().vox_q4_matmul(ctx.h, t.h, C.c_void_p(dx), 1, m, C.c_void_p(dy), 1))
    ctx.synchronize()
    out = ctx.download(dy, (m, n))
    assert np.abs(out - orc.q4_matmul(raw, n, k, x)).max() < 1e-4
    ctx.free(dx); ctx.free(dy)


def test_log_mel_vs_oracle(pkg, orc, ctx):
    """audio/mel.rs:128-165 on the GPU vs the oracle (f64-accumulated DFT): max|d| <= 1e-4.
    Cases: the reference's own KAT inputs (silence, 440 Hz sine; mel.rs:416-465), noise, a padded clip."""
    mel = pkg.MelSpectrogram.voxtral(ctx)
    t = np.arange(16000) / 16000.0
    sil = mel.compute_log(np.zeros(16000, np.float32))
    assert sil.shape == (100, 128) and np.allclose(sil, (1.5 - 8 + 4) / 4)
    sine = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    lm = mel.compute_log(sine)
    assert lm.shape == (100, 128) and lm.min() >= -2.0 and lm.max() <= 3.0
    assert np.abs(lm - orc.mel_compute_log(sine)).max() < 1e-4
    rng = np.random.default_rng(2)
    for n in (400, 1280 * 3, 16000 * 2 + 37):
        x = (0.1 * rng.standard_normal(n)).astype(np.float32)
        a = mel.compute_log(x); b = orc.mel_compute_log(x)
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-4, n
    clip = pkg.pad_audio(pkg.peak_normalize(pkg.synth.synth_audio(2.0, seed=5)))
    a = mel.compute_log(clip); b = orc.mel_compute_log(clip)
    assert a.shape == (clip.size // 160, 128) and np.abs(a - b).max() < 1e-4
    assert mel.compute_log(np.zeros(100, np.float32)).shape == (0, 128)


def test_log_mel_vs_reference_python_golden(pkg, ctx):
    """The HIP mel kernel against the reference's own PyTorch front-end (scripts/test_proper_inference.py:62-98 compute_mel), fixture from
    tests/golden/make_golden.py; tolerance 5e-4 absolute on the log-mel (<0.1 % of the values above 1e-4: torch's f32 FFT rounds at low-power bins) (the reference's own fixture tolerance is 1e-2, mel.rs:608-613)."""
    from model_fixtures import golden
    g = golden()
    out = pkg.MelSpectrogram.voxtral(ctx).compute_log(g["in_mel_audio"])
    assert out.shape == g["out_log_mel"].T.shape
    d = np.abs(out - g["out_log_mel"].T)
    assert d.max() < 5e-4 and (d > 1e-4).mean() < 1e-3, d.max()


@pytest.mark.parametrize("m,k,n", [(16, 128, 16), (16, 128, 384), (7, 256, 130), (16, 3072, 512), (9, 9216, 96), (16, 4096, 40), (5, 256, 1000), (13, 1280, 2048), (16, 5120, 272)])
def test_q4_skinny_batched_decode_gemm(pkg, orc, ctx, m, k, n):
    """5..16 rows (one per sequence of a decode batch) x K % 128 == 0: the skinny MFMA kernel (split-K across waves,
    in-register 4x4 dword transpose of the weight fragments); ragged N, asymmetric data."""
    rng = np.random.default_rng(m * 3 + k + n)
    raw = pkg.synth.synth_q4_blocks(rng, n * k, 0.05)
    x = (rng.standard_normal((1, m, k)) * (1 + np.arange(k) / k)).astype(np.float32)
    x[0, :, ::7] *= 3.0
    exp = orc.q4_matmul(raw, n, k, x)
    out = pkg.q4_matmul(x, pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx))
    assert np.abs(out - exp).max() / np.abs(exp).max() < 2e-5
    bias = rng.standard_normal(n).astype(np.float32)
    outb = pkg.Q4Linear.new(pkg.Q4Tensor.from_q4_bytes(raw, [n, k], ctx), bias).forward(x)
    assert np.abs(outb - (exp + bias)).max() / np.abs(exp).max() < 2e-5


@pytest.mark.parametrize("M,kv,H,KV,hd,off,win", [
    (16, 16, 2, 2, 64, 0, -1),          # one partial tile (tiny-model encoder shape)
    (1, 1, 2, 1, 128, 0, -1),           # single query / key
    (70, 70, 4, 4, 64, 0, 750),         # two query blocks, ragged tail
    (200, 200, 3, 3, 64, 0, 20),        # sliding window much smaller than the sequence (masking.rs:26-44)
    (38, 38, 8, 2, 128, 0, 8192),       # decoder prefill shape: GQA 4:1 (gguf/model.rs:177-197)
    (5, 133, 4, 2, 128, 128, -1),       # continuation: 5 new queries against a 133-long cache (offset > 0)
    (130, 190, 2, 1, 64, 60, 64),       # offset + window + GQA + several key tiles
])
def test_attention_vs_oracle(pkg, orc, ctx, M, kv, H, KV, hd, off, win):
    """MFMA flash attention (hi/lo bf16 split) against the oracle's f32 score path; tolerance: |d| <= 2e-5 * max|ref| (f32-class)."""
    from importlib import import_module
    attention = import_module(pkg.__name__ + ".gguf").attention
    rng = np.random.default_rng(M * 1000 + kv)
    q = (rng.standard_normal((M, H * hd)) * 1.5).astype(np.float32)
    k = (rng.standard_normal((kv, KV * hd)) * 1.5).astype(np.float32)
    v = rng.standard_normal((kv, KV * hd)).astype(np.float32)
    ref = np.zeros((M, H * hd), np.float32)
    orc.lib().orc_attention(q, k, v, M, kv, H, KV, hd, off, 1, win, ref)
    out = attention(ctx, q, k, v, H, KV, offset=off, window=win)
    err = np.abs(out - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err
    with pytest.raises(pkg.VoxError):
        attention(ctx, q, k, v, H, KV, offset=kv, window=win)       # queries outside the key range


def test_attention_block_vs_reference_python_component_golden(pkg, orc, ctx):
    """The HIP attention core inside the reference's OWN per-component vector (scripts/reference_forward.py test_attention, run on synthetic weights of the real shapes by
    tests/golden/make_component_golden.py -- what the reference's Rust test_attention_vs_reference loads): projections, RoPE and the output projection through the oracle's
    f32 operators (pinned to the same vector on CPU, tests/test_oracle_components.py), the attention itself on the GPU."""
    from importlib import import_module
    from model_fixtures import component_weight
    attention = import_module(pkg.__name__ + ".gguf").attention
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_components.npz"))
    ENC = "mm_streams_embeddings.embedding_module.whisper_encoder."
    x = g["attn_input"][0]; S, H, hd = x.shape[0], 32, 64; L = orc.lib()
    w = {k: component_weight(ENC + f"transformer.layers.0.attention.{k}.weight") for k in ("wq", "wk", "wv", "wo")}
    b = {k: component_weight(ENC + f"transformer.layers.0.attention.{k}.bias") for k in ("wq", "wv", "wo")}
    q = np.ascontiguousarray((orc.reference_matmul(x, w["wq"]) + b["wq"]).reshape(S, H, hd)); k = np.ascontiguousarray(orc.reference_matmul(x, w["wk"]).reshape(S, H, hd))
    v = orc.reference_matmul(x, w["wv"]) + b["wv"]
    L.orc_rope(q, S, H, hd, 0, 1e6); L.orc_rope(k, S, H, hd, 0, 1e6)
    att = attention(ctx, q.reshape(S, H * hd), k.reshape(S, H * hd), v, H, H)
    out = orc.reference_matmul(att, w["wo"]) + b["wo"]
    err = np.abs(out - g["attn_output"][0]).max() / np.abs(g["attn_output"][0]).max()
    assert err < 2e-4, err


def _component_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_components.npz"))


def test_swiglu_hip_operators_vs_reference_python_component_golden(pkg, ctx):
    """The reference's own SwiGLU vector (scripts/reference_forward.py:86-88 run on synthetic weights of the real encoder shapes; what models/layers/swiglu.rs:101
    test_swiglu_vs_reference loads) through the HIP operators of the f32 path: w1 | w3 as ONE interleaved dense operand with the fused SiLU(gate) * up epilogue
    (the product's own FFN form), then w2 -- 10 rows x 1280 -> 5120 -> 1280."""
    from model_fixtures import component_weight, rel_err
    g = _component_golden(); ENC = "mm_streams_embeddings.embedding_module.whisper_encoder."
    w1, w2, w3 = (component_weight(ENC + f"transformer.layers.0.feed_forward.w{i}.weight") for i in (1, 2, 3))
    w13 = pkg.Q4Tensor.from_f32(w1, ctx, other=w3); t2 = pkg.Q4Tensor.from_f32(w2, ctx)
    h = pkg.linear_forward(w13, g["swiglu_input"], epilogue=2)
    assert h.shape == (1, 10, 5120)
    out = pkg.linear_forward(t2, h)
    e = rel_err(out[0], g["swiglu_output"][0]); print(f"SwiGLU (HIP dense operators) vs the reference's vector: {e:.2e}")
    assert e < 2e-4
    w13.close(); t2.close()


def test_conv_downsampler_hip_vs_reference_python_component_golden(pkg, ctx):
    """The reference's own ConvDownsampler vector (reference_forward.py:185-192; models/layers/conv.rs:78-83): 128 x 100 mel frames -> 1280 x 25 through the conv stem's
    im2col GEMMs on the matrix cores (dense2_gemm_kernel, bf16 hi + lo weight planes, GELU epilogue) -- the path encode_audio runs."""
    from model_fixtures import component_weight, rel_err
    g = _component_golden(); ENC = "mm_streams_embeddings.embedding_module.whisper_encoder."
    w1, b1 = component_weight(ENC + "conv_layers.0.conv.weight"), component_weight(ENC + "conv_layers.0.conv.bias")
    w2, b2 = component_weight(ENC + "conv_layers.1.conv.weight"), component_weight(ENC + "conv_layers.1.conv.bias")
    out = pkg.conv_downsample(ctx, g["conv_input"][0], w1, b1, w2, b2)
    assert out.shape == g["conv_output"][0].shape == (1280, 25)
    e = rel_err(out, g["conv_output"][0]); print(f"conv downsampler (HIP im2col MFMA path) vs the reference's vector: {e:.2e}")
    assert e < 2e-4


def test_ada_modulation_hip_vs_reference_python_component_golden(pkg, ctx):
    """The reference's own Ada-modulation vector (reference_forward.py:308-317; models/layers/rms_norm.rs:109-118): scale = w2 gelu(w0 t) through the HIP single-row
    operators with the fused GELU epilogue (the launches vox_model_set_t_embed issues), then x * (1 + scale)."""
    from model_fixtures import component_weight, rel_err
    g = _component_golden()
    w0, w2 = component_weight("layers.0.ada_rms_norm_t_cond.0.weight"), component_weight("layers.0.ada_rms_norm_t_cond.2.weight")
    t0 = pkg.Q4Tensor.from_f32(w0, ctx); t2 = pkg.Q4Tensor.from_f32(w2, ctx)
    hid = pkg.linear_forward(t0, g["ada_rms_norm_t_embed"], epilogue=1)
    scale = pkg.linear_forward(t2, hid)[0]
    assert rel_err(scale, g["ada_rms_norm_scale"][0]) < 2e-4
    assert rel_err(g["ada_rms_norm_input"][0] * (1.0 + scale), g["ada_rms_norm_output"][0]) < 2e-4
    t0.close(); t2.close()


def test_attention_f32_switch_vs_oracle(pkg, orc, ctx, monkeypatch):
    """VOX_ATTN_F32=1 (the all-f32 VALU attention kept beside the MFMA kernel as a cross-check) on the decoder-prefill and a windowed encoder shape."""
    from importlib import import_module
    attention = import_module(pkg.__name__ + ".gguf").attention
    monkeypatch.setenv("VOX_ATTN_F32", "1")
    for (M, kv, H, KV, hd, off, win) in [(38, 38, 8, 2, 128, 0, 8192), (130, 190, 2, 1, 64, 60, 64)]:
        rng = np.random.default_rng(M)
        q = (rng.standard_normal((M, H * hd)) * 1.5).astype(np.float32); k = (rng.standard_normal((kv, KV * hd)) * 1.5).astype(np.float32)
        v = rng.standard_normal((kv, KV * hd)).astype(np.float32)
        ref = np.zeros((M, H * hd), np.float32)
        orc.lib().orc_attention(q, k, v, M, kv, H, KV, hd, off, 1, win, ref)
        out = attention(ctx, q, k, v, H, KV, offset=off, window=win)
        assert np.abs(out - ref).max() / np.abs(ref).max() < 2e-5
    monkeypatch.delenv("VOX_ATTN_F32")


@pytest.mark.parametrize("m,k,n", [(1001, 1280, 6144), (2344, 2048, 1280 + 48), (700, 5120, 10240)])
def test_q4_gemm_big_kernel_matches_tile_kernel(pkg, orc, ctx, monkeypatch, m, k, n):
    """Large-M MFMA GEMM (64x256 tiles, tile-ordered weights, bit-trick B fragments + -136 correction MFMA) against the 32x128 kernel on
    ragged M / N edges, and a row sample against the oracle's sequential-k f32 sums."""
    rng = np.random.default_rng(m + n)
    q = pkg.synth.synth_q4_blocks(rng, n * k, 0.02); w = pkg.Q4Tensor.from_q4_bytes(q, [n, k], ctx)
    x = rng.standard_normal((1, m, k)).astype(np.float32)
    monkeypatch.setenv("VOX_GEMM_BIG", "1"); big = pkg.q4_matmul(x, w)
    monkeypatch.setenv("VOX_GEMM_BIG", "-1"); old = pkg.q4_matmul(x, w)
    monkeypatch.delenv("VOX_GEMM_BIG")
    scale = np.abs(old).max()
    assert np.abs(big - old).max() < 2e-5 * scale
    rows = [0, 63, 64, m - 1]
    ref = orc.q4_matmul(q, n, k, x[:, rows])
    assert np.abs(big[:, rows] - ref).max() < 2e-5 * scale
    w.close()
