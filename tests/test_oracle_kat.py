"""Pin the CPU oracle against the known-answer tests the reference holds in-tree
(SURVEY.md section 8c).  Each test cites the reference test it restates.  CPU only."""
import ctypes as C
import math

import numpy as np
import pytest


# ----------------------------------------------------------------------------- audio/io.rs
def test_peak_normalize(orc):
    # audio/io.rs:202-230: max becomes 0.95 +- 1e-6; silent buffer unchanged
    x = np.array([0.1, -0.5, 0.25, 0.0], dtype=np.float32)
    orc.lib().orc_peak_normalize(x, x.size, 0.95)
    assert abs(np.abs(x).max() - 0.95) < 1e-6
    assert abs(x[0] - 0.19) < 1e-6
    z = np.zeros(16, dtype=np.float32)
    orc.lib().orc_peak_normalize(z, z.size, 0.95)
    assert (z == 0).all()


# ----------------------------------------------------------------------------- audio/pad.rs
def test_pad_kats(orc):
    L = orc.lib(); cfg = orc.PadCfg(); L.orc_pad_cfg_voxtral(C.byref(cfg))
    # pad.rs:115-130 defaults
    assert (cfg.sample_rate, cfg.n_left_pad_tokens, cfg.extra_right_pad_tokens) == (16000, 76, 17)
    assert L.orc_pad_samples_per_token(C.byref(cfg)) == 1280
    assert L.orc_pad_left_samples(C.byref(cfg)) == 97280
    # pad.rs:150-175: right_pad(12800)=21760, (12900)=1180+21760, (12801)=1279+21760
    assert L.orc_pad_right_samples(C.byref(cfg), 12800) == 21760
    assert L.orc_pad_right_samples(C.byref(cfg), 12900) == 1180 + 21760
    assert L.orc_pad_right_samples(C.byref(cfg), 12801) == 1279 + 21760
    # pad.rs:178-218: 255168 samples -> left 97280, right 832+21760, total 375040 = 293 tokens
    n = 255168
    assert L.orc_pad_len(C.byref(cfg), n) == 375040 == 293 * 1280
    x = np.ones(n, dtype=np.float32)
    y = orc.pad_audio(x)
    assert y.size == 375040 and (y[:97280] == 0).all() and (y[97280:97280 + n] == 1).all() and (y[97280 + n:] == 0).all()
    # the 16 s clip of the published metric
    assert L.orc_pad_len(C.byref(cfg), 256000) == 375040
    assert L.orc_mel_num_frames(375040) == 2344


# ----------------------------------------------------------------------------- audio/chunk.rs
def _plan(orc, n, mx, ov=0):
    cfg = orc.ChunkCfg(mx, 160, 16000, ov)
    cnt = orc.lib().orc_chunk_plan(n, C.byref(cfg), None, 0)
    arr = (orc.Chunk * max(cnt, 1))()
    orc.lib().orc_chunk_plan(n, C.byref(cfg), arr, cnt)
    return [(arr[i].start_sample, arr[i].end_sample, bool(arr[i].is_last)) for i in range(cnt)], cfg


def test_chunk_kats(orc):
    # chunk.rs:185-265: 500000 samples @1500 frames -> starts 0 / 240000 / 480000
    plan, cfg = _plan(orc, 500000, 1500)
    assert [p[0] for p in plan] == [0, 240000, 480000]
    assert plan[-1] == (480000, 500000, True) and not plan[0][2]
    assert orc.lib().orc_needs_chunking(500000, C.byref(cfg)) == 1
    assert orc.lib().orc_needs_chunking(240000, C.byref(cfg)) == 0
    # overlap 100 frames -> step 224000
    plan, _ = _plan(orc, 500000, 1500, 100)
    assert plan[1][0] == 224000
    # CLI default 1200 frames on the 16 s clip -> [192000, 64000]
    plan, _ = _plan(orc, 256000, 1200)
    assert [(a, b) for a, b, _ in plan] == [(0, 192000), (192000, 256000)]
    assert _plan(orc, 0, 1500)[0] == []


# ----------------------------------------------------------------------------- audio/mel.rs
def test_hann_and_mel_scale(orc):
    L = orc.lib()
    w4 = np.zeros(4, np.float32); L.orc_hann_window(4, w4)
    assert np.allclose(w4, [0, .5, 1, .5], atol=1e-6)                      # mel.rs:384-394
    w = np.zeros(400, np.float32); L.orc_hann_window(400, w)
    assert w[0] == 0 and abs(w[1] - 6.1690807e-05) < 1e-8                   # mel.rs:396-405
    assert abs(w[200] - 1.0) < 1e-6
    for hz in (100.0, 1000.0, 8000.0):                                       # mel.rs:468-483
        assert abs(L.orc_mel_to_hz(L.orc_hz_to_mel(hz)) - hz) < hz * 1e-4 + 0.1
    assert abs(L.orc_hz_to_mel(1000.0) - 15.0) < 1e-4


def test_filterbank(orc):
    fb = np.zeros((128, 201), np.float32); orc.lib().orc_mel_filterbank(fb)  # mel.rs:376-382 dims
    assert (fb >= 0).all() and (fb.sum(axis=1) > 0).all()
    # independent float64 restatement of Slaney (librosa.filters.mel, htk=False, norm='slaney')
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64); m = f / (200.0 / 3)
        return np.where(f >= 1000, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27), m)
    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27) * (m - 15.0)), m * 200.0 / 3)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), 130)); freqs = np.arange(201) * 40.0
    ref = np.zeros((128, 201))
    for i in range(128):
        lo, ce, up = pts[i:i + 3]
        ref[i] = np.maximum(0, np.minimum((freqs - lo) / (ce - lo), (up - freqs) / (up - ce))) * 2.0 / (up - lo)
    assert np.abs(fb - ref).max() < 1e-3                                      # tolerance of mel.rs:485-520


def test_mel_kats(orc):
    L = orc.lib()
    assert 99 <= L.orc_mel_num_frames(16000) <= 101                          # mel.rs:407-414
    sil = orc.mel_compute(np.zeros(16000, np.float32))                       # mel.rs:416-428
    assert sil.shape == (100, 128) and (sil < 1e-6).all()
    t = np.arange(16000) / 16000.0
    lm = orc.mel_compute_log((0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32))   # mel.rs:430-465
    assert lm.shape == (100, 128) and np.isfinite(lm).all() and lm.min() >= -2.0 and lm.max() <= 3.0
    assert abs(lm.min() - (1.5 - 8 + 4) / 4) < 1e-6 or lm.min() > (1.5 - 8 + 4) / 4
    # the energy peak sits in the mel bin containing 440 Hz
    peak = int(lm.mean(axis=0).argmax())
    fb = np.zeros((128, 201), np.float32); L.orc_mel_filterbank(fb)
    assert fb[peak, 11] > 0                                                  # bin 11 = 440 Hz


def test_mel_vs_torch_stft(orc):
    """Cross-check of the STFT framing/reflect-pad/window/`[:-1]` convention against torch.stft,
    the convention the reference follows (mel.rs:175-212; scripts/test_proper_inference.py:64-98)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    x = (0.1 * rng.standard_normal(16000 * 2 + 37)).astype(np.float32)
    fb = np.zeros((128, 201), np.float32); orc.lib().orc_mel_filterbank(fb)
    st = torch.stft(torch.from_numpy(x), 400, 160, window=torch.hann_window(400), return_complex=True)
    mag = st[..., :-1].abs() ** 2
    ref = (torch.from_numpy(fb).double() @ mag.double()).T.numpy()
    got = orc.mel_compute(x)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    lg = np.maximum(np.log10(np.maximum(ref, 1e-10)), 1.5 - 8.0)
    assert np.abs(orc.mel_compute_log(x) - (lg + 4.0) / 4.0).max() < 1e-4


# ----------------------------------------------------------------------------- time embedding
def test_time_embedding(orc):
    e = orc.time_embedding(1.0, 4, 10000.0)                                  # time_embedding.rs:91-128
    assert np.allclose(e, [math.cos(1), math.cos(0.01), math.sin(1), math.sin(0.01)], atol=1e-5)
    e0 = orc.time_embedding(0.0, 3072)
    assert np.allclose(e0[:1536], 1) and np.allclose(e0[1536:], 0)
    e6 = orc.time_embedding(6.0, 3072)
    assert e6.shape == (3072,) and np.abs(e6).max() <= 1.0


# ----------------------------------------------------------------------------- q4 codec
def test_f16_conversion(orc):
    L = orc.lib(); rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1, 100, 7e4)]
                          + [np.array([0, -0.0, 65504, 65519.99, 65520, 5.96e-8, 2.98e-8, 2.9802325e-8, 6.1e-5], np.float32)])
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16)
    for v, r in zip(vals, ref):
        h = L.orc_f32_to_f16(float(v))
        assert h == int(r.view(np.uint16)), (v, h, int(r.view(np.uint16)))
    for h in list(range(0, 0x7c00, 37)) + [0x8001, 0xbc00, 0x0001, 0x03ff, 0x0400]:
        assert L.orc_f16_to_f32(h) == float(np.uint16(h).view(np.float16).astype(np.float32))


def test_q4_block(orc):
    # gguf/tests.rs:191-227
    original = ((np.arange(32, dtype=np.float32) - 15.5) / 15.5).astype(np.float32)
    q = orc.q4_quantize(original)
    assert q.size == 18
    d = float(q[:2].copy().view(np.float16)[0]); assert abs(d - np.abs(original).max() / 7.0) < 0.01
    deq = orc.q4_dequantize(q, 32)
    assert np.abs(deq - original).max() < 0.08
    # gguf/tests.rs:229-274 edge cases
    assert (orc.q4_dequantize(orc.q4_quantize(np.zeros(32, np.float32)), 32) == 0).all()
    u = np.full(32, 0.5, np.float32); assert np.abs(orc.q4_dequantize(orc.q4_quantize(u), 32) - u).max() < 0.08
    large = ((np.arange(32, dtype=np.float32) - 15.5) * 100).astype(np.float32)
    dl = np.abs(large).max() / 7
    assert np.abs(orc.q4_dequantize(orc.q4_quantize(large), 32) - large).max() < dl / 2 + 1.0


def test_q4_codec_matches_numpy_twin(orc, pkg):
    rng = np.random.default_rng(5)
    w = (rng.standard_normal(64 * 96) * 0.05).astype(np.float32)
    w[:32] = 0
    assert (orc.q4_quantize(w) == pkg.synth.quantize_q4_0(w)).all()
    raw = pkg.synth.synth_q4_blocks(rng, 4096, 0.02)
    assert (orc.q4_dequantize(raw, 4096) == pkg.synth.dequantize_q4_0(raw, 4096)).all()


def _sin_weights(n, k, f=0.0007, a=0.05, fn=np.cos):
    return (fn(np.arange(n * k, dtype=np.float32) * np.float32(f)) * np.float32(a)).astype(np.float32)


def test_q4_matmul_small(orc):
    # gguf/tests.rs:371-412: 32x32, M=1, tol 1e-3 vs reference_matmul(dequant)
    k = n = 32
    w = (np.sin(np.arange(n * k, dtype=np.float32) * np.float32(0.1)) * np.float32(0.5)).astype(np.float32)
    q = orc.q4_quantize(w); wd = orc.q4_dequantize(q, n * k).reshape(n, k)
    act = (np.arange(k, dtype=np.float32) * np.float32(0.1)).reshape(1, k)
    exp = orc.reference_matmul(act, wd)
    got = orc.q4_matmul(q, n, k, act)
    assert np.abs(got - exp).max() < 1e-3
    assert (got == exp).all()  # the oracle's fused path is value-identical to dequant + reference_matmul


@pytest.mark.parametrize("b,s,k,n,tol", [(1, 1, 128, 64, 1e-2), (1, 10, 3072, 3072, 1e-2), (1, 1, 3072, 9216, 1e-2),
                                         (4, 10, 128, 64, 1e-3)])
def test_q4_matmul_shapes(orc, b, s, k, n, tol):
    # gguf/tests.rs:414-478 (shapes) and :642-694 (batch); inputs are the reference's sin/cos generators
    act = (np.sin(np.arange(b * s * k, dtype=np.float32) * np.float32(0.001)) * np.float32(0.1)).astype(np.float32)
    w = _sin_weights(n, k)
    q = orc.q4_quantize(w); wd = orc.q4_dequantize(q, n * k).reshape(n, k)
    got = orc.q4_matmul(q, n, k, act.reshape(b, s, k))
    exp = (act.reshape(-1, k).astype(np.float64) @ wd.T.astype(np.float64)).reshape(b, s, n)
    assert got.shape == (b, s, n) and np.abs(got - exp).max() < tol


def test_q4_linear_bias(orc):
    # gguf/tests.rs:506-562
    i, o = 64, 32
    w = (np.sin(np.arange(o * i, dtype=np.float32) * np.float32(0.001)) * np.float32(0.1)).astype(np.float32)
    q = orc.q4_quantize(w); wd = orc.q4_dequantize(q, o * i).reshape(o, i)
    bias = (np.arange(o, dtype=np.float32) * np.float32(0.01)); act = (np.arange(i, dtype=np.float32) * np.float32(0.1)).reshape(1, i)
    exp = orc.reference_matmul(act, wd) + bias
    assert np.abs(orc.q4_matmul(q, o, i, act, bias) - exp).max() < 1e-3


# ----------------------------------------------------------------------------- gguf reader
def test_gguf_roundtrip(orc, pkg, tmp_path):
    # gguf/tests.rs:280-325: v3, 1 KV, Q4_0 dtype 2, dims as given; multi-tensor lookups
    L = orc.lib(); S = pkg.synth
    w = np.sin(np.arange(32 * 64, dtype=np.float32) * np.float32(0.001) - 1.0).astype(np.float32)
    q = orc.q4_quantize(w)
    p = str(tmp_path / "one.gguf")
    # the reference builder writes dims verbatim [32, 64]; our writer reverses PyTorch order -> pass (64, 32)
    S.write_gguf(p, [("test.weight", (64, 32), S.GGML_Q4_0, q)])
    g = L.orc_gguf_open(p.encode()); assert g, orc.err()
    assert L.orc_gguf_version(g) == 3 and L.orc_gguf_tensor_count(g) == 1
    dims = (C.c_uint64 * 4)(); nd = C.c_uint32(); dt = C.c_uint32(); nb = C.c_uint64()
    assert L.orc_gguf_tensor_info(g, b"test.weight", C.byref(dims), C.byref(nd), C.byref(dt), C.byref(nb)) == 0
    assert list(dims)[:2] == [32, 64] and nd.value == 2 and dt.value == 2 and nb.value == q.size
    ptr = L.orc_gguf_tensor_data(g, b"test.weight")
    assert bytes((C.c_uint8 * q.size).from_address(ptr)) == q.tobytes()
    assert L.orc_gguf_tensor_info(g, b"nonexistent", C.byref(dims), C.byref(nd), C.byref(dt), C.byref(nb)) != 0
    L.orc_gguf_close(g)
    a, b_, c = (orc.q4_quantize(np.full(n, v, np.float32)) for n, v in ((1024, .1), (2048, .2), (2048, -.1)))
    p3 = str(tmp_path / "three.gguf")
    S.write_gguf(p3, [("weight_a", (32, 32), 2, a), ("weight_b", (32, 64), 2, b_), ("weight_c", (64, 32), 2, c)])
    g = L.orc_gguf_open(p3.encode()); assert L.orc_gguf_tensor_count(g) == 3
    for nm, ref in ((b"weight_a", a), (b"weight_b", b_), (b"weight_c", c)):
        ptr = L.orc_gguf_tensor_data(g, nm)
        assert bytes((C.c_uint8 * ref.size).from_address(ptr)) == ref.tobytes()
    L.orc_gguf_close(g)
    bad = tmp_path / "bad.gguf"; bad.write_bytes(b"NOPE" + b"\0" * 64)
    assert not L.orc_gguf_open(str(bad).encode()) and "magic" in orc.err()


# ----------------------------------------------------------------------------- attention / cache semantics
def test_attention_cached_equals_full(orc):
    # models/layers/attention.rs:429-474: chunks 3+2 through the cache == full causal pass, <= 1e-5
    rng = np.random.default_rng(1); S, H, hd = 5, 4, 16
    q, k, v = (rng.standard_normal((S, H, hd)).astype(np.float32) for _ in range(3))
    full = np.zeros((S, H * hd), np.float32)
    orc.lib().orc_attention(q, k, v, S, S, H, H, hd, 0, 1, -1, full)
    a = np.zeros((3, H * hd), np.float32); orc.lib().orc_attention(q[:3].copy(), k[:3].copy(), v[:3].copy(), 3, 3, H, H, hd, 0, 1, -1, a)
    b = np.zeros((2, H * hd), np.float32); orc.lib().orc_attention(q[3:].copy(), k, v, 2, 5, H, H, hd, 3, 1, -1, b)
    assert np.abs(np.concatenate([a, b]) - full).max() <= 1e-5


def test_sliding_window_visibility(orc):
    # masking.rs:26-44: |i-j| > window masked => window+1 keys visible
    S, hd, w = 8, 8, 2
    q = np.zeros((S, 1, hd), np.float32); k = np.zeros((S, 1, hd), np.float32)
    v = np.eye(S, hd, dtype=np.float32).reshape(S, 1, hd)
    out = np.zeros((S, hd), np.float32)
    orc.lib().orc_attention(q, k, v, S, S, 1, 1, hd, 0, 1, w, out)
    for i in range(S):
        vis = [j for j in range(S) if j <= i and i - j <= w]
        exp = np.zeros(hd, np.float32); exp[vis] = 1.0 / len(vis)
        assert np.allclose(out[i], exp, atol=1e-6), i


def test_config_constants(orc, pkg, tmp_path):
    # models/config.rs tests (:560-766), decoder.rs:538-549, encoder.rs:241-252, gguf/loader.rs:567-591
    d = pkg.synth.ModelDims()
    assert (d.enc_layers, d.enc_dim, d.enc_heads, d.ENC_HD, d.enc_ffn) == (32, 1280, 32, 64, 5120)
    assert (d.dec_layers, d.dec_dim, d.dec_heads, d.dec_kv_heads, d.DEC_HD, d.dec_ffn, d.vocab) == (26, 3072, 32, 8, 128, 9216, 131072)
    man = pkg.synth.tensor_manifest(d)
    assert len(man) == 711
    sizes = {n: s for n, s, _, _ in man}
    E = pkg.synth.ENC
    assert sizes[f"{E}.transformer.layers.0.attention.wq.weight"] == (2048, 1280)   # weights.rs:420-504
    assert sizes[pkg.synth.ADP + ".0.weight"] == (3072, 5120) and sizes["norm.weight"] == (3072,)
    q4_bytes = sum(int(np.prod(s)) // 32 * 18 for _, s, k, _ in man if k == "q4")
    assert abs(q4_bytes - 2.488e9) < 0.01e9                                   # ~2.5 GB Q4 GGUF
    t = pkg.synth.tiny_dims()
    p = str(tmp_path / "tiny.gguf"); pkg.synth.write_synthetic_gguf(p, t, seed=1)
    m = orc.Model(p); c = m.cfg
    assert (c.enc_layers, c.enc_dim, c.enc_heads, c.enc_head_dim, c.enc_ffn, c.enc_window) == (2, 128, 2, 64, 256, 750)
    assert (c.dec_layers, c.dec_dim, c.dec_heads, c.dec_kv_heads, c.dec_head_dim, c.dec_ffn, c.dec_window, c.vocab) == \
           (2, 256, 4, 2, 128, 512, 8192, 512)
    assert c.t_cond_dim == 32 and abs(c.rope_theta - 1e6) < 1 and abs(c.norm_eps - 1e-5) < 1e-9
    m.close()
