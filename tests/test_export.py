"""SafeTensors -> Q4_0 GGUF exporter: tensor selection per gguf/loader.rs, block bytes per gguf/tests.rs:24-57, and the
exported file loads through the C-ABI reader and the CPU oracle."""
import numpy as np
import pytest


def test_export_tiny_checkpoint(pkg, orc, tmp_path):
    S = pkg.synth; E = pkg.export
    d = S.tiny_dims()
    st = str(tmp_path / "c.safetensors"); S.write_synthetic_safetensors(st, d, seed=3)
    out = str(tmp_path / "q4.gguf")
    stats = E.export_q4_gguf(st, out)
    man = S.tensor_manifest(d)
    assert stats["q4"] == sum(1 for _, _, k, _ in man if k == "q4") and stats["q4"] + stats["f32"] == len(man)
    src = E.read_safetensors(st); got = S.read_gguf_tensors(out)
    assert set(got) == set(src) == {n for n, _, _, _ in man}
    for name, shape, kind, _ in man:
        gshape, gdt, raw = got[name]
        assert tuple(gshape) == tuple(shape)
        v = E.to_f32(src[name][2], src[name][1])
        if kind == "q4":
            assert gdt == S.GGML_Q4_0 and (np.asarray(raw) == S.quantize_q4_0(v)).all()            # gguf/tests.rs:24-57 bytes
        else:
            assert gdt == S.GGML_F32 and (np.asarray(raw).view(np.float32) == v).all()
    # the reference's reader (C ABI) and the oracle's loader accept it
    r = pkg.GgufReader.open(out)
    assert r.tensor_count() == len(man) and r.version() == 3
    r.close()
    m = orc.Model(out)
    assert (m.cfg.enc_layers, m.cfg.dec_layers, m.cfg.dec_dim, m.cfg.vocab) == (d.enc_layers, d.dec_layers, d.dec_dim, d.vocab)
    # dequantised weights stay within the Q4_0 error bound of the source (|err| <= d/2 + f16 rounding, d = amax / 7)
    w = E.to_f32(src["layers.0.attention.wq.weight"][2], "BF16"); dq = S.dequantize_q4_0(np.asarray(got["layers.0.attention.wq.weight"][2]), w.size)
    amax = np.abs(w.reshape(-1, 32)).max(axis=1)
    assert (np.abs(dq - w).reshape(-1, 32).max(axis=1) <= amax / 7 * 0.5001 + amax * 2e-3 + 1e-9).all() or np.abs(dq - w).max() < 0.08
    m.close()


def test_ggml_scheme_and_errors(pkg, tmp_path):
    E = pkg.export; S = pkg.synth
    rng = np.random.default_rng(0); w = rng.standard_normal(32 * 50).astype(np.float32)
    raw = E.quantize_q4_0_ggml(w).reshape(-1, 18)
    d = raw[:, :2].copy().view(np.float16).astype(np.float32)[:, 0]
    mx = w.reshape(-1, 32)[np.arange(50), np.abs(w.reshape(-1, 32)).argmax(1)]
    assert np.allclose(d, (mx / -8).astype(np.float16).astype(np.float32))
    dq = S.dequantize_q4_0(raw.reshape(-1), w.size)
    assert np.abs(dq - w).max() <= np.abs(d).max() * 0.5 + 0.02 and (raw[:, 2:] & 0xF).max() <= 15
    assert np.abs(dq - w).mean() < np.abs(S.dequantize_q4_0(S.quantize_q4_0(w), w.size) - w).mean() * 1.2      # same class of error as the reference quantiser
    assert (E.quantize_q4_0_ggml(np.zeros(64, np.float32)).reshape(-1, 18)[:, 2:] == 0x88).all()                # zero block: q = 8
    bad = tmp_path / "x.safetensors"; bad.write_bytes(b"\x05\0\0\0\0\0\0\0{}{}{")
    with pytest.raises(Exception):
        E.read_safetensors(str(bad))
    assert E.is_q4_tensor("layers.0.attention.wq.weight", (512, 256)) and not E.is_q4_tensor("layers.0.attention_norm.weight", (256,))
    assert not E.is_q4_tensor("norm.weight", (256,)) and not E.is_q4_tensor("x.conv_layers.0.conv.weight", (128, 128, 3)) and not E.is_q4_tensor("a.bias", (64,))


def test_heavy_tail_dense_generator_is_exact_in_bf16_and_f16():
    """The stress checkpoint of the f32 path (synth.dense_checkpoint_tensors(heavy_tail=True), tests/golden/make_fullsize_f32_heavytail_golden.py): block scales are
    powers of two, so every value is a bf16 number AND survives the oracle's f16 copy exactly; outlier rows are 64 x larger; the plain generator is unchanged."""
    import numpy as np
    from __graft_entry__ import load_package
    S = load_package().synth
    dims = S.tiny_dims()
    plain = {n: b for n, _, _, b in S.dense_checkpoint_tensors(dims, 8)}
    seen = 0
    for name, shape, kind, bits in S.dense_checkpoint_tensors(dims, 8, heavy_tail=True):
        f = S.bf16_bits_to_f32(bits)
        assert np.array_equal(S.f32_to_bf16_bits(f), bits)                                   # bf16-exact by construction
        if kind == "q4" and len(shape) == 2 and int(shape[1]) % 32 == 0:
            f16 = S._bf16_bits_to_f16_bits(bits).view(np.float16).astype(np.float32)
            assert np.array_equal(f16, f)                                                    # ... and exact in the oracle's f16 copy
            ratio = np.abs(f) / np.abs(S.bf16_bits_to_f32(plain[name]))
            lg = np.log2(ratio); assert np.array_equal(lg, np.rint(lg)) and lg.min() >= -2 and lg.max() <= 10      # power-of-two block scales
            seen += 1
        elif kind != "norm":
            assert np.array_equal(bits, plain[name])
    assert seen > 4
