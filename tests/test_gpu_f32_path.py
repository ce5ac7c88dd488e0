"""GPU parity of the f32 (SafeTensors) path -- SURVEY.md section 8 row a32 / BASELINE configs[0-1]: VoxtralModelLoader ->
encode_audio / transcribe (transcribe_f32_with_model, bin/transcribe.rs:362-438) / decoder pieces, against the CPU oracle
running the same dense weights.  Weights are BF16 on disk like the published checkpoint (exact on both sides);
tolerance: max|d| <= 2e-4 * max|ref| on hidden states and logits, greedy ids identical up to the first near-tie."""
import numpy as np
import pytest

from model_fixtures import check_batch_rows, check_greedy_ids, fake_mel, rel_err, tiny_f32_pair

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def pair(pkg, orc):
    ctx = pkg.Context(0)
    st, gg, dims = tiny_f32_pair()
    m = pkg.VoxtralModelLoader.from_file(st).load(ctx)
    o = orc.Model(gg)
    yield m, o, ctx
    m.close(); o.close(); ctx.close()


def test_f32_loader_and_embed(pkg, pair, tmp_path):
    m, o, ctx = pair
    c = m.config
    assert (c.enc_layers, c.enc_dim, c.enc_heads, c.dec_layers, c.dec_dim, c.dec_heads, c.dec_kv_heads, c.vocab) == (2, 128, 2, 2, 256, 4, 2, 512)
    ids = np.array([1, 32, 0, 511, 77], dtype=np.int32)
    assert (m.decoder().embed_tokens_from_ids(ids, 1, 5)[0] == o.embed_tokens(ids)).all()          # exact: bf16 -> f32 row copy
    S = pkg.synth
    with pytest.raises(pkg.VoxError):
        pkg.VoxtralModelLoader.from_file(str(tmp_path / "missing.safetensors")).load(ctx)
    junk = tmp_path / "junk.safetensors"; junk.write_bytes(b"\x10\0\0\0\0\0\0\0not json at all!!")
    with pytest.raises(pkg.VoxError):
        pkg.VoxtralModelLoader.from_file(str(junk)).load(ctx)


@pytest.mark.parametrize("T", [64, 1144])
def test_f32_encode_audio(pair, T):
    m, o, _ = pair
    mel = fake_mel(T, seed=3 + T)
    ref = o.encode_audio(mel); out = m.encode_audio(mel[None])
    assert out.shape == (1,) + ref.shape and rel_err(out[0], ref) < TOL, rel_err(out[0], ref)


def test_f32_decoder_pieces(pkg, pair):
    m, o, _ = pair
    rng = np.random.default_rng(5); x = (0.5 * rng.standard_normal((12, 256))).astype(np.float32)
    t = pkg.TimeEmbedding(256).embed(6.0)
    oc = o.cache(16); ref = np.concatenate([o.forward_hidden_with_cache(x[:8], t, oc)] + [o.forward_hidden_with_cache(x[i:i + 1], t, oc) for i in range(8, 12)])
    dec = m.decoder(); c = dec.create_cache_preallocated(16)
    out = np.concatenate([dec.forward_hidden_with_cache(x[None, :8], t, c)[0]] + [dec.forward_hidden_with_cache(x[None, i:i + 1], t, c)[0] for i in range(8, 12)])
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    assert rel_err(dec.lm_head(out[None])[0], o.lm_head(ref)) < TOL
    o.cache_free(oc)


def test_f32_transcribe(pkg, pair):
    m, o, _ = pair
    mel = fake_mel(1144, seed=17); t = pkg.TimeEmbedding(256).embed(6.0)
    rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
    ids, lg = m.transcribe_streaming(mel[None], t, return_logits=True)
    assert len(ids) == len(rids) == 33
    scale = max(1.0, np.abs(rlg).max())
    assert np.abs(lg - rlg).max() <= TOL * scale
    check_greedy_ids(ids, rids, rlg, TOL)                                  # greedy ids identical to the CPU reference
    ids_g = m.transcribe_streaming(mel[None], t)                            # graph-replayed decode
    assert (ids_g == ids).all() and (m.transcribe_streaming(mel[None], t) == ids).all()


def test_f32_transcribe_batch(pkg, pair):
    """The batch API on the dense path (no tile-ordered weights -> the f32-activation step): rows == one-by-one transcription."""
    m, _, ctx = pair
    t = pkg.TimeEmbedding(256).embed(6.0)
    clips = [pkg.synth.synth_audio(sec, seed=90 + i) for i, sec in enumerate((2.0, 2.6, 2.0))]
    outs = m.transcribe_batch(clips, t)
    check_batch_rows(pkg, ctx, m, clips, t, outs, TOL)                          # every row: single-stream ids up to its first near-tie
    assert all((a == b).all() for a, b in zip(outs, m.transcribe_batch(clips, t)))
    # wider than 64 on a checkpoint the XF step does not cover (dense weights): vox_transcribe_batch serves it as lock-step batches of <= 64 in its sorted order
    many = [pkg.synth.synth_audio(0.6 + 0.11 * ((5 * i) % 13), seed=400 + i) for i in range(70)]
    wide = m.transcribe_batch(many, t)
    assert len(wide) == 70 and m.timings()["decode_tokens"] == sum(len(o) for o in wide)
    check_batch_rows(pkg, ctx, m, [many[i] for i in (0, 13, 69)], t, [wide[i] for i in (0, 13, 69)], TOL)


@pytest.mark.parametrize("dtype", ["F32", "F16"])
def test_f32_path_arbitrary_float_checkpoint(pkg, orc, tmp_path, dtype):
    """A dense checkpoint whose values are NOT bf16-representable (models/weights.rs:16-66 load_tensor accepts any F32 / F16 / BF16): the loader keeps
    the exact f32 values (WFMT_F32: f32 plane for the decode GEMV and the embedding, bf16 hi + lo planes for the MFMA GEMMs).  Parity vs the CPU
    oracle running the same values, same tolerances as the BF16 checkpoint."""
    S = pkg.synth; d = S.tiny_dims()
    rng = np.random.default_rng(77)
    tensors, gg = [], []
    for name, shape, kind, sigma in S.tensor_manifest(d):
        ne = int(np.prod(shape))
        v = ((1.0 + sigma * rng.standard_normal(ne)) if kind == "norm" else (sigma * rng.standard_normal(ne))).astype(np.float32)
        if dtype == "F16":
            v = v.astype(np.float16).astype(np.float32)            # f16-exact values: exact in the oracle's F16 GGUF and in the SafeTensors file
        tensors.append((name, shape, dtype, v.astype(np.float16) if dtype == "F16" else v))
        gg.append((name, shape, S.GGML_F32, v))
    assert any((np.asarray(t[3], np.float32).view(np.uint32) & 0xFFFF).any() for t in tensors)          # really not bf16-representable
    st = str(tmp_path / "m.safetensors"); S.write_safetensors(st, tensors)
    gp = str(tmp_path / "m.gguf"); S.write_gguf(gp, gg)
    ctx = pkg.Context(0)
    m = pkg.VoxtralModelLoader.from_file(st).load(ctx); o = orc.Model(gp)
    ids = np.array([1, 32, 0, 511, 77], dtype=np.int32)
    assert (m.decoder().embed_tokens_from_ids(ids, 1, 5)[0] == o.embed_tokens(ids)).all()          # exact f32 rows
    mel = fake_mel(1144, seed=21)
    assert rel_err(m.encode_audio(mel[None])[0], o.encode_audio(mel)) < TOL
    t = pkg.TimeEmbedding(256).embed(6.0)
    rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
    idsg, lg = m.transcribe_streaming(mel[None], t, return_logits=True)
    assert np.abs(lg - rlg).max() <= TOL * max(1.0, np.abs(rlg).max())
    check_greedy_ids(idsg, rids, rlg, TOL)
    assert (m.transcribe_streaming(mel[None], t) == idsg).all()                                        # graph replay
    clips = [pkg.synth.synth_audio(sec, seed=5 + i) for i, sec in enumerate((2.0, 2.4))]
    check_batch_rows(pkg, ctx, m, clips, t, m.transcribe_batch(clips, t), TOL)
    m.close(); o.close(); ctx.close()
