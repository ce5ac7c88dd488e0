"""Streaming encoder: vox_encode_audio_with_cache against the CPU oracle's restatement of Q4AudioEncoder::forward_with_cache /
encode_audio_with_cache (gguf/model.rs:437-452, 299-317, 125-174, 791-799) and KVCache::apply_sliding_window (kv_cache.rs:176-203).
Tolerance 2e-4 * max|ref| like every other hidden-state comparison; eviction must not change results at all (keys older than the window are
masked anyway)."""
import numpy as np
import pytest

from model_fixtures import fake_mel, rel_err, tiny_gguf

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def tiny(pkg, orc):
    ctx = pkg.Context(0)
    path, dims = tiny_gguf()
    m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    o = orc.Model(path)
    yield m, o
    m.close(); o.close(); ctx.close()


def test_single_chunk_equals_encode_audio(tiny):
    m, o = tiny
    mel = fake_mel(1144, seed=2)
    c = m.create_encoder_cache()
    a = m.encode_audio_with_cache(mel[None], c); b = m.encode_audio(mel[None])
    assert a.shape == b.shape and rel_err(a[0], b[0]) < 1e-6
    assert c.seq_len() == o.enc_seq_len(1144) == c.abs_pos()


def test_chunked_matches_oracle_chunked(orc, tiny):
    m, o = tiny
    mel = fake_mel(1600, seed=5)
    cuts = [0, 320, 328, 900, 1600]                       # ragged chunks, one tiny (8 frames -> 2 rows -> 0 adapter rows)
    c = m.create_encoder_cache(); oc = o.enc_cache(2048)
    tot = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        ref = o.encode_audio_with_cache(mel[:, a:b], oc)
        out = m.encode_audio_with_cache(mel[None, :, a:b], c)[0]
        assert out.shape == ref.shape
        if ref.size:
            assert rel_err(out, ref) < TOL, (a, b, rel_err(out, ref))
        tot += o.enc_seq_len(b - a)
        assert c.seq_len() == tot == orc.lib().orc_enc_cache_len(oc)
    orc.lib().orc_enc_cache_free(oc)


def test_eviction_is_exact_and_matches_oracle(pkg, orc, tiny):
    """1000 encoder positions streamed in 50-row chunks through an 800-row cache (window 750: evicts as it goes) == the same stream through a cache
    that never evicts, and == the oracle (cat-mode cache + apply_sliding_window).  Also beyond the 4096-row load-time RoPE table."""
    m, o = tiny
    assert m.config.enc_window == 750
    mel = fake_mel(200 * 20, seed=9)
    small = m.create_encoder_cache(800); big = m.create_encoder_cache(1100); oc = o.enc_cache(1100)
    for i in range(20):
        ch = mel[:, 200 * i:200 * (i + 1)]
        a = m.encode_audio_with_cache(ch[None], small)[0]; b = m.encode_audio_with_cache(ch[None], big)[0]
        ref = o.encode_audio_with_cache(ch, oc)
        assert a.shape == b.shape == ref.shape == (12, 256)                 # 200 frames -> 50 rows -> 12 adapter rows (2 rows dropped, adapter.rs:114)
        assert rel_err(a, b) < 5e-5, (i, rel_err(a, b))                     # eviction changes nothing but the key-tile alignment of the flash softmax (f32 re-association, measured 7e-6)
        assert rel_err(a, ref) < TOL, (i, rel_err(a, ref))
    assert small.abs_pos() == big.abs_pos() == 1000 == big.seq_len() and small.seq_len() <= 800
    orc.lib().orc_enc_cache_apply_sliding_window(oc, 750); assert orc.lib().orc_enc_cache_len(oc) == 750      # kv_cache.rs:176-203
    big.apply_sliding_window(750); assert big.seq_len() == 750 and big.abs_pos() == 1000
    ch = fake_mel(200, seed=77)
    a = m.encode_audio_with_cache(ch[None], small)[0]; b = m.encode_audio_with_cache(ch[None], big)[0]; ref = o.encode_audio_with_cache(ch, oc)
    assert rel_err(a, b) < 5e-5 and rel_err(b, ref) < TOL
    orc.lib().orc_enc_cache_free(oc)
    # long stream: positions beyond the 4096-row RoPE table of the loader (gguf/loader.rs:196-198) use the streaming table
    c = m.create_encoder_cache(1024); oc = o.enc_cache(4400)
    rng = np.random.default_rng(1)
    for i in range(22):
        ch = (0.6 * rng.standard_normal((128, 800)) + 0.3).astype(np.float32)   # 200 rows per chunk -> 4400 positions
        a = m.encode_audio_with_cache(ch[None], c)[0]; ref = o.encode_audio_with_cache(ch, oc)
        if i >= 19:
            assert rel_err(a, ref) < TOL, (i, rel_err(a, ref))
    assert c.abs_pos() == 4400
    orc.lib().orc_enc_cache_free(oc)


def test_streaming_encoder_errors(pkg, tiny):
    m, _ = tiny
    with pytest.raises(pkg.VoxError, match="exceed"):
        m.create_encoder_cache(700)                                           # capacity must exceed the window
    c = m.create_encoder_cache(800)
    with pytest.raises(pkg.VoxError, match="does not fit"):
        m.encode_audio_with_cache(fake_mel(800)[None], c)                     # 200 rows > 800 - 750
    d = m.decoder().create_cache_preallocated(64)
    with pytest.raises(pkg.VoxError, match="encoder cache"):
        m.encode_audio_with_cache(fake_mel(160)[None], d)
    with pytest.raises(pkg.VoxError):
        m.decoder().forward_hidden_with_cache(np.zeros((1, 1, 256), np.float32), pkg.TimeEmbedding(256).embed(6.0), c)
    c.reset(); assert c.seq_len() == 0 and c.abs_pos() == 0
