"""Pin the CPU oracle against golden vectors produced by the reference's own PyTorch restatement
(scripts/test_proper_inference.py, imported by tests/golden/make_golden.py in the build container).
CPU only.  Tolerances: the reference's own fixture tolerance for these ops is 1e-3 max-abs
(rms_norm.rs:156-291, rope.rs:167-253); torch sums in a different order, so we use 2e-4 relative."""
import numpy as np

from model_fixtures import golden, golden_gguf, rel_err


def test_rms_norm_rope_time_embedding(orc):
    g = golden(); L = orc.lib()
    x, w = g["in_norm_x"], g["in_norm_w"]
    out = np.zeros_like(x); L.orc_rms_norm(x, x.shape[0], x.shape[1], w, 1e-5, out)
    assert np.abs(out - g["out_rms_norm"]).max() < 1e-5
    r = g["in_rope_x"][0].copy()                       # [seq 7][heads 3][hd 64]
    L.orc_rope(r, 7, 3, 64, 0, 1e6)
    assert np.abs(r - g["out_rope"][0]).max() < 1e-5
    assert np.abs(orc.time_embedding(6.0, 256) - g["out_time_embedding_6"]).max() < 1e-5
    assert np.abs(orc.time_embedding(6.0, 3072) - g["out_time_embedding_full"]).max() < 1e-5


def test_encoder_matches_reference_python(orc):
    g = golden(); path, dims = golden_gguf()
    m = orc.Model(path)
    assert (m.cfg.enc_layers, m.cfg.enc_dim, m.cfg.enc_heads, m.cfg.dec_layers, m.cfg.dec_heads, m.cfg.dec_kv_heads) == (32, 1280, 32, 26, 32, 8)
    out = m.encode_audio(g["in_mel"])
    ref = g["out_encoder_out"]
    assert out.shape == ref.shape == (10, 256)
    assert rel_err(out, ref) < 2e-4, rel_err(out, ref)
    m.close()


def test_decoder_matches_reference_python(orc):
    g = golden(); path, _ = golden_gguf()
    m = orc.Model(path)
    x = g["in_dec_x"]; t = orc.time_embedding(6.0, 256)
    c = m.cache(16)
    hid = m.forward_hidden_with_cache(x, t, c)           # 12 positions in one causal pass
    assert rel_err(hid, g["out_decoder_hidden"]) < 2e-4, rel_err(hid, g["out_decoder_hidden"])
    lg = m.lm_head(hid)
    assert rel_err(lg, g["out_decoder_logits"]) < 2e-4
    assert (lg.argmax(1) == g["out_decoder_logits"].argmax(1)).all()
    m.cache_free(c)
    # KV-cache path (8-token prefill + 4 single steps) == full pass (attention.rs:429-474 semantics, model level)
    c = m.cache(16)
    h1 = m.forward_hidden_with_cache(x[:8], t, c)
    hs = [m.forward_hidden_with_cache(x[i:i + 1], t, c) for i in range(8, 12)]
    assert orc.lib().orc_cache_len(c) == 12
    assert np.abs(np.concatenate([h1] + hs) - hid).max() < 1e-5
    m.cache_free(c); m.close()


def test_log_mel_matches_reference_python(orc):
    """Pins the oracle's log-mel VALUES to the reference's own PyTorch front-end (scripts/test_proper_inference.py:62-98 compute_mel:
    torch.stft with the periodic Hann window, power spectrum without the last frame, Slaney bank, log10 / max(-6.5) / (x+4)/4), run on
    an already-padded clip by tests/golden/make_golden.py.  The reference's Rust test pins its mel to the same kind of fixture at 1e-2
    (mel.rs:608-613); f32 STFT vs the oracle's f64-accumulated DFT agree far tighter."""
    g = golden()
    ref = g["out_log_mel"]                                  # [128][T]
    out = orc.mel_compute_log(g["in_mel_audio"])            # [T][128]
    assert out.shape == ref.T.shape == (g["in_mel_audio"].size // 160, 128)
    d = np.abs(out - ref.T)
    assert d.max() < 5e-4 and (d > 1e-4).mean() < 1e-3, d.max()     # worst points: low-power bins next to the tones, where torch's f32 FFT rounds (2.3e-4)
    assert (ref > -0.6).mean() > 0.2                        # the fixture is not all floor values
