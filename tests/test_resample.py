"""Sample-rate conversion (audio/resample.rs:16-52).  PARITY UNPINNED versus the reference's rubato FFT resampler (third-party, not in the tree):
what is pinned is (a) the reference's own contract tests (same rate = identity, length within 100 samples, resample.rs:56-83), (b) the CPU oracle's
f64 restatement of the replacement's spec against analytic signals and scipy's polyphase resampler, (c) the GPU kernel against that oracle."""
import numpy as np
import pytest


def test_oracle_resample_contract_and_quality(orc):
    x = np.full(48000, 0.5, np.float32)
    assert (orc.resample(x, 16000, 16000) == np.full(48000, 0.5, np.float32)[:48000]).all() and orc.resample(x[:16000], 16000).size == 16000   # resample.rs:56-63
    y = orc.resample(x, 48000)                                                   # resample.rs:66-83
    assert abs(y.size - 16000) < 100 and np.abs(y[200:-200] - 0.5).max() < 1e-6
    for sr in (44100, 22050, 8000, 32000, 96000, 11025):
        n = sr * 2; t = np.arange(n) / sr
        x = (0.4 * np.sin(2 * np.pi * 1000 * t) + 0.3 * np.sin(2 * np.pi * 3100 * t + 0.5)).astype(np.float32) if sr > 8000 else (0.4 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
        y = orc.resample(x, sr)
        assert abs(y.size - 2 * 16000) <= 1
        t2 = np.arange(y.size) / 16000.0
        ref = 0.4 * np.sin(2 * np.pi * 1000 * t2) + (0.3 * np.sin(2 * np.pi * 3100 * t2 + 0.5) if sr > 8000 else 0)
        m = slice(400, -400)
        assert np.abs(y[m] - ref[m]).max() < 2e-4, (sr, np.abs(y[m] - ref[m]).max())         # in-band tones are reproduced
    # stop band: a 10 kHz tone at 48 kHz is above the 16 kHz output's Nyquist -> removed
    t = np.arange(48000) / 48000.0
    y = orc.resample(np.sin(2 * np.pi * 10000 * t).astype(np.float32), 48000)
    assert np.abs(y[300:-300]).max() < 1e-4
    # against scipy's polyphase resampler (different filter): band-limited noise agrees to a few 1e-3
    from scipy.signal import resample_poly, butter, sosfiltfilt
    rng = np.random.default_rng(0)
    z = sosfiltfilt(butter(8, 5000, fs=44100, output="sos"), rng.standard_normal(44100)).astype(np.float32)
    a = orc.resample(z, 44100); b = resample_poly(z.astype(np.float64), 160, 441)
    k = min(a.size, b.size)
    assert np.abs(a[300:k - 300] - b[300:k - 300]).max() < 5e-3 * np.abs(b).max()


def test_filter_table_matches_oracle_design(pkg, orc):
    """vox_resample_filter (host, no GPU needed): unit DC gain per phase, symmetric prototype, and the table reproduces the oracle's output."""
    for sr in (48000, 44100, 24000):
        P, Q, W, h = pkg.resample_filter(sr, 16000)
        g = np.gcd(sr, 16000); assert (P, Q) == (sr // g, 16000 // g) and h.shape == (Q, 2 * W + 1)
        assert np.abs(h.sum(axis=1) - 1).max() < 1e-6
        assert np.abs(h[0] - h[0][::-1]).max() < 1e-7                            # phase 0 is symmetric
        rng = np.random.default_rng(sr); x = rng.standard_normal(3000).astype(np.float32)
        ref = orc.resample(x, sr)
        m = np.arange(ref.size); num = m * P; n0 = num // Q; ph = num % Q
        idx = n0[:, None] - W + np.arange(2 * W + 1)[None, :]
        xv = np.where((idx >= 0) & (idx < x.size), x[np.clip(idx, 0, x.size - 1)], 0.0)
        out = (xv.astype(np.float64) * h[ph].astype(np.float64)).sum(axis=1)
        assert np.abs(out - ref).max() < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in,sr_out", [(48000, 16000), (44100, 16000), (22050, 16000), (8000, 16000), (32000, 16000), (16000, 24000), (11025, 16000)])
def test_gpu_resample_vs_oracle(pkg, orc, sr_in, sr_out):
    ctx = pkg.Context(0)
    rng = np.random.default_rng(sr_in + sr_out)
    x = (0.3 * rng.standard_normal(sr_in * 3 + 17)).astype(np.float32)
    ref = orc.resample(x, sr_in, sr_out); out = pkg.resample(ctx, x, sr_in, sr_out)
    assert out.shape == ref.shape and abs(out.size - x.size * sr_out / sr_in) <= 1
    assert np.abs(out - ref).max() < 1e-5, np.abs(out - ref).max()
    same = pkg.resample(ctx, x, sr_in, sr_in); assert (same == x).all()          # resample.rs:17-19
    assert pkg.resample(ctx, np.zeros(0, np.float32), sr_in, sr_out).size == 0
    ctx.close()
