"""Sample-rate conversion (audio/resample.rs:16-52).  The reference's resampler is rubato 1.0's synchronous FFT resampler (`Fft`, chunk 1024, 2 sub-chunks,
FixedSync::Input), a third-party crate that is not in the reference's tree: PARITY UNPINNED versus the crate's own output.  What is pinned:
(a) the reference's contract tests (same rate = identity, length within 100 samples, duration: resample.rs:56-108);
(b) the CPU oracle's restatement of rubato's published algorithm (direct DFT sums, f64) against an independent numpy restatement that runs real FFTs, analytic
    signals and scipy's polyphase resampler (a different filter);
(c) the product's plan + filter taps (host) and its GPU block-matrix kernels against that oracle."""
import math
import numpy as np
import pytest

RATES = [(48000, 16000), (44100, 16000), (22050, 16000), (8000, 16000), (32000, 16000), (16000, 24000), (11025, 16000), (96000, 16000), (24000, 16000)]


def rubato_fft_numpy(x, sr_in, sr_out):
    """rubato's `Fft` resampler, step by step with numpy's FFTs (f32 filter design as the crate does it, f64 transforms)."""
    g = math.gcd(sr_in, sr_out); mi, mo = sr_in // g, sr_out // g
    chunks = math.ceil(np.float32(1024) / np.float32(2) / np.float32(mi)); Ni, No = chunks * mi, chunks * mo
    cutoff = np.float32(0.4) ** (np.float32(16.0) / np.float32(Ni)) * (np.float32(No) / np.float32(Ni) if Ni > No else np.float32(1))
    k = np.arange(Ni, dtype=np.float32); npf = np.float32(Ni); pi = np.float32(np.pi)
    w = (np.float32(0.35875) - np.float32(0.48829) * np.cos(np.float32(2) * pi * k / npf) + np.float32(0.14128) * np.cos(np.float32(4) * pi * k / npf)
         - np.float32(0.01168) * np.cos(np.float32(6) * pi * k / npf)).astype(np.float32)
    v = ((k - np.float32(Ni // 2)) * cutoff).astype(np.float32); sinc = np.ones(Ni, np.float32); nz = v != 0
    sinc[nz] = np.sin(v[nz] * pi) / (v[nz] * pi)
    h = (w * w * sinc).astype(np.float32); h = (h / h.sum(dtype=np.float32) / np.float32(2 * Ni)).astype(np.float32)
    ft = np.zeros(2 * Ni); ft[:Ni] = h; H = np.fft.rfft(ft)
    new_len = Ni + 1 if Ni < No else No
    n_out = math.ceil((sr_out / sr_in) * len(x)); delay = No // 2
    nblk = (n_out + delay + No - 1) // No
    xp = np.zeros(nblk * Ni); xp[:min(len(x), nblk * Ni)] = x[:nblk * Ni]
    stream = np.zeros((nblk + 1) * No)
    for c in range(nblk):
        buf = np.zeros(2 * Ni); buf[:Ni] = xp[c * Ni:(c + 1) * Ni]
        X = np.fft.rfft(buf); Y = np.zeros(No + 1, complex); Y[:new_len] = X[:new_len] * H[:new_len]; Y[0] = Y[0].real
        stream[c * No:c * No + 2 * No] += np.fft.irfft(Y, 2 * No) * (2 * No)          # realfft's inverse is unnormalised
    return stream[delay:delay + n_out].astype(np.float32), (Ni, No, delay, float(cutoff), h)


def test_oracle_resample_contract(orc):
    """the reference's own tests, resample.rs:56-108"""
    x = np.full(16000, 0.5, np.float32)
    assert (orc.resample(x, 16000, 16000) == x).all()                              # test_resample_same_rate
    assert abs(orc.resample(np.full(48000, 0.5, np.float32), 48000).size - 16000) < 100      # test_resample_downsample
    assert abs(orc.resample(np.full(8000, 0.5, np.float32), 8000).size - 16000) < 100        # test_resample_upsample
    assert abs(orc.resample(np.full(24000, 0.5, np.float32), 24000).size / 16000.0 - 1.0) < 0.02   # test_resample_preserves_duration
    y = orc.resample(np.full(48000, 0.5, np.float32), 48000)
    assert np.abs(y[200:-200] - 0.5).max() < 1e-6                                  # unit DC gain
    assert orc.resample(np.zeros(0, np.float32), 44100).size == 0
    for n in (1, 5, 440, 441, 442, 1763, 1764, 1765):                              # n_out = ceil(n * (f64(out) / f64(in))), rubato's formula
        assert orc.resample(np.ones(n, np.float32), 44100).size == math.ceil((16000 / 44100) * n)


@pytest.mark.parametrize("sr_in,sr_out", RATES)
def test_oracle_resample_vs_numpy_fft_restatement(orc, sr_in, sr_out):
    rng = np.random.default_rng(sr_in)
    x = (0.3 * rng.standard_normal(sr_in + 17)).astype(np.float32)
    ref, (Ni, No, delay, cutoff, h) = rubato_fft_numpy(x, sr_in, sr_out)
    out = orc.resample(x, sr_in, sr_out)
    a, b, d, c, taps = orc.resample_plan(sr_in, sr_out)
    assert (a, b, d) == (Ni, No, delay) and abs(c - cutoff) < 1e-7 and np.abs(taps - h).max() < 1e-9
    assert out.shape == ref.shape and np.abs(out - ref).max() < 2e-6, np.abs(out - ref).max()


def test_oracle_resample_quality(orc):
    for sr in (48000, 44100, 22050, 8000, 32000, 11025, 24000):
        Ni, No, delay, _, _ = orc.resample_plan(sr)
        tau = (Ni // 2) * No / Ni - delay       # the filter is centred on the INTEGER half of fft_in: odd block sizes leave a constant sub-sample delay
        n = sr * 2; t = np.arange(n) / sr; two = sr > 8000
        x = (0.4 * np.sin(2 * np.pi * 1000 * t) + (0.3 * np.sin(2 * np.pi * 3100 * t + 0.5) if two else 0)).astype(np.float32)
        y = orc.resample(x, sr)
        assert abs(y.size - 2 * 16000) <= 1
        t2 = (np.arange(y.size) - tau) / 16000.0
        ref = 0.4 * np.sin(2 * np.pi * 1000 * t2) + (0.3 * np.sin(2 * np.pi * 3100 * t2 + 0.5) if two else 0)
        m = slice(2000, -2000)
        assert np.abs(y[m] - ref[m]).max() < 2e-4, (sr, np.abs(y[m] - ref[m]).max())         # in-band tones are reproduced
    # stop band: a 10 kHz tone at 48 kHz is above the 16 kHz output's Nyquist -> removed
    t = np.arange(48000) / 48000.0
    y = orc.resample(np.sin(2 * np.pi * 10000 * t).astype(np.float32), 48000)
    assert np.abs(y[300:-300]).max() < 1e-3
    # against scipy's polyphase resampler (a different filter): band-limited noise agrees to a few 1e-3
    from scipy.signal import resample_poly, butter, sosfiltfilt
    rng = np.random.default_rng(0)
    z = sosfiltfilt(butter(8, 5000, fs=44100, output="sos"), rng.standard_normal(44100)).astype(np.float32)
    a = orc.resample(z, 44100); b = resample_poly(z.astype(np.float64), 160, 441)
    k = min(a.size, b.size)
    assert np.abs(a[300:k - 300] - b[300:k - 300]).max() < 5e-3 * np.abs(b).max()


def test_resample_plan_matches_oracle(pkg, orc):
    """vox_resample_plan / vox_resample_len (host, no GPU needed): the plan, the cutoff and the filter taps equal the oracle's; unit DC gain."""
    for sr_in, sr_out in RATES + [(192000, 16000), (88200, 16000), (12000, 16000), (44101, 16000)]:
        fi, fo, d, c, h = pkg.resample_plan(sr_in, sr_out)
        a, b, dd, cc, taps = orc.resample_plan(sr_in, sr_out)
        assert (fi, fo, d) == (a, b, dd) and c == cc, (sr_in, sr_out)
        assert np.abs(h - taps).max() < 1e-9 and abs(h.sum(dtype=np.float64) * 2 * fi - 1) < 1e-5
    import ctypes as C
    L = pkg._lib.lib() if hasattr(pkg, "_lib") else None
    if L is not None:
        for n in (0, 1, 441, 442, 1764, 705600, 705601):
            v = C.c_size_t(); assert L.vox_resample_len(n, 44100, 16000, C.byref(v)) == 0 and v.value == math.ceil((16000 / 44100) * n)
            assert L.vox_resample_len(n, 16000, 16000, C.byref(v)) == 0 and v.value == n


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in,sr_out", RATES)
def test_gpu_resample_vs_oracle(pkg, orc, sr_in, sr_out):
    ctx = pkg.Context(0)
    rng = np.random.default_rng(sr_in + sr_out)
    x = (0.3 * rng.standard_normal(sr_in * 3 + 17)).astype(np.float32)
    ref = orc.resample(x, sr_in, sr_out); out = pkg.resample(ctx, x, sr_in, sr_out)
    assert out.shape == ref.shape and abs(out.size - x.size * sr_out / sr_in) <= 1
    assert np.abs(out - ref).max() < 1e-5, np.abs(out - ref).max()
    again = pkg.resample(ctx, x, sr_in, sr_out); assert (again == out).all()    # cached matrix, bit-identical
    short = pkg.resample(ctx, x[:100], sr_in, sr_out); assert np.abs(short - orc.resample(x[:100], sr_in, sr_out)).max() < 1e-5      # less than one block
    same = pkg.resample(ctx, x, sr_in, sr_in); assert (same == x).all()          # resample.rs:17-19
    assert pkg.resample(ctx, np.zeros(0, np.float32), sr_in, sr_out).size == 0
    ctx.close()


@pytest.mark.gpu
def test_gpu_resample_rate_switch_and_refusal(pkg, orc):
    """one context, alternating rate pairs (the block matrix is rebuilt); co-prime rates are refused loudly instead of degrading"""
    ctx = pkg.Context(0)
    rng = np.random.default_rng(7)
    for sr in (44100, 48000, 44100, 192000):
        x = (0.3 * rng.standard_normal(sr + 3)).astype(np.float32)
        assert np.abs(pkg.resample(ctx, x, sr, 16000) - orc.resample(x, sr, 16000)).max() < 1e-5
    with pytest.raises(Exception, match="not supported"):
        pkg.resample(ctx, np.zeros(1000, np.float32), 44101, 16000)
    ctx.close()
