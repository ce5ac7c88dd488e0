"""Host-side mirror of the reference's `src/audio` module over the C ABI (same names, argument meaning
and error behaviour): PadConfig/pad_audio (audio/pad.rs), ChunkConfig/chunk_audio/needs_chunking
(audio/chunk.rs), peak_normalize (audio/io.rs:59-68), MelSpectrogram (audio/mel.rs), TimeEmbedding
(models/time_embedding.rs)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import check, lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class PadConfig:
    """audio/pad.rs:20-46"""
    sample_rate: int = 16000
    n_left_pad_tokens: int = 76
    frame_rate: float = 12.5
    extra_right_pad_tokens: int = 17

    @classmethod
    def voxtral(cls):
        c = _lib.PadCfg(); check(lib().vox_pad_cfg_voxtral(C.byref(c)))
        return cls(c.sample_rate, c.n_left_pad_tokens, c.frame_rate, c.extra_right_pad_tokens)

    def _c(self):
        return _lib.PadCfg(self.sample_rate, self.n_left_pad_tokens, self.frame_rate, self.extra_right_pad_tokens)

    def samples_per_token(self):
        return int(np.float32(self.sample_rate) / np.float32(self.frame_rate))

    def left_pad_samples(self):
        return self.n_left_pad_tokens * self.samples_per_token()

    def padded_len(self, n):
        out = C.c_size_t(); c = self._c(); check(lib().vox_pad_len(n, C.byref(c), C.byref(out)))
        return out.value


def pad_audio(samples, config: PadConfig | None = None):
    """audio/pad.rs:89-103"""
    config = config or PadConfig.voxtral()
    x = _f32(samples); c = config._c()
    out = np.empty(config.padded_len(x.size), dtype=np.float32)
    check(lib().vox_pad_audio(_ptr(x), x.size, C.byref(c), _ptr(out)))
    return out


def num_audio_tokens(n, config: PadConfig | None = None):
    config = config or PadConfig.voxtral()
    out = C.c_size_t(); c = config._c(); check(lib().vox_num_audio_tokens(n, C.byref(c), C.byref(out)))
    return out.value


@dataclass
class ChunkConfig:
    """audio/chunk.rs:9-68"""
    max_mel_frames: int = 1500
    hop_length: int = 160
    sample_rate: int = 16000
    overlap_frames: int = 0

    @classmethod
    def voxtral(cls):
        return cls()

    def with_max_frames(self, n):
        return ChunkConfig(n, self.hop_length, self.sample_rate, self.overlap_frames)

    def with_overlap(self, n):
        return ChunkConfig(self.max_mel_frames, self.hop_length, self.sample_rate, n)

    def max_samples_per_chunk(self):
        return self.max_mel_frames * self.hop_length

    def _c(self):
        return _lib.ChunkCfg(self.max_mel_frames, self.hop_length, self.sample_rate, self.overlap_frames)


@dataclass
class AudioChunk:
    samples: np.ndarray
    start_sample: int
    end_sample: int
    index: int
    is_last: bool


def needs_chunking(num_samples, config: ChunkConfig):
    out = C.c_int32(); c = config._c(); check(lib().vox_needs_chunking(num_samples, C.byref(c), C.byref(out)))
    return bool(out.value)


def chunk_plan(num_samples, config: ChunkConfig):
    """The chunk boundaries alone (audio/chunk.rs:125-157): [(start_sample, end_sample)] -- what the batched driver needs to make every chunk a unit of work
    without copying samples (cli.py)."""
    c = config._c(); n = C.c_size_t()
    check(lib().vox_chunk_plan(int(num_samples), C.byref(c), None, 0, C.byref(n)))
    arr = (_lib.Chunk * max(n.value, 1))()
    check(lib().vox_chunk_plan(int(num_samples), C.byref(c), arr, n.value, C.byref(n)))
    return [(int(arr[i].start_sample), int(arr[i].end_sample)) for i in range(n.value)]


def chunk_audio(samples, config: ChunkConfig):
    """audio/chunk.rs:159-161"""
    x = _f32(samples); c = config._c(); n = C.c_size_t()
    check(lib().vox_chunk_plan(x.size, C.byref(c), None, 0, C.byref(n)))
    arr = (_lib.Chunk * max(n.value, 1))()
    check(lib().vox_chunk_plan(x.size, C.byref(c), arr, n.value, C.byref(n)))
    return [AudioChunk(x[arr[i].start_sample:arr[i].end_sample].copy(), arr[i].start_sample, arr[i].end_sample,
                       arr[i].index, bool(arr[i].is_last)) for i in range(n.value)]


def resample_len(num_samples, sample_rate, target_rate=16000):
    """length of `resample`'s output (audio/resample.rs:33-34: ceil(n * ratio))"""
    n = C.c_size_t(); check(lib().vox_resample_len(int(num_samples), int(sample_rate), int(target_rate), C.byref(n)))
    return int(num_samples) if sample_rate == target_rate else n.value


def resample(ctx, samples, sample_rate, target_rate=16000):
    """audio/resample.rs:16-52 `resample` (rubato's synchronous FFT resampler as a block matrix product on the GPU; same rate -> copy)"""
    x = _f32(samples); n = C.c_size_t()
    check(lib().vox_resample_len(x.size, int(sample_rate), int(target_rate), C.byref(n)))
    out = np.empty(x.size if sample_rate == target_rate else n.value, dtype=np.float32)
    check(lib().vox_resample(ctx.h, _ptr(x), x.size, int(sample_rate), int(target_rate), _ptr(out), out.size, C.byref(n), 0))
    return out[:n.value]


def resample_to_16k(ctx, samples, sample_rate):
    """audio/resample.rs:10-13"""
    return resample(ctx, samples, sample_rate, 16000)


def resample_plan(sample_rate, target_rate=16000):
    """(fft_in, fft_out, output_delay, cutoff, taps[fft_in]) rubato's `Fft` resampler derives from the two rates (host)"""
    a_, b_, d_, c_ = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
    check(lib().vox_resample_plan(int(sample_rate), int(target_rate), C.byref(a_), C.byref(b_), C.byref(d_), C.byref(c_), None, 0))
    h = np.empty(a_.value, dtype=np.float32)
    check(lib().vox_resample_plan(int(sample_rate), int(target_rate), C.byref(a_), C.byref(b_), C.byref(d_), C.byref(c_), _ptr(h), h.size))
    return a_.value, b_.value, d_.value, c_.value, h


def peak_normalize(samples, target_peak=0.95):
    """AudioBuffer::peak_normalize, audio/io.rs:59-68 (returns a new array)"""
    x = _f32(samples).copy()
    check(lib().vox_peak_normalize(_ptr(x), x.size, target_peak))
    return x


class MelSpectrogram:
    """audio/mel.rs:63-350 with MelConfig::voxtral(); compute_log runs on the GPU of `ctx`."""

    def __init__(self, ctx):
        self.ctx = ctx

    @classmethod
    def voxtral(cls, ctx):
        return cls(ctx)

    def num_frames(self, num_samples):
        out = C.c_size_t(); check(lib().vox_mel_num_frames(num_samples, C.byref(out)))
        return out.value

    @staticmethod
    def mel_filterbank():
        fb = np.empty((128, 201), dtype=np.float32); check(lib().vox_mel_filterbank(_ptr(fb)))
        return fb

    @staticmethod
    def hann_window(length):
        w = np.empty(length, dtype=np.float32); check(lib().vox_hann_window(length, _ptr(w)))
        return w

    def compute_log(self, samples):
        """-> [n_frames, 128] float32 (audio/mel.rs:128-165)"""
        x = _f32(samples)
        out = np.empty((self.num_frames(x.size), 128), dtype=np.float32)
        if out.size:
            check(lib().vox_mel_compute_log(self.ctx.h, _ptr(x), x.size, _ptr(out), 0))
        return out

    def compute_log_flat(self, samples):
        return self.compute_log(samples).reshape(-1)


class TimeEmbedding:
    """models/time_embedding.rs:21-71"""

    def __init__(self, dim=3072):
        self.dim = dim

    def embed(self, t):
        out = np.empty(self.dim, dtype=np.float32)
        check(lib().vox_time_embedding(float(t), self.dim, _ptr(out)))
        return out
