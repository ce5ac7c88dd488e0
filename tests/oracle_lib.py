"""ctypes binding of the CPU oracle (oracle/libvox_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libvox_oracle.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class PadCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("n_left_pad_tokens", C.c_uint32), ("frame_rate", C.c_float),
                ("extra_right_pad_tokens", C.c_uint32)]


class ChunkCfg(C.Structure):
    _fields_ = [("max_mel_frames", C.c_uint32), ("hop_length", C.c_uint32), ("sample_rate", C.c_uint32),
                ("overlap_frames", C.c_uint32)]


class Chunk(C.Structure):
    _fields_ = [("start_sample", C.c_size_t), ("end_sample", C.c_size_t), ("index", C.c_size_t), ("is_last", C.c_int)]


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("enc_layers", "enc_dim", "enc_heads", "enc_head_dim", "enc_ffn", "enc_window",
                                       "dec_layers", "dec_dim", "dec_heads", "dec_kv_heads", "dec_head_dim", "dec_ffn",
                                       "dec_window", "vocab", "n_mels", "reshape_factor", "t_cond_dim")] + \
               [("rope_theta", C.c_float), ("norm_eps", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "vox_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        build()
    # the oracle's OpenMP regions are tiny for the small parity models; a 256-thread team makes them slower, not faster
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 32)))
    L = C.CDLL(SO)
    vp = C.c_void_p
    sig = {
        "orc_peak_normalize": (None, [f32p, C.c_size_t, C.c_float]),
        "orc_pad_cfg_voxtral": (None, [C.POINTER(PadCfg)]),
        "orc_pad_samples_per_token": (C.c_size_t, [C.POINTER(PadCfg)]),
        "orc_pad_left_samples": (C.c_size_t, [C.POINTER(PadCfg)]),
        "orc_pad_right_samples": (C.c_size_t, [C.POINTER(PadCfg), C.c_size_t]),
        "orc_pad_len": (C.c_size_t, [C.POINTER(PadCfg), C.c_size_t]),
        "orc_pad_audio": (None, [C.POINTER(PadCfg), f32p, C.c_size_t, f32p]),
        "orc_needs_chunking": (C.c_int, [C.c_size_t, C.POINTER(ChunkCfg)]),
        "orc_chunk_plan": (C.c_size_t, [C.c_size_t, C.POINTER(ChunkCfg), C.POINTER(Chunk), C.c_size_t]),
        "orc_hann_window": (None, [C.c_int, f32p]),
        "orc_hz_to_mel": (C.c_float, [C.c_float]),
        "orc_mel_to_hz": (C.c_float, [C.c_float]),
        "orc_mel_filterbank": (None, [f32p]),
        "orc_mel_num_frames": (C.c_size_t, [C.c_size_t]),
        "orc_mel_compute": (None, [f32p, C.c_size_t, f32p]),
        "orc_mel_compute_log": (None, [f32p, C.c_size_t, f32p]),
        "orc_time_embedding": (None, [C.c_float, C.c_int, C.c_float, f32p]),
        "orc_q4_quantize": (None, [f32p, C.c_size_t, u8p]),
        "orc_q4_dequantize": (None, [u8p, C.c_size_t, f32p]),
        "orc_reference_matmul": (None, [f32p, f32p, C.c_int, C.c_int, C.c_int, f32p]),
        "orc_q4_matmul": (None, [u8p, C.c_int64, C.c_int64, f32p, C.c_int64, vp, f32p]),
        "orc_gguf_open": (vp, [C.c_char_p]),
        "orc_gguf_close": (None, [vp]),
        "orc_gguf_version": (C.c_uint32, [vp]),
        "orc_gguf_tensor_count": (C.c_uint64, [vp]),
        "orc_gguf_tensor_info": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_uint64 * 4), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
        "orc_gguf_tensor_data": (vp, [vp, C.c_char_p]),
        "orc_gguf_tensor_name": (C.c_char_p, [vp, C.c_uint64]),
        "orc_last_error": (C.c_char_p, []),
        "orc_rms_norm": (None, [f32p, C.c_int, C.c_int, f32p, C.c_float, f32p]),
        "orc_rope": (None, [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
        "orc_conv_out_len": (C.c_int, [C.c_int]),
        "orc_conv1d_gelu": (None, [f32p, C.c_int, C.c_int, f32p, f32p, C.c_int, f32p]),
        "orc_gelu": (C.c_float, [C.c_float]),
        "orc_silu": (C.c_float, [C.c_float]),
        "orc_attention": (None, [f32p, f32p, f32p] + [C.c_int] * 8 + [f32p]),
        "orc_model_load_gguf": (vp, [C.c_char_p]),
        "orc_model_free": (None, [vp]),
        "orc_model_config": (None, [vp, C.POINTER(ModelCfg)]),
        "orc_enc_seq_len": (C.c_int, [vp, C.c_int]),
        "orc_encoder_conv": (None, [vp, f32p, C.c_int, f32p]),
        "orc_encoder_layer": (None, [vp, C.c_int, f32p, C.c_int]),
        "orc_encoder_final_norm": (None, [vp, f32p, C.c_int]),
        "orc_encode_audio": (C.c_int, [vp, f32p, C.c_int, f32p]),
        "orc_resample_len": (C.c_size_t, [C.c_size_t, C.c_uint32, C.c_uint32]),
        "orc_resample": (None, [f32p, C.c_size_t, C.c_uint32, C.c_uint32, f32p]),
        "orc_resample_plan": (None, [C.c_uint32, C.c_uint32, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_float), vp]),
        "orc_enc_cache_create": (vp, [vp, C.c_int]),
        "orc_enc_cache_free": (None, [vp]),
        "orc_enc_cache_len": (C.c_int, [vp]),
        "orc_enc_cache_abs": (C.c_int, [vp]),
        "orc_enc_cache_apply_sliding_window": (None, [vp, C.c_int]),
        "orc_encode_audio_with_cache": (C.c_int, [vp, f32p, C.c_int, vp, f32p]),
        "orc_embed_tokens": (None, [vp, i32p, C.c_int, f32p]),
        "orc_cache_create": (vp, [vp, C.c_int]),
        "orc_cache_free": (None, [vp]),
        "orc_cache_len": (C.c_int, [vp]),
        "orc_cache_reset": (None, [vp]),
        "orc_cache_update": (None, [vp, C.c_int, C.c_int, f32p, f32p, C.c_int]),
        "orc_forward_hidden_with_cache": (None, [vp, f32p, C.c_int, f32p, vp, f32p]),
        "orc_lm_head": (None, [vp, f32p, C.c_int, f32p]),
        "orc_transcribe_streaming": (C.c_int, [vp, f32p, C.c_int, f32p, i32p, C.c_int, vp]),
        "orc_last_timings": (None, [C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "orc_num_threads": (C.c_int, []),
        "orc_f32_to_f16": (C.c_uint16, [C.c_float]),
        "orc_f16_to_f32": (C.c_float, [C.c_uint16]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def err() -> str:
    return (lib().orc_last_error() or b"").decode()


# ------------------------------------------------------------------ convenience wrappers

def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pad_audio(x, cfg=None):
    L = lib()
    if cfg is None:
        cfg = PadCfg(); L.orc_pad_cfg_voxtral(C.byref(cfg))
    x = f32(x)
    out = np.zeros(L.orc_pad_len(C.byref(cfg), x.size), dtype=np.float32)
    L.orc_pad_audio(C.byref(cfg), x, x.size, out)
    return out


def mel_compute_log(x):
    L = lib(); x = f32(x)
    T = L.orc_mel_num_frames(x.size)
    out = np.zeros((T, 128), dtype=np.float32)
    L.orc_mel_compute_log(x, x.size, out)
    return out


def mel_compute(x):
    L = lib(); x = f32(x)
    T = L.orc_mel_num_frames(x.size)
    out = np.zeros((T, 128), dtype=np.float32)
    L.orc_mel_compute(x, x.size, out)
    return out


def resample(x, sr_in, sr_out=16000):
    L = lib(); x = f32(x)
    out = np.zeros(L.orc_resample_len(x.size, sr_in, sr_out), dtype=np.float32)
    L.orc_resample(x, x.size, sr_in, sr_out, out)
    return out


def resample_plan(sr_in, sr_out=16000):
    """(fft_in, fft_out, output_delay, cutoff, taps[fft_in]) of the rubato `Fft` restatement"""
    L = lib(); a, b, d, c = C.c_long(), C.c_long(), C.c_long(), C.c_float()
    L.orc_resample_plan(sr_in, sr_out, C.byref(a), C.byref(b), C.byref(d), C.byref(c), None)
    taps = np.zeros(a.value, dtype=np.float32)
    L.orc_resample_plan(sr_in, sr_out, C.byref(a), C.byref(b), C.byref(d), C.byref(c), taps.ctypes.data_as(C.c_void_p))
    return a.value, b.value, d.value, c.value, taps


def time_embedding(t, dim=3072, theta=10000.0):
    out = np.zeros(dim, dtype=np.float32)
    lib().orc_time_embedding(t, dim, theta, out)
    return out


def q4_quantize(w):
    w = f32(w).reshape(-1)
    out = np.zeros(w.size // 32 * 18, dtype=np.uint8)
    lib().orc_q4_quantize(w, w.size, out)
    return out


def q4_dequantize(raw, n):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_q4_dequantize(np.ascontiguousarray(raw, dtype=np.uint8), n, out)
    return out


def reference_matmul(a, bt):
    a = f32(a); bt = f32(bt)
    m, k = a.shape; n = bt.shape[0]
    out = np.zeros((m, n), dtype=np.float32)
    lib().orc_reference_matmul(a, bt, m, k, n, out)
    return out


def q4_matmul(raw, N, K, x, bias=None):
    x = f32(x); bm = x.size // K
    out = np.zeros(x.shape[:-1] + (N,), dtype=np.float32)
    b = None if bias is None else f32(bias)
    lib().orc_q4_matmul(np.ascontiguousarray(raw, dtype=np.uint8), N, K, x, bm,
                        None if b is None else b.ctypes.data_as(C.c_void_p), out)
    return out


class Model:
    def __init__(self, path):
        self.h = lib().orc_model_load_gguf(path.encode())
        if not self.h:
            raise RuntimeError("oracle model load failed: " + err())
        self.cfg = ModelCfg(); lib().orc_model_config(self.h, C.byref(self.cfg))

    def close(self):
        if self.h:
            lib().orc_model_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enc_seq_len(self, T):
        return lib().orc_enc_seq_len(self.h, T)

    def encoder_conv(self, mel):
        mel = f32(mel); T = mel.shape[1]; S = self.enc_seq_len(T)
        out = np.zeros((S, self.cfg.enc_dim), dtype=np.float32)
        lib().orc_encoder_conv(self.h, mel, T, out)
        return out

    def encoder_layer(self, li, x):
        x = f32(x).copy()
        lib().orc_encoder_layer(self.h, li, x, x.shape[0])
        return x

    def encode_audio(self, mel):
        mel = f32(mel); T = mel.shape[1]
        S4 = self.enc_seq_len(T) // self.cfg.reshape_factor
        out = np.zeros((max(S4, 1), self.cfg.dec_dim), dtype=np.float32)
        n = lib().orc_encode_audio(self.h, mel, T, out)
        return out[:n]

    def enc_cache(self, cap):
        return lib().orc_enc_cache_create(self.h, cap)

    def encode_audio_with_cache(self, mel, cache):
        mel = f32(mel); T = mel.shape[1]
        S4 = self.enc_seq_len(T) // self.cfg.reshape_factor
        out = np.zeros((max(S4, 1), self.cfg.dec_dim), dtype=np.float32)
        n = lib().orc_encode_audio_with_cache(self.h, mel, T, cache, out)
        if n < 0:
            raise RuntimeError("oracle: chunk does not fit the encoder cache")
        return out[:n]

    def embed_tokens(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.zeros((ids.size, self.cfg.dec_dim), dtype=np.float32)
        lib().orc_embed_tokens(self.h, ids, ids.size, out)
        return out

    def cache(self, max_seq):
        return lib().orc_cache_create(self.h, max_seq)

    def cache_free(self, c):
        lib().orc_cache_free(c)

    def cache_update(self, c, layer, pos, k, v):
        k = f32(k); v = f32(v)
        lib().orc_cache_update(c, layer, pos, k, v, k.shape[1])

    def forward_hidden_with_cache(self, x, t_embed, cache):
        x = f32(x); M = x.shape[0]
        out = np.zeros_like(x)
        lib().orc_forward_hidden_with_cache(self.h, x, M, f32(t_embed), cache, out)
        return out

    def lm_head(self, h):
        h = f32(h); M = h.shape[0]
        out = np.zeros((M, self.cfg.vocab), dtype=np.float32)
        lib().orc_lm_head(self.h, h, M, out)
        return out

    def transcribe_streaming(self, mel, t_embed, want_logits=False):
        mel = f32(mel); T = mel.shape[1]
        S = self.enc_seq_len(T) // self.cfg.reshape_factor
        cap = max(S - 38, 1)
        ids = np.zeros(cap, dtype=np.int32)
        lg = np.zeros((cap, self.cfg.vocab), dtype=np.float32) if want_logits else None
        n = lib().orc_transcribe_streaming(self.h, mel, T, f32(t_embed), ids, cap,
                                           None if lg is None else lg.ctypes.data_as(C.c_void_p))
        return (ids[:n], lg[:n]) if want_logits else ids[:n]

    def timings(self):
        e = C.c_double(); d = C.c_double()
        lib().orc_last_timings(C.byref(e), C.byref(d))
        return e.value, d.value
