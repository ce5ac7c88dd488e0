"""ctypes loader for libvoxtral_hip.so (the C ABI in include/voxtral_hip.h).  Fails loudly if the
HIP library has not been built -- there is no fallback implementation."""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH


class VoxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[vox error {code}] {msg}")
        self.code = code


class PadCfg(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("n_left_pad_tokens", C.c_uint32), ("frame_rate", C.c_float),
                ("extra_right_pad_tokens", C.c_uint32)]


class ChunkCfg(C.Structure):
    _fields_ = [("max_mel_frames", C.c_uint32), ("hop_length", C.c_uint32), ("sample_rate", C.c_uint32),
                ("overlap_frames", C.c_uint32)]


class Chunk(C.Structure):
    _fields_ = [("start_sample", C.c_size_t), ("end_sample", C.c_size_t), ("index", C.c_size_t), ("is_last", C.c_int32)]


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("enc_layers", "enc_dim", "enc_heads", "enc_head_dim", "enc_ffn", "enc_window",
                                         "dec_layers", "dec_dim", "dec_heads", "dec_kv_heads", "dec_head_dim", "dec_ffn",
                                         "dec_window", "vocab", "n_mels", "reshape_factor", "t_cond_dim")] + \
               [("rope_theta", C.c_float), ("norm_eps", C.c_float)]


class Timings(C.Structure):
    _fields_ = [("preprocess_ms", C.c_double), ("encode_ms", C.c_double), ("decode_ms", C.c_double), ("total_ms", C.c_double),
                ("decode_tokens", C.c_int32), ("graph_replays", C.c_int32)]


vp, i32, i64, u32, u64, sz, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_size_t, C.c_float
P = C.POINTER

# every symbol include/voxtral_hip.h declares (tests check the .so exports all of them)
SIGNATURES = {
    "vox_last_error": (C.c_char_p, []),
    "vox_abi_version": (i32, []),
    "vox_device_count": (i32, [P(i32)]),
    "vox_ctx_create": (i32, [i32, P(vp)]),
    "vox_ctx_destroy": (i32, [vp]),
    "vox_debug_reload_knobs": (i32, []),
    "vox_ctx_synchronize": (i32, [vp]),
    "vox_ctx_set_shared": (i32, [vp, i32]),
    "vox_ctx_stream": (i32, [vp, P(vp)]),
    "vox_dev_alloc": (i32, [vp, sz, P(vp)]),
    "vox_dev_free": (i32, [vp, vp]),
    "vox_dev_upload": (i32, [vp, vp, vp, sz]),
    "vox_dev_download": (i32, [vp, vp, vp, sz]),
    "vox_dev_copy": (i32, [vp, vp, vp, sz]),
    "vox_peak_normalize": (i32, [vp, sz, f32]),
    "vox_pad_cfg_voxtral": (i32, [P(PadCfg)]),
    "vox_pad_len": (i32, [sz, P(PadCfg), P(sz)]),
    "vox_pad_audio": (i32, [vp, sz, P(PadCfg), vp]),
    "vox_num_audio_tokens": (i32, [sz, P(PadCfg), P(sz)]),
    "vox_needs_chunking": (i32, [sz, P(ChunkCfg), P(i32)]),
    "vox_chunk_plan": (i32, [sz, P(ChunkCfg), P(Chunk), sz, P(sz)]),
    "vox_mel_num_frames": (i32, [sz, P(sz)]),
    "vox_mel_filterbank": (i32, [vp]),
    "vox_hann_window": (i32, [i32, vp]),
    "vox_mel_compute_log": (i32, [vp, vp, sz, vp, i32]),
    "vox_time_embedding": (i32, [f32, i32, vp]),
    "vox_gguf_open": (i32, [C.c_char_p, P(vp)]),
    "vox_gguf_open_memory": (i32, [vp, C.c_size_t, P(vp)]),
    "vox_gguf_open_shards": (i32, [vp, vp, i32, P(vp)]),
    "vox_q4_model_load_gguf": (i32, [vp, vp, C.c_uint32, P(vp)]),
    "vox_gguf_close": (i32, [vp]),
    "vox_gguf_version": (i32, [vp, P(u32)]),
    "vox_gguf_tensor_count": (i32, [vp, P(u64)]),
    "vox_gguf_tensor_name": (i32, [vp, u64, P(C.c_char_p)]),
    "vox_gguf_tensor_info": (i32, [vp, C.c_char_p, P(u64 * 4), P(u32), P(u32), P(u64)]),
    "vox_gguf_tensor_data": (i32, [vp, C.c_char_p, vp, sz]),
    "vox_q4_tensor_from_bytes": (i32, [vp, vp, sz, i64, i64, P(vp)]),
    "vox_q4_tensor_shape": (i32, [vp, P(i64), P(i64)]),
    "vox_q4_tensor_num_blocks": (i32, [vp, P(i64)]),
    "vox_q4_tensor_dequantize": (i32, [vp, vp, vp]),
    "vox_q4_tensor_free": (i32, [vp]),
    "vox_q4_matmul": (i32, [vp, vp, vp, i32, i32, vp, i32]),
    "vox_q4_linear_forward": (i32, [vp, vp, vp, vp, i32, i32, vp, i32]),
    "vox_dense_tensor_from_f32": (i32, [vp, vp, vp, i64, i64, P(vp)]),
    "vox_linear_forward_ex": (i32, [vp, vp, vp, vp, i32, i32, vp, i32, i32]),
    "vox_conv_downsample": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp]),
    "vox_attention": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, i32]),
    "vox_q4_model_load": (i32, [vp, C.c_char_p, P(vp)]),
    "vox_q4_model_load_ex": (i32, [vp, C.c_char_p, u32, P(vp)]),
    "vox_f32_model_load": (i32, [vp, C.c_char_p, P(vp)]),
    "vox_model_free": (i32, [vp]),
    "vox_model_config": (i32, [vp, P(ModelCfg)]),
    "vox_model_weight_bytes": (i32, [vp, P(u64)]),
    "vox_model_arena": (i32, [vp, P(vp), P(u64)]),
    "vox_model_set_t_embed": (i32, [vp, vp]),
    "vox_model_set_decode_engine": (i32, [vp, i32, P(i32)]),
    "vox_debug_occupy": (i32, [vp, i32, i32]),
    "vox_model_memory": (i32, [vp, P(C.c_uint64)]),
    "vox_model_set_batch_engine": (i32, [vp, i32, P(i32), P(C.c_uint64)]),
    "vox_model_arena_finalize": (i32, [vp]),
    "vox_model_replicate": (i32, [vp, vp, P(vp)]),
    "vox_model_set_sessions": (i32, [vp, i32]),
    "vox_generate_step_with_cache": (i32, [vp, vp, i32, vp, vp, vp]),
    "vox_encode_audio": (i32, [vp, vp, i32, vp, i32, P(i32), i32]),
    "vox_transcribe_streaming": (i32, [vp, vp, i32, vp, vp, i32, P(i32), vp, i32]),
    "vox_transcribe_audio": (i32, [vp, vp, sz, vp, vp, i32, P(i32), i32]),
    "vox_transcribe_batch": (i32, [vp, i32, P(vp), P(sz), vp, P(vp), P(i32), P(i32), i32]),
    "vox_transcribe_batch_ex": (i32, [vp, i32, P(vp), P(sz), vp, vp, P(vp), P(i32), P(i32), i32]),
    "vox_decoder_cache_create": (i32, [vp, i32, P(vp)]),
    "vox_cache_free": (i32, [vp]),
    "vox_cache_seq_len": (i32, [vp, P(i32)]),
    "vox_cache_reset": (i32, [vp]),
    "vox_cache_update": (i32, [vp, i32, i32, vp, vp, i32, i32]),
    "vox_cache_truncate": (i32, [vp, i32]),
    "vox_resample_len": (i32, [sz, C.c_uint32, C.c_uint32, P(sz)]),
    "vox_resample": (i32, [vp, vp, sz, C.c_uint32, C.c_uint32, vp, sz, P(sz), i32]),
    "vox_resample_plan": (i32, [C.c_uint32, C.c_uint32, P(i32), P(i32), P(i32), P(C.c_float), vp, sz]),
    "vox_encoder_cache_create": (i32, [vp, i32, P(vp)]),
    "vox_encoder_cache_apply_sliding_window": (i32, [vp, i32]),
    "vox_cache_abs_pos": (i32, [vp, P(i32)]),
    "vox_encode_audio_with_cache": (i32, [vp, vp, i32, vp, vp, i32, P(i32), i32]),
    "vox_embed_tokens_from_ids": (i32, [vp, vp, i32, vp]),
    "vox_forward_hidden_with_cache": (i32, [vp, vp, i32, vp, vp, vp]),
    "vox_lm_head": (i32, [vp, vp, i32, vp]),
    "vox_embed_tokens_from_ids_ex": (i32, [vp, vp, i32, vp, i32]),
    "vox_tensor_add": (i32, [vp, vp, vp, sz, vp, i32]),
    "vox_forward_hidden_with_cache_ex": (i32, [vp, vp, i32, vp, vp, vp, P(vp), i32]),
    "vox_lm_head_ex": (i32, [vp, vp, i32, vp, i32]),
    "vox_argmax_rows": (i32, [vp, vp, i32, i32, vp, i32]),
    "vox_lm_head_argmax": (i32, [vp, vp, i32, vp, i32]),
    "vox_forward": (i32, [vp, vp, i32, vp, vp, i32, P(i32), i32]),
    "vox_forward_streaming": (i32, [vp, vp, i32, vp, i32, vp, vp, i32, P(i32), i32]),
    "vox_forward_with_cache": (i32, [vp, vp, i32, vp, vp, vp, vp, i32, P(i32), i32]),
    "vox_get_stage_timings": (i32, [vp, P(Timings)]),
    "vox_bench_decode_gemv": (i32, [vp, i32, i32, P(C.c_double), P(C.c_double), P(C.c_char_p)]),
    "vox_bench_wide": (i32, [vp, i32, i32, i32, P(C.c_double)]),
    "vox_debug_timeline_start": (i32, [vp, i32, i32]),
    "vox_debug_timeline_fetch": (i32, [vp, vp, sz, P(i32), vp]),
}

_LIB = None
_KNOBS = None


def _knob_env():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("VOX_")))


def lib():
    """dlopen libvoxtral_hip.so and bind every declared symbol.  Raises if the library is missing.
    The library snapshots its VOX_* measurement knobs at vox_ctx_create; this Python mirror (tests and tools flip knobs between two calls) asks it to
    re-read them whenever the VOX_* part of os.environ has changed since the last call."""
    global _LIB, _KNOBS
    if _LIB is not None:
        k = _knob_env()
        if k != _KNOBS:
            _KNOBS = k; _LIB.vox_debug_reload_knobs()
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise VoxError(-1, f"{LIB_PATH} not built: run `python __graft_entry__.py build` (hipcc, gfx950). "
                           "There is no CPU fallback for the HIP path.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError here == missing export == ABI drift; fail loudly
        fn.restype = res
        fn.argtypes = args
    _LIB = L; _KNOBS = _knob_env()
    return L


def check(code: int):
    if code != 0:
        raise VoxError(code, (lib().vox_last_error() or b"").decode(errors="replace"))
