"""Host-side mirror of the reference's `src/gguf` module over the C ABI: GgufReader (reader.rs),
Q4Tensor (tensor.rs), q4_matmul (op.rs), Q4Linear (linear.rs), Q4ModelLoader (loader.rs),
Q4VoxtralModel / Q4LanguageModel surface (model.rs).  All tensors cross as numpy float32 / int32."""
from __future__ import annotations

import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import check, lib, VoxError

F32, F16, Q4_0 = 0, 1, 2


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One HIP device + stream (the reference's implicit WgpuDevice::default())."""

    def __init__(self, device: int = 0):
        self.h = C.c_void_p()
        check(lib().vox_ctx_create(device, C.byref(self.h)))
        self.device = device

    def occupy(self, workgroups: int, micros: int):
        """Test hook: `workgroups` x 1024 threads spin for `micros` us on a side stream (returns at once)."""
        check(lib().vox_debug_occupy(self.h, workgroups, micros))

    def set_shared(self, shared: bool = True):
        """vox_ctx_set_shared: this context shares its GPU with other sessions (no batched decode engines, slot planner on the scaled cost table); results unchanged."""
        check(lib().vox_ctx_set_shared(self.h, 1 if shared else 0))

    def synchronize(self):
        check(lib().vox_ctx_synchronize(self.h))

    def stream(self):
        s = C.c_void_p(); check(lib().vox_ctx_stream(self.h, C.byref(s)))
        return s.value

    def alloc(self, nbytes):
        p = C.c_void_p(); check(lib().vox_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, p):
        check(lib().vox_dev_free(self.h, C.c_void_p(p)))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr); p = self.alloc(arr.nbytes)
        check(lib().vox_dev_upload(self.h, C.c_void_p(p), _ptr(arr), arr.nbytes))
        return p

    def download(self, p, shape, dtype=np.float32):
        out = np.empty(shape, dtype=dtype)
        check(lib().vox_dev_download(self.h, _ptr(out), C.c_void_p(p), out.nbytes))
        return out

    def copy(self, dst, src, nbytes):
        check(lib().vox_dev_copy(self.h, C.c_void_p(dst), C.c_void_p(src), nbytes))

    def close(self):
        if self.h:
            lib().vox_ctx_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int32(); check(lib().vox_device_count(C.byref(n)))
    return n.value


class GgufTensorInfo:
    def __init__(self, name, dims, dtype, nbytes):
        self.name, self._dims, self._dtype, self._nbytes = name, dims, dtype, nbytes

    def shape(self):
        return list(self._dims)

    def dtype(self):
        return self._dtype

    def num_elements(self):
        return int(np.prod(self._dims)) if self._dims else 1

    def byte_size(self):
        return self._nbytes


class GgufReader:
    """gguf/reader.rs:98-223 (v2/v3; dtypes F32/F16/Q4_0)"""

    def __init__(self, path=None):
        self.h = C.c_void_p(); self._keep = None
        if path is not None:
            check(lib().vox_gguf_open(str(path).encode(), C.byref(self.h)))

    @classmethod
    def open(cls, path):
        return cls(path)

    @classmethod
    def from_bytes(cls, data):
        """gguf/reader.rs:98-103: parse a GGUF image held in memory (kept alive by this object, not copied)."""
        r = cls(); r._keep = np.frombuffer(data, dtype=np.uint8)
        check(lib().vox_gguf_open_memory(r._keep.ctypes.data, r._keep.size, C.byref(r.h)))
        return r

    @classmethod
    def from_shards(cls, shards):
        """gguf/loader.rs:101-107: consecutive pieces (<= 512 MB each in the reference's WASM loader) of one GGUF image."""
        arrs = [np.frombuffer(s, dtype=np.uint8) for s in shards]; n = len(arrs)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs]); sizes = (C.c_size_t * n)(*[a.size for a in arrs])
        r = cls(); check(lib().vox_gguf_open_shards(ptrs, sizes, n, C.byref(r.h)))
        return r

    def version(self):
        v = C.c_uint32(); check(lib().vox_gguf_version(self.h, C.byref(v))); return v.value

    def tensor_count(self):
        v = C.c_uint64(); check(lib().vox_gguf_tensor_count(self.h, C.byref(v))); return v.value

    def tensor_names(self):
        out = []
        for i in range(self.tensor_count()):
            s = C.c_char_p(); check(lib().vox_gguf_tensor_name(self.h, i, C.byref(s))); out.append(s.value.decode())
        return out

    def tensor_info(self, name):
        dims = (C.c_uint64 * 4)(); nd = C.c_uint32(); dt = C.c_uint32(); nb = C.c_uint64()
        r = lib().vox_gguf_tensor_info(self.h, name.encode(), C.byref(dims), C.byref(nd), C.byref(dt), C.byref(nb))
        if r == 4:
            return None                      # Option::None, reader.rs:200-202
        check(r)
        return GgufTensorInfo(name, [int(dims[i]) for i in range(nd.value)], dt.value, nb.value)

    def tensor_data(self, name):
        info = self.tensor_info(name)
        if info is None:
            raise VoxError(4, f"Tensor '{name}' not found in GGUF")
        out = np.empty(info.byte_size(), dtype=np.uint8)
        check(lib().vox_gguf_tensor_data(self.h, name.encode(), _ptr(out), out.size))
        return out

    def close(self):
        if self.h:
            lib().vox_gguf_close(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Q4Tensor:
    """gguf/tensor.rs:21-113"""

    def __init__(self, ctx, h, shape):
        self.ctx, self.h, self._shape = ctx, h, shape

    @classmethod
    def from_q4_bytes(cls, raw_bytes, shape, ctx: Context):
        raw = np.ascontiguousarray(np.frombuffer(raw_bytes, dtype=np.uint8) if not isinstance(raw_bytes, np.ndarray) else raw_bytes, dtype=np.uint8)
        n, k = shape; h = C.c_void_p()
        check(lib().vox_q4_tensor_from_bytes(ctx.h, _ptr(raw), raw.size, n, k, C.byref(h)))
        return cls(ctx, h, [n, k])

    @classmethod
    def from_f32(cls, w, ctx: Context, other=None):
        """Dense f32-path weight [N, K] (models/weights.rs:16-66) in the f32 model's device format; `other`: a second [N, K] tensor interleaved row by row
        (the fused gate | up operand of SwiGLU, models/layers/swiglu.rs:72-77)."""
        w = _f32(w); n, k = w.shape; h = C.c_void_p()
        o = None if other is None else _f32(other)
        check(lib().vox_dense_tensor_from_f32(ctx.h, _ptr(w), None if o is None else _ptr(o), n, k, C.byref(h)))
        return cls(ctx, h, [n if o is None else 2 * n, k])

    def shape(self):
        return list(self._shape)

    def num_blocks(self):
        v = C.c_int64(); check(lib().vox_q4_tensor_num_blocks(self.h, C.byref(v))); return v.value

    def dequantize(self):
        out = np.empty(self._shape, dtype=np.float32)
        check(lib().vox_q4_tensor_dequantize(self.ctx.h, self.h, _ptr(out)))
        return out

    def close(self):
        if self.h:
            lib().vox_q4_tensor_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def q4_matmul(x, weights: Q4Tensor):
    """gguf/op.rs:86-137: x [B, M, K] float32 -> [B, M, N]; raises on rank/shape mismatch (the reference panics)."""
    x = _f32(x)
    if x.ndim != 3:
        raise VoxError(1, f"q4_matmul expects a 3-D input, got {x.ndim}-D")
    b, m, k = x.shape
    n, kw = weights.shape()
    if k != kw:
        raise VoxError(1, f"q4_matmul: input K={k} != weight K={kw}")
    out = np.empty((b, m, n), dtype=np.float32)
    check(lib().vox_q4_matmul(weights.ctx.h, weights.h, _ptr(x), b, m, _ptr(out), 0))
    return out


def linear_forward(weights: Q4Tensor, x, bias=None, epilogue=0):
    """Linear::forward with a fused epilogue (0 none, 1 GELU, 2 SwiGLU over interleaved gate / up rows -> N / 2 columns); x [B, M, K]."""
    x = _f32(x); b, m, k = x.shape; n = weights.shape()[0]
    out = np.empty((b, m, n // 2 if epilogue == 2 else n), dtype=np.float32)
    bb = None if bias is None else _f32(bias)
    check(lib().vox_linear_forward_ex(weights.ctx.h, weights.h, None if bb is None else _ptr(bb), _ptr(x), b, m, _ptr(out), epilogue, 0))
    return out


def conv_downsample(ctx, x, w1, b1, w2, b2):
    """ConvDownsampler::forward (models/layers/conv.rs:78-83): x [C, L] -> [O, L2] through the conv stem's im2col MFMA path."""
    x, w1, b1, w2, b2 = (_f32(a) for a in (x, w1, b1, w2, b2))
    c_, l_ = x.shape; o_ = w1.shape[0]; l2 = ((l_ + 1) // 2 + 1) // 2
    out = np.empty((o_, l2), dtype=np.float32)
    check(lib().vox_conv_downsample(ctx.h, _ptr(x), c_, l_, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), o_, _ptr(out)))
    return out


def attention(ctx, q, k, v, n_heads, n_kv_heads, offset=0, window=-1):
    """Attention core (gguf/model.rs:100-120,125-198 + masking.rs:9-107): q [M, n_heads*hd], k/v [kv_len, n_kv_heads*hd]
    -> [M, n_heads*hd]; query m sits at position offset+m, causal, optional sliding window."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    if q.ndim != 2 or k.ndim != 2 or k.shape != v.shape or q.shape[1] % n_heads:
        raise VoxError(1, "attention: bad shapes")
    hd = q.shape[1] // n_heads
    if k.shape[1] != n_kv_heads * hd:
        raise VoxError(1, "attention: k/v width != n_kv_heads*head_dim")
    out = np.empty_like(q)
    check(lib().vox_attention(ctx.h, _ptr(q), _ptr(k), _ptr(v), q.shape[0], k.shape[0], n_heads, n_kv_heads, hd, offset, window, _ptr(out), 0))
    return out


class Q4Linear:
    """gguf/linear.rs:17-40"""

    def __init__(self, weights: Q4Tensor, bias=None):
        self.weights = weights
        self.bias = None if bias is None else _f32(bias)

    @classmethod
    def new(cls, weights, bias=None):
        return cls(weights, bias)

    def forward(self, x):
        x = _f32(x); b, m, k = x.shape; n = self.weights.shape()[0]
        out = np.empty((b, m, n), dtype=np.float32)
        check(lib().vox_q4_linear_forward(self.weights.ctx.h, self.weights.h, None if self.bias is None else _ptr(self.bias),
                                          _ptr(x), b, m, _ptr(out), 0))
        return out


class LayerCaches:
    """create_cache_preallocated, gguf/model.rs:711-723 / kv_cache.rs:221-258"""

    def __init__(self, model, max_seq):
        self.model = model; self.h = C.c_void_p()
        check(lib().vox_decoder_cache_create(model.h, max_seq, C.byref(self.h)))
        model._caches.add(self)

    def seq_len(self):
        v = C.c_int32(); check(lib().vox_cache_seq_len(self.h, C.byref(v))); return v.value

    def reset(self):
        check(lib().vox_cache_reset(self.h))

    def update(self, layer, pos, k, v):
        """KVCache::update on one layer (kv_cache.rs:116-136): k / v [kv_heads][n][head_dim] host arrays -> rows pos .. pos + n"""
        k = _f32(k); v = _f32(v); assert k.shape == v.shape and k.ndim == 3
        check(lib().vox_cache_update(self.h, layer, pos, _ptr(k), _ptr(v), k.shape[1], 0))

    def truncate(self, n):
        check(lib().vox_cache_truncate(self.h, n))

    def close(self):
        if self.h:
            lib().vox_cache_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class EncoderCaches(LayerCaches):
    """Q4AudioEncoder::create_cache (gguf/model.rs:454-459): K / V of the streaming encoder; evicts rows older than the sliding window by itself."""

    def __init__(self, model, capacity_rows=0):
        self.model = model; self.h = C.c_void_p()
        check(lib().vox_encoder_cache_create(model.h, capacity_rows, C.byref(self.h)))
        model._caches.add(self)

    def abs_pos(self):
        v = C.c_int32(); check(lib().vox_cache_abs_pos(self.h, C.byref(v))); return v.value

    def apply_sliding_window(self, window):
        """kv_cache.rs:176-203 (every layer)"""
        check(lib().vox_encoder_cache_apply_sliding_window(self.h, window))


class Q4LanguageModel:
    """The decoder surface used by e2e-bench (gguf/model.rs:566-723)."""

    def __init__(self, model):
        self._m = model

    def n_layers(self):
        return self._m.config.dec_layers

    def d_model(self):
        return self._m.config.dec_dim

    def embed_tokens_from_ids(self, ids, batch=1, seq=None):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        out = np.empty((ids.size, self.d_model()), dtype=np.float32)
        check(lib().vox_embed_tokens_from_ids(self._m.h, _ptr(ids), ids.size, _ptr(out)))
        return out.reshape(batch, -1, self.d_model())

    def create_cache_preallocated(self, max_seq):
        return LayerCaches(self._m, max_seq)

    def forward_hidden_with_cache(self, x, t_embed, caches: LayerCaches):
        x = _f32(x); shp = x.shape; x2 = x.reshape(-1, self.d_model())
        out = np.empty_like(x2)
        check(lib().vox_forward_hidden_with_cache(self._m.h, _ptr(x2), x2.shape[0], _ptr(_f32(t_embed).reshape(-1)), caches.h, _ptr(out)))
        return out.reshape(shp)

    def lm_head(self, hidden):
        h = _f32(hidden); shp = h.shape; h2 = h.reshape(-1, self.d_model())
        out = np.empty((h2.shape[0], self._m.config.vocab), dtype=np.float32)
        check(lib().vox_lm_head(self._m.h, _ptr(h2), h2.shape[0], _ptr(out)))
        return out.reshape(shp[:-1] + (self._m.config.vocab,))


    # ---- device-resident forms (VOX_MEM_DEVICE): raw device pointers in, nothing copied, nothing synchronised -- the loop of bin/e2e_bench.rs:179-224 on "tensors"
    def embed_tokens_from_ids_dev(self, ids, out_ptr):
        ids = np.ascontiguousarray(ids, dtype=np.int32).reshape(-1)
        check(lib().vox_embed_tokens_from_ids_ex(self._m.h, _ptr(ids), ids.size, C.c_void_p(out_ptr), 1))

    def forward_hidden_with_cache_dev(self, x_ptr, rows, t_embed, caches: LayerCaches, out_ptr=None):
        """-> device pointer of the model-owned hidden rows (read-only; valid until the next decoder call).  One row on an engine-eligible cache = one engine launch."""
        ws = C.c_void_p()
        check(lib().vox_forward_hidden_with_cache_ex(self._m.h, C.c_void_p(x_ptr), rows, _ptr(_f32(t_embed).reshape(-1)), caches.h,
                                                     None if out_ptr is None else C.c_void_p(out_ptr), C.byref(ws), 1))
        return ws.value

    def lm_head_dev(self, hidden_ptr, rows, logits_ptr):
        check(lib().vox_lm_head_ex(self._m.h, C.c_void_p(hidden_ptr), rows, C.c_void_p(logits_ptr), 1))

    def lm_head_argmax(self, hidden_ptr, rows):
        """lm_head + argmax(2) + read-back: `rows` token ids (host)."""
        ids = np.zeros(rows, dtype=np.int32)
        check(lib().vox_lm_head_argmax(self._m.h, C.c_void_p(hidden_ptr), rows, _ptr(ids), 1))
        return ids


def tensor_add_dev(ctx, a_ptr, b_ptr, n, out_ptr):
    """out = a + b on device pointers (n floats), on the context's stream"""
    check(lib().vox_tensor_add(ctx.h, C.c_void_p(a_ptr), C.c_void_p(b_ptr), n, C.c_void_p(out_ptr), 1))


def argmax_rows_dev(ctx, logits_ptr, rows, vocab):
    """`logits.argmax(2)` + scalar read-back of device logits [rows][vocab] -> host ids; synchronises the stream"""
    ids = np.zeros(rows, dtype=np.int32)
    check(lib().vox_argmax_rows(ctx.h, C.c_void_p(logits_ptr), rows, vocab, _ptr(ids), 1))
    return ids


class Q4VoxtralModel:
    """gguf/model.rs:759-989"""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h
        self._caches = weakref.WeakSet()
        self.config = _lib.ModelCfg(); check(lib().vox_model_config(h, C.byref(self.config)))

    def decoder(self):
        return Q4LanguageModel(self)

    def create_decoder_cache_preallocated(self, max_seq):
        return LayerCaches(self, max_seq)

    def generate_step_with_cache(self, token_ids, t_embed, caches):
        """gguf/model.rs:857-867: logits [n][vocab] of the text tokens `token_ids` against the decoder cache (which advances by n)."""
        ids = np.ascontiguousarray(token_ids, dtype=np.int32).reshape(-1)
        out = np.empty((ids.size, self.config.vocab), dtype=np.float32)
        check(lib().vox_generate_step_with_cache(self.h, _ptr(ids), ids.size, _ptr(_f32(t_embed).reshape(-1)), caches.h, _ptr(out)))
        return out

    def _mel2(self, mel):
        mel = _f32(mel); return mel.reshape(mel.shape[-2], mel.shape[-1])

    def forward(self, mel, t_embed):
        """gguf/model.rs:820-830: mel -> logits [1, S, vocab], the audio embeddings alone as decoder input."""
        mel = self._mel2(mel); T = mel.shape[1]; cap = T // 16 + 2
        out = np.empty((cap, self.config.vocab), dtype=np.float32); S = C.c_int32()
        check(lib().vox_forward(self.h, _ptr(mel), T, _ptr(_f32(t_embed).reshape(-1)), _ptr(out), cap, C.byref(S), 0))
        return out[:S.value][None].copy()

    def forward_streaming(self, mel, token_ids, t_embed):
        """gguf/model.rs:802-816: mel + one token id per audio position -> logits [1, S, vocab]."""
        mel = self._mel2(mel); T = mel.shape[1]; cap = T // 16 + 2
        ids = np.ascontiguousarray(token_ids, dtype=np.int32).reshape(-1)
        out = np.empty((cap, self.config.vocab), dtype=np.float32); S = C.c_int32()
        check(lib().vox_forward_streaming(self.h, _ptr(mel), T, _ptr(ids), ids.size, _ptr(_f32(t_embed).reshape(-1)), _ptr(out), cap, C.byref(S), 0))
        return out[:S.value][None].copy()

    def forward_with_cache(self, mel, t_embed, encoder_cache, decoder_cache):
        """gguf/model.rs:833-843: one chunk through the streaming encoder and the cached decoder -> logits [1, S_chunk, vocab]."""
        mel = self._mel2(mel); T = mel.shape[1]; cap = T // 16 + 2
        out = np.empty((cap, self.config.vocab), dtype=np.float32); S = C.c_int32()
        check(lib().vox_forward_with_cache(self.h, _ptr(mel), T, _ptr(_f32(t_embed).reshape(-1)), encoder_cache.h, decoder_cache.h, _ptr(out), cap, C.byref(S), 0))
        return out[:S.value][None].copy()

    def set_decode_engine(self, on: bool) -> bool:
        """Persistent decode-step engine (one launch per token) on / off; returns whether it is active (it needs the real decoder geometry on a 256-CU device)."""
        a = C.c_int32(); check(lib().vox_model_set_decode_engine(self.h, 1 if on else 0, C.byref(a))); return bool(a.value)

    def memory(self):
        """Device bytes: weight arena, its primary (broadcast) part, the decode engines' weight stream, the engines' edge buffers."""
        v = (C.c_uint64 * 4)(); check(lib().vox_model_memory(self.h, v)); return {"arena": v[0], "arena_primary": v[1], "engine_stream": v[2], "engine_state": v[3]}

    def set_batch_engine(self, on=None):
        """Batched decode-layer engine (one launch per 16-row group and step) on / off (None: query); returns (active, engine launches enqueued so far)."""
        a = C.c_int32(); n = C.c_uint64()
        check(lib().vox_model_set_batch_engine(self.h, -1 if on is None else (1 if on else 0), C.byref(a), C.byref(n))); return bool(a.value), int(n.value)

    def weight_bytes(self):
        v = C.c_uint64(); check(lib().vox_model_weight_bytes(self.h, C.byref(v))); return v.value

    def arena(self):
        p = C.c_void_p(); n = C.c_uint64(); check(lib().vox_model_arena(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def replicate(self, dst_ctx):
        """vox_model_replicate: one more replica of this Q4 model on `dst_ctx` (another GPU or the same one) -- layout from the tensor manifest, one device-to-device
        copy of the primary arena, derived copies rebuilt there; no file, no collective library."""
        h = C.c_void_p(); check(lib().vox_model_replicate(self.h, dst_ctx.h, C.byref(h)))
        return type(self)(dst_ctx, h)

    def set_sessions(self, sessions: int):
        """vox_model_set_sessions: batch calls with >= 128 units per session run as `sessions` concurrent sessions on this model's GPU (hidden contexts + replicas +
        library threads); 1 = off (replicas freed).  Same ids per unit."""
        check(lib().vox_model_set_sessions(self.h, int(sessions)))

    def arena_finalize(self):
        """Receiver side of the multi-GPU start-up: the bytes of arena() have been written (e.g. by an RCCL broadcast); rebuild the derived copies on this GPU."""
        check(lib().vox_model_arena_finalize(self.h))

    def encode_audio(self, mel):
        """mel [1,128,T] or [128,T] -> [1,S,dec_dim] (gguf/model.rs:783-788)"""
        mel = _f32(mel); mel = mel.reshape(mel.shape[-2], mel.shape[-1]); T = mel.shape[1]
        cap = T // 16 + 2
        out = np.empty((cap, self.config.dec_dim), dtype=np.float32); S = C.c_int32()
        check(lib().vox_encode_audio(self.h, _ptr(mel), T, _ptr(out), cap, C.byref(S), 0))
        return out[:S.value].reshape(1, S.value, self.config.dec_dim).copy()

    def create_encoder_cache(self, capacity_rows=0):
        return EncoderCaches(self, capacity_rows)

    def encode_audio_with_cache(self, mel, encoder_cache):
        """mel chunk [1,128,T] or [128,T] -> [1, floor(S_chunk/4), dec_dim] (gguf/model.rs:791-799); K / V appended to `encoder_cache`"""
        mel = _f32(mel); mel = mel.reshape(mel.shape[-2], mel.shape[-1]); T = mel.shape[1]
        cap = T // 16 + 2
        out = np.empty((cap, self.config.dec_dim), dtype=np.float32); S = C.c_int32()
        check(lib().vox_encode_audio_with_cache(self.h, _ptr(mel), T, encoder_cache.h, _ptr(out), cap, C.byref(S), 0))
        return out[:S.value].reshape(1, S.value, self.config.dec_dim).copy()

    def transcribe_streaming(self, mel, t_embed, return_logits=False):
        """-> ids of length max(S-38, 1) for S >= 38 decoder positions, empty below (gguf/model.rs:873-963)"""
        mel = _f32(mel); mel = mel.reshape(mel.shape[-2], mel.shape[-1]); T = mel.shape[1]
        cap = T // 16 + 2
        ids = np.zeros(cap, dtype=np.int32); n = C.c_int32()
        t = _f32(t_embed).reshape(-1)
        lg = np.empty((cap, self.config.vocab), dtype=np.float32) if return_logits else None
        check(lib().vox_transcribe_streaming(self.h, _ptr(mel), T, _ptr(t), _ptr(ids), cap, C.byref(n),
                                             None if lg is None else _ptr(lg), 0))
        if return_logits:
            return ids[:n.value].copy(), lg[:n.value].copy()
        return ids[:n.value].copy()

    def transcribe_audio(self, samples, t_embed, device_ptr=None, n_samples=None):
        """Whole path from 16 kHz samples (peak-normalise, pad, mel, encode, decode)."""
        t = _f32(t_embed).reshape(-1)
        if device_ptr is None:
            x = _f32(samples); n_samples = x.size; ptr = _ptr(x); kind = 0
        else:
            ptr = C.c_void_p(device_ptr); kind = 1
        cap = n_samples // 1280 + 128
        ids = np.zeros(cap, dtype=np.int32); n = C.c_int32()
        check(lib().vox_transcribe_audio(self.h, ptr, n_samples, _ptr(t), _ptr(ids), cap, C.byref(n), kind))
        return ids[:n.value].copy()

    def transcribe_batch(self, samples_list, t_embed, device_ptrs=None, n_samples=None, norm_group=None):
        """Batched whole-path transcription of independent utterances (<= 4096; wider than 16: continuous batching over decode slots): list of float32 sample arrays
        (or device pointers + lengths) -> list of id arrays.  Decode steps are batched so weights stream once per step.
        norm_group (vox_transcribe_batch_ex): per unit, the id of the FILE whose peak normalises it (the CLI's semantics: normalise the file, then chunk it,
        bin/transcribe.rs:207-226); < 0 = use the unit as it is; None = every unit normalises itself."""
        t = _f32(t_embed).reshape(-1)
        if device_ptrs is None:
            arrs = [_f32(x) for x in samples_list]; n = len(arrs)
            ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs]); lens = (C.c_size_t * n)(*[a.size for a in arrs]); kind = 0
        else:
            n = len(device_ptrs); ptrs = (C.c_void_p * n)(*device_ptrs); lens = (C.c_size_t * n)(*n_samples); kind = 1
        caps = [int(lens[i]) // 1280 + 128 for i in range(n)]
        outs = [np.zeros(cp, dtype=np.int32) for cp in caps]
        optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs]); ccaps = (C.c_int32 * n)(*caps); nids = (C.c_int32 * n)()
        if norm_group is None:
            check(lib().vox_transcribe_batch(self.h, n, ptrs, lens, _ptr(t), optrs, ccaps, nids, kind))
        else:
            if len(norm_group) != n:
                raise ValueError("norm_group needs one entry per unit")
            grp = (C.c_int32 * n)(*[int(g) for g in norm_group])
            check(lib().vox_transcribe_batch_ex(self.h, n, ptrs, lens, grp, _ptr(t), optrs, ccaps, nids, kind))
        return [outs[i][:nids[i]].copy() for i in range(n)]

    def timings(self):
        t = _lib.Timings(); check(lib().vox_get_stage_timings(self.h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.Timings._fields_}

    def bench_decode_gemv(self, which, iters=200):
        us = C.c_double(); by = C.c_double(); nm = C.c_char_p()
        check(lib().vox_bench_decode_gemv(self.h, which, iters, C.byref(us), C.byref(by), C.byref(nm)))
        return us.value, by.value, (nm.value or b"").decode()

    def close(self):
        if self.h:
            for c in list(self._caches):
                c.close()
            lib().vox_model_free(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Q4ModelLoader:
    """gguf/loader.rs:67-128"""

    def __init__(self, path=None, reader=None):
        self.path = None if path is None else str(path); self.reader = reader

    @classmethod
    def from_file(cls, path):
        return cls(path)

    @classmethod
    def from_bytes(cls, data):
        """gguf/loader.rs:92-99"""
        return cls(reader=GgufReader.from_bytes(data))

    @classmethod
    def from_shards(cls, shards):
        """gguf/loader.rs:101-107"""
        return cls(reader=GgufReader.from_shards(shards))

    def load(self, ctx: Context, layout_only: bool = False) -> Q4VoxtralModel:
        """`layout_only`: allocate the device arena without reading tensor data (multi-GPU ranks > 0
        receive the arena bytes from rank 0 with one RCCL broadcast)."""
        h = C.c_void_p()
        if self.reader is not None:
            check(lib().vox_q4_model_load_gguf(ctx.h, self.reader.h, 1 if layout_only else 0, C.byref(h)))
        else:
            check(lib().vox_q4_model_load_ex(ctx.h, self.path.encode(), 1 if layout_only else 0, C.byref(h)))
        return Q4VoxtralModel(ctx, h)
