/*
 * voxtral_hip.h -- C ABI of libvoxtral_hip.so: the MI355X (gfx950) implementation of the
 * Voxtral-Mini realtime ASR hot path of TrevorS/voxtral-mini-realtime-rs (reference v0.2.0).
 *
 * This is the drop-in boundary: every entry point is what a Rust `-sys` crate for the
 * reference's model-forward surface (src/audio, src/gguf, src/models) would bind.  The
 * reference interface each function replaces is cited as file:line (paths relative to the
 * reference repository).  INTEGRATION.md shows the Rust-side binding.
 *
 * Conventions
 *   - every function returns int32_t status: 0 = VOX_OK, non-zero = error; the message is
 *     available from vox_last_error() (thread-local).  Where the reference panics (shape
 *     mismatch in q4_matmul, gguf/op.rs:92-100) or returns anyhow::Error, we return a status.
 *   - opaque handles; caller-owned host buffers; plain pointers and sizes only.
 *   - a vox_ctx binds one HIP device + one stream.  Handles created from a ctx are not
 *     thread-safe; distinct ctxs are independent (one ctx per GPU for multi-GPU sharding).
 *   - `mem_kind`: VOX_MEM_HOST = pointers are host memory (copied in/out, synchronous),
 *     VOX_MEM_DEVICE = pointers are device memory on the ctx's device (asynchronous on the
 *     ctx stream; call vox_ctx_synchronize before reading results from another stream).
 *   - layouts follow the reference: weights [N,K] = [out,in] row-major (gguf/tensor.rs:32-34);
 *     Q4_0 block = {f16 d; u8 qs[16]}, element i <-> low nibble of qs[i], element i+16 <-> high
 *     nibble (gguf/tensor.rs:98-109); mel handed to the model as [128][T] (time fastest,
 *     bin/transcribe.rs:295-305) while compute_log returns [T][128] (audio/mel.rs:128); ids are i32.
 *   - there is NO CPU fallback anywhere behind this ABI: compute entry points fail with
 *     VOX_ERR_HIP if no gfx950 device is usable.
 */
#ifndef VOXTRAL_HIP_H
#define VOXTRAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOX_OK 0
#define VOX_ERR_INVALID 1   /* bad argument / shape mismatch            */
#define VOX_ERR_IO 2        /* file / GGUF parse error                  */
#define VOX_ERR_HIP 3       /* HIP runtime error or no device           */
#define VOX_ERR_NOTFOUND 4  /* tensor name not present                  */
#define VOX_ERR_UNSUPPORTED 5

#define VOX_MEM_HOST 0
#define VOX_MEM_DEVICE 1

typedef struct vox_ctx vox_ctx;
typedef struct vox_gguf vox_gguf;
typedef struct vox_q4 vox_q4;
typedef struct vox_model vox_model;
typedef struct vox_cache vox_cache;
typedef struct vox_graph vox_graph;

const char* vox_last_error(void);
int32_t vox_abi_version(void);                      /* bumps on any signature change */
int32_t vox_device_count(int32_t* n);

/* ---- context (reference: the implicit `WgpuDevice::default()`, bin/transcribe.rs:67) ---- */
int32_t vox_ctx_create(int32_t device, vox_ctx** out);
int32_t vox_ctx_destroy(vox_ctx* ctx);
int32_t vox_ctx_synchronize(vox_ctx* ctx);
/* shared != 0: this context SHARES its GPU with other sessions (more contexts of this process -- one host thread each, see vox_model_replicate -- or other processes).
 * The batch entry points then stay off the batched decode engines (their 256 persistent workgroups need the GPU to themselves: next to another session their bounded
 * hand-off waits expire and the session is run twice) and the slot planner prices its steps with the table scaled by ONE measured factor instead of per-form
 * measurements (which scatter under contention) and records none.  Results do not change.  Default 0.  (No reference counterpart: the reference runs one utterance at a
 * time, bin/transcribe.rs:112-126; this is what a multi-threaded host sets on every context of a GPU it runs more than one session on.) */
int32_t vox_ctx_set_shared(vox_ctx* ctx, int32_t shared);
int32_t vox_ctx_stream(vox_ctx* ctx, void** hip_stream_out);   /* hipStream_t, for event timing */
/* device memory helpers so non-HIP callers can use VOX_MEM_DEVICE */
int32_t vox_dev_alloc(vox_ctx* ctx, size_t nbytes, void** out);
int32_t vox_dev_free(vox_ctx* ctx, void* p);
int32_t vox_dev_upload(vox_ctx* ctx, void* dst_dev, const void* src_host, size_t nbytes);
int32_t vox_dev_download(vox_ctx* ctx, void* dst_host, const void* src_dev, size_t nbytes);
int32_t vox_dev_copy(vox_ctx* ctx, void* dst_dev, const void* src_dev, size_t nbytes);   /* device -> device, synchronous */

/* ---- audio front-end (src/audio) ------------------------------------------------------- */
/* resample / resample_to_16k, audio/resample.rs:10-52.  The reference's resampler is rubato 1.0's synchronous FFT resampler
 * (`Fft::<f32>::new(sr_in, sr_out, 1024, 2, 1, FixedSync::Input)` + `process_all_into_buffer`, resample.rs:22-45).  This is that crate's published algorithm on
 * the GPU: blocks of fft_in samples through a Blackman-Harris^2 windowed-sinc filter in the frequency domain, the low bins re-synthesised at length 2 fft_out,
 * overlap-added, the fft_out / 2 samples of delay dropped; n_out = ceil(n_in * (f64(sr_out) / f64(sr_in))); same rate -> copy (resample.rs:17-19).  rubato is not
 * in the reference's tree, so parity against the crate itself is unpinned; the CPU oracle restates the same algorithm independently.
 * Rate pairs whose blocks would not fit a 64 MB matrix (co-prime rates) are refused with VOX_ERR_UNSUPPORTED.  in / out host or device buffers (mem_kind). */
int32_t vox_resample_len(size_t n_in, uint32_t sr_in, uint32_t sr_out, size_t* n_out);
int32_t vox_resample(vox_ctx* ctx, const float* in, size_t n_in, uint32_t sr_in, uint32_t sr_out, float* out, size_t cap, size_t* n_out,
                     int32_t mem_kind);
/* the plan rubato derives from the two rates, and the filter taps (fft_in floats, already divided by 2 fft_in) when taps_or_null is given */
int32_t vox_resample_plan(uint32_t sr_in, uint32_t sr_out, int32_t* fft_in, int32_t* fft_out, int32_t* delay, float* cutoff, float* taps_or_null, size_t cap);
/* AudioBuffer::peak_normalize, audio/io.rs:59-68 (host, in place) */
int32_t vox_peak_normalize(float* samples, size_t n, float target_peak);

/* PadConfig, audio/pad.rs:20-46 */
typedef struct {
    uint32_t sample_rate;             /* 16000 */
    uint32_t n_left_pad_tokens;       /* 76    */
    float    frame_rate;              /* 12.5  */
    uint32_t extra_right_pad_tokens;  /* 17    */
} vox_pad_cfg;
int32_t vox_pad_cfg_voxtral(vox_pad_cfg* cfg);                                   /* PadConfig::voxtral, pad.rs:50-52 */
int32_t vox_pad_len(size_t n, const vox_pad_cfg* cfg, size_t* out);              /* pad.rs:89-93 */
int32_t vox_pad_audio(const float* in, size_t n, const vox_pad_cfg* cfg, float* out); /* pad_audio, pad.rs:89-103 */
int32_t vox_num_audio_tokens(size_t n, const vox_pad_cfg* cfg, size_t* out);     /* pad.rs:106-108 */

/* ChunkConfig / chunk_audio / needs_chunking, audio/chunk.rs:9-166 */
typedef struct {
    uint32_t max_mel_frames;  /* 1500; CLI default 1200 (bin/transcribe.rs:55-57) */
    uint32_t hop_length;      /* 160 */
    uint32_t sample_rate;     /* 16000 */
    uint32_t overlap_frames;  /* 0 */
} vox_chunk_cfg;
typedef struct { size_t start_sample, end_sample, index; int32_t is_last; } vox_chunk;
int32_t vox_needs_chunking(size_t n, const vox_chunk_cfg* cfg, int32_t* out);
int32_t vox_chunk_plan(size_t n, const vox_chunk_cfg* cfg, vox_chunk* out, size_t cap, size_t* n_chunks);

/* MelSpectrogram (MelConfig::voxtral), audio/mel.rs:63-350 */
int32_t vox_mel_num_frames(size_t n_samples, size_t* out);                       /* mel.rs:175-182 */
int32_t vox_mel_filterbank(float* out_128x201);                                  /* mel.rs:288-339 */
int32_t vox_hann_window(int32_t length, float* out);                             /* mel.rs:345-349 */
/* compute_log, mel.rs:128-165: samples (already padded by the caller) -> [T][128].  STFT + mel
 * filterbank + log run on the GPU. */
int32_t vox_mel_compute_log(vox_ctx* ctx, const float* samples, size_t n, float* out_Tx128, int32_t mem_kind);

/* TimeEmbedding::embed, models/time_embedding.rs:41-71 (theta 10000) */
int32_t vox_time_embedding(float t, int32_t dim, float* out);

/* ---- GGUF reader (src/gguf/reader.rs:98-223) -------------------------------------------- */
int32_t vox_gguf_open(const char* path, vox_gguf** out);                         /* GgufReader::open */
/* GgufReader::from_bytes (gguf/reader.rs:98-103): parse an image already in host memory (BORROWED until vox_gguf_close) */
int32_t vox_gguf_open_memory(const void* data, size_t size, vox_gguf** out);
/* Q4ModelLoader::from_shards (gguf/loader.rs:101-107, the WASM <= 512 MB pieces): consecutive pieces of one GGUF image (copied) */
int32_t vox_gguf_open_shards(const void* const* shards, const size_t* sizes, int32_t n, vox_gguf** out);
int32_t vox_gguf_close(vox_gguf* g);
int32_t vox_gguf_version(const vox_gguf* g, uint32_t* out);
int32_t vox_gguf_tensor_count(const vox_gguf* g, uint64_t* out);
int32_t vox_gguf_tensor_name(const vox_gguf* g, uint64_t index, const char** out); /* tensor_names */
/* tensor_info: dims in file order (GGUF = reversed PyTorch order), dtype 0 F32 / 1 F16 / 2 Q4_0 */
int32_t vox_gguf_tensor_info(const vox_gguf* g, const char* name, uint64_t dims[4], uint32_t* ndims,
                             uint32_t* dtype, uint64_t* nbytes);
int32_t vox_gguf_tensor_data(const vox_gguf* g, const char* name, void* dst, size_t cap); /* tensor_data */

/* ---- Q4 operator boundary (src/gguf/tensor.rs, op.rs, linear.rs) ------------------------ */
/* Q4Tensor::from_q4_bytes, gguf/tensor.rs:35-71: validates N*K % 32 == 0 and nbytes == blocks*18 */
int32_t vox_q4_tensor_from_bytes(vox_ctx* ctx, const uint8_t* raw, size_t nbytes, int64_t N, int64_t K, vox_q4** out);
int32_t vox_q4_tensor_shape(const vox_q4* q, int64_t* N, int64_t* K);            /* tensor.rs:74-76 */
int32_t vox_q4_tensor_num_blocks(const vox_q4* q, int64_t* out);                 /* tensor.rs:79-81 */
int32_t vox_q4_tensor_dequantize(vox_ctx* ctx, const vox_q4* q, float* out_NxK); /* tensor.rs:88-113 (host out) */
int32_t vox_q4_tensor_free(vox_q4* q);
/* q4_matmul, gguf/op.rs:86-137: out[B,M,N] = x[B,M,K] x W[N,K]^T, f32 accumulate.
 * B*M <= 4 runs the GEMV kernel (reference: tiled shader), larger the MFMA GEMM (reference: naive). */
int32_t vox_q4_matmul(vox_ctx* ctx, const vox_q4* w, const float* x, int32_t B, int32_t M, float* out, int32_t mem_kind);
/* Q4Linear::forward, gguf/linear.rs:34-40: q4_matmul (+ bias[N], may be NULL; same mem_kind) */
int32_t vox_q4_linear_forward(vox_ctx* ctx, const vox_q4* w, const float* bias_or_null, const float* x,
                              int32_t B, int32_t M, float* out, int32_t mem_kind);

/* ---- dense (f32 SafeTensors path) layer operators on their own (src/models/layers) ------ */
/* The device form of an F32 burn `Linear` weight [N][K] as models/weights.rs:16-66 loads it: the exact f32 plane + bf16 hi / lo planes (what vox_f32_model_load builds
 * for a checkpoint that is not bf16-representable).  w_other_or_null: a second [N][K] tensor interleaved row by row with the first (row 2 i = w[i], 2 i + 1 =
 * other[i]) -- the fused gate | up operand SwiGLU::forward (models/layers/swiglu.rs:72-77) runs on.  The handle is a vox_q4: vox_q4_tensor_shape / _free apply,
 * vox_q4_linear_forward / vox_q4_matmul multiply by it. */
int32_t vox_dense_tensor_from_f32(vox_ctx* ctx, const float* w_NxK, const float* w_other_or_null, int64_t N, int64_t K, vox_q4** out);
/* Linear::forward with a fused epilogue, Q4 or dense weight: 0 none, 1 GELU (the Ada t_cond MLP, models/layers/rms_norm.rs:109-118), 2 SwiGLU over interleaved
 * gate / up rows: out[.][i] = silu(row 2 i) * row 2 i + 1, N / 2 columns (swiglu.rs:72-77; no bias). */
int32_t vox_linear_forward_ex(vox_ctx* ctx, const vox_q4* w, const float* bias_or_null, const float* x, int32_t B, int32_t M, float* out, int32_t epilogue, int32_t mem_kind);
/* ConvDownsampler::forward, models/layers/conv.rs:78-83: gelu(conv1d k = 3, s = 2, p = 1) twice; x [C][L], w1 [O][C][3], w2 [O][O][3] -> out [O][L2], L2 = ((L + 1) / 2 + 1) / 2
 * (host pointers; the im2col MFMA path the encoder's conv stem runs on). */
int32_t vox_conv_downsample(vox_ctx* ctx, const float* x_CxL, int32_t C, int32_t L, const float* w1, const float* b1, const float* w2, const float* b2, int32_t O, float* out_OxL2);

/* ---- model-forward surface (src/gguf/loader.rs, src/gguf/model.rs) ---------------------- */
typedef struct {
    int32_t enc_layers, enc_dim, enc_heads, enc_head_dim, enc_ffn, enc_window;
    int32_t dec_layers, dec_dim, dec_heads, dec_kv_heads, dec_head_dim, dec_ffn, dec_window, vocab;
    int32_t n_mels, reshape_factor, t_cond_dim;
    float rope_theta, norm_eps;
} vox_model_cfg;

/* Attention core of both stacks: softmax(q k^T * head_dim^-0.5 + causal/sliding-window mask) v with grouped-query heads
 * (gguf/model.rs:100-120 encoder MHA, :125-198 decoder GQA without materialising the x4 KV expansion; masking.rs:9-107:
 * query at position offset+m sees keys j <= offset+m and, when window >= 0, offset+m-j <= window).
 * q [M][n_heads*head_dim], k / v [kv_len][n_kv_heads*head_dim], out [M][n_heads*head_dim]; head_dim 64 or 128. */
int32_t vox_attention(vox_ctx* ctx, const float* q, const float* k, const float* v, int32_t M, int32_t kv_len, int32_t n_heads,
                      int32_t n_kv_heads, int32_t head_dim, int32_t offset, int32_t window, float* out, int32_t mem_kind);

/* Q4ModelLoader::from_file(..).load(), gguf/loader.rs:82-128 */
int32_t vox_q4_model_load(vox_ctx* ctx, const char* gguf_path, vox_model** out);
/* flags: VOX_LOAD_LAYOUT_ONLY parses the GGUF header and allocates the identical device arena layout but does
 * not read or upload tensor data -- the caller fills the arena (vox_model_arena) e.g. from an RCCL broadcast. */
#define VOX_LOAD_LAYOUT_ONLY 1u
int32_t vox_q4_model_load_ex(vox_ctx* ctx, const char* gguf_path, uint32_t flags, vox_model** out);
/* Q4ModelLoader::from_bytes / from_shards -> load (gguf/loader.rs:92-128): from an open reader (file, memory image or shards);
 * the reader is only read during the call and stays owned by the caller. */
int32_t vox_q4_model_load_gguf(vox_ctx* ctx, vox_gguf* g, uint32_t flags, vox_model** out);
/* VoxtralModelLoader::from_file(..).load(), models/loader.rs:35-78: the f32 SafeTensors path (F32 / F16 / BF16 tensors,
 * models/weights.rs:16-66).  Same vox_model handle and the same forward entry points as the Q4 model.  A linear weight whose values
 * are all bf16-representable (the published checkpoint is BF16) is stored as one bf16 plane; any other F32 / F16 tensor keeps its EXACT
 * values on device (f32 plane for the decode GEMV and the embedding lookup, bf16 hi + lo planes for the MFMA GEMMs) -- tested at full size
 * against the oracle (tests/test_gpu_f32_path.py, test_gpu_fullsize.py: 108 / 108 and 196 / 196 greedy ids). */
int32_t vox_f32_model_load(vox_ctx* ctx, const char* safetensors_path, vox_model** out);
int32_t vox_model_free(vox_model* m);
int32_t vox_model_config(const vox_model* m, vox_model_cfg* out);
int32_t vox_model_weight_bytes(const vox_model* m, uint64_t* out);   /* device bytes of the weight arena */
/* Multi-GPU: export / import the packed device weight arena so rank 0 can parse the GGUF once and
 * the other ranks receive it with one RCCL broadcast over xGMI (no data-path collective). */
int32_t vox_model_arena(const vox_model* m, void** dev_ptr, uint64_t* nbytes);
/* Receiver side: after the bytes of vox_model_arena (the PRIMARY part of the arena: every tensor as parsed from the file, 2.5 GB for the
 * Q4 model) have been written into a VOX_LOAD_LAYOUT_ONLY model, rebuild what is derived from them on this GPU (the tile-ordered copies
 * of the Q4 linears; the decode engine's weight stream is packed lazily at the first decode step on every rank). */
int32_t vox_model_arena_finalize(vox_model* m);
/* In-process multi-GPU start-up (one host thread + one vox_ctx + one replica per GPU: the shape SURVEY.md section 8(e) gives the reference's serial loop,
 * bin/transcribe.rs:112-126): one more replica of a loaded Q4 model on `dst_ctx` -- another GPU or the same one -- without the file and without a collective library.
 * The destination arena is laid out from the source's tensor manifest, its primary part is copied device to device (hipMemcpyPeerAsync: xGMI between two GPUs) and the
 * derived copies are rebuilt on the destination (as vox_model_arena_finalize).  The replica is independent of the source afterwards (own arena, own workspaces); distinct
 * contexts may be driven from distinct host threads concurrently (tests/test_gpu_model.py::test_two_contexts_two_threads). */
int32_t vox_model_replicate(const vox_model* src, vox_ctx* dst_ctx, vox_model** out);
/* Device memory the model holds besides caches and workspaces, bytes: out[0] weight arena (= vox_model_weight_bytes), out[1] its primary part (what a multi-GPU start-up
 * broadcasts), out[2] the decode engines' weight stream (a third copy of the decoder's Q4 bytes in consumption order; 0 until the first decode step builds it),
 * out[3] the engines' edge buffers (single-stream granules + one block per 16-row group of the batched engine). */
int32_t vox_model_memory(const vox_model* m, uint64_t out[4]);

/* the delay conditioning used by every decoder call: t_embed = TimeEmbedding(dec_dim).embed(delay)
 * (bin/transcribe.rs:104-105).  Ada scales 1 + w2(gelu(w0 t_embed)) (gguf/model.rs:250-255) are
 * loop-invariant and cached per t_embed. */
int32_t vox_model_set_t_embed(vox_model* m, const float* t_embed_host);
/* Single-stream decode loop of transcribe_streaming (gguf/model.rs:938-960): by default every step is ONE launch of the persistent decode engine
 * (all 26 layers + final norm + lm_head; real decoder geometry, Q4 weights, 256-CU device); on = 0 selects the per-operator launches (4 per layer),
 * which other geometries / dense checkpoints always use.  *active_or_null reports whether the engine will be used.  Results of both paths agree to
 * summation-order noise (same ids; tests/test_gpu_fullsize.py).  The engine's 256 workgroups wait for each other with bounded (20 ms) waits: if the GPU is shared and a wait expires,
 * the utterance is decoded again on the per-operator launches (a warning on stderr), the engine is re-armed for the next utterance and switched off after three such
 * strikes.  Environment VOX_ENGINE=0 sets the default to off at load time. */
int32_t vox_model_set_decode_engine(vox_model* m, int32_t on, int32_t* active_or_null);
/* Batched decode loop of vox_transcribe_batch (BASELINE configs[3] / [4]; the reference's model.rs:938-960 is batch-1): the 26 decoder layers of a step run as ONE
 * launch of the batched decode-layer engine (same eligibility as above) whenever one or two 16-row groups are active -- a batch of n <= 16 rows (one group per
 * launch), and in a wider batch's continuous decode the steps with one or two active slot groups (TWO groups per launch: group B's phase runs while group A's
 * hand-off resolves; DESIGN.md sections 3.3c / 3.3e).  Steps with three or four active groups take the launch-based step (4 launches per layer and group, the groups'
 * chains forked on side streams: measured faster than engine launches back to back).  on = 0 selects the launch-based step everywhere, on < 0 only queries.
 * *active_or_null: is the engine armed; *launches_or_null: engine launches enqueued so far (eager + graph replays) -- the number that says whether a given call used
 * it.  A hand-off timeout inside the engine re-runs the batch (the session, for a wide batch) on the launch-based step (a warning on stderr); three such strikes
 * switch the engine off for the model.  Environment VOX_BATCH_ENGINE=0: off at load time. */
int32_t vox_model_set_batch_engine(vox_model* m, int32_t on, int32_t* active_or_null, uint64_t* launches_or_null);

/* Q4VoxtralModel::encode_audio, gguf/model.rs:783-788: mel [128][T] -> [S][dec_dim]; *S = floor(S_enc/4) */
int32_t vox_encode_audio(vox_model* m, const float* mel_128xT, int32_t T, float* out, int32_t cap_rows,
                         int32_t* S, int32_t mem_kind);
/* Q4VoxtralModel::transcribe_streaming, gguf/model.rs:873-963 -> ids, length S-38 (0 if S < 38).
 * logits_or_null (host, [n_ids][vocab]) is a parity/debug tap for the per-step decoder logits. */
int32_t vox_transcribe_streaming(vox_model* m, const float* mel_128xT, int32_t T, const float* t_embed,
                                 int32_t* out_ids, int32_t cap, int32_t* n_ids, float* logits_or_null,
                                 int32_t mem_kind);
/* Whole hot path from 16 kHz samples: peak_normalize(0.95) -> pad -> log-mel -> transcribe_streaming
 * (bin/e2e_bench.rs:98-232 un-chunked pipeline).  samples may be device-resident (VOX_MEM_DEVICE). */
int32_t vox_transcribe_audio(vox_model* m, const float* samples, size_t n, const float* t_embed,
                             int32_t* out_ids, int32_t cap, int32_t* n_ids, int32_t mem_kind);

/* Batched transcription (BASELINE.json configs[3] "Batch=16 x 16 s utterances" and configs[4], a rank's share of a corpus; extension -- the reference's callers
 * loop over files, bin/transcribe.rs:112-126).  n <= 4096 independent utterances, each through the whole path of vox_transcribe_audio; the decode loop advances
 * many sequences per step so the weights are streamed once per step for all of them.  samples[i] / out_ids[i] are per-utterance buffers (samples host or device per
 * mem_kind, ids always host); n_ids[i] receives S_i - 38 (or 0).  Batches may be ragged and results always land in the caller's slot i.
 * n <= 16: one 16-row group, one decode-layer engine launch per step.  n > 16: CONTINUOUS BATCHING -- every utterance is encoded (stacked, packed: no padding to the
 * longest) and prefilled up front; the decode step then runs over 16 .. 128 SLOTS, and a slot whose utterance has its last token takes the next utterance of its
 * host-planned queue inside the same step (token counts are a pure function of the sample count: there is no EOS, gguf/model.rs:936-960), so the groups stay full
 * until the queues run dry; steps with one or two active groups are one engine launch for all their layers (vox_model_set_batch_engine).  Ids per utterance do not
 * depend on n, on the slot or on the neighbours (tested at full size). */
int32_t vox_transcribe_batch(vox_model* m, int32_t n, const float* const* samples, const size_t* n_samples, const float* t_embed,
                             int32_t* const* out_ids, const int32_t* caps, int32_t* n_ids, int32_t mem_kind);
/* The same call with the CLI's normalisation semantics (bin/transcribe.rs:207-265): the reference peak-normalises the FILE once, splits it into chunks of
 * --max-mel-frames (default 1200, :55-57; audio/chunk.rs:125-166) and transcribes every chunk as an independent unit (own padding, own 38-token prefix), joining the
 * chunk texts with " " (:261-275).  Here a chunk is a unit of this call: units naming the same norm_group[i] >= 0 share ONE peak scale 0.95 / max|x| taken over all of
 * them (the chunks tile their file, so that is the file's peak: one device reduction per unit folded per group, exact); norm_group[i] < 0: the unit is used as handed
 * over (the caller normalised it); norm_group == NULL: every unit normalises itself (= vox_transcribe_batch, the un-chunked e2e-bench pipeline).  Units may be views
 * into one file buffer (host or device).  Ids per unit equal vox_transcribe_streaming's on the chunk's mel (tests/test_gpu_fullsize.py, test_tokenizer_cli.py). */
int32_t vox_transcribe_batch_ex(vox_model* m, int32_t n, const float* const* samples, const size_t* n_samples, const int32_t* norm_group_or_null, const float* t_embed,
                                int32_t* const* out_ids, const int32_t* caps, int32_t* n_ids, int32_t mem_kind);
/* sessions = 2..4: vox_transcribe_batch / _ex calls with at least 128 units per session run as that many CONCURRENT sessions on the model's GPU -- the calling thread on the
 * model's context, the others on library threads with hidden contexts and replicas of the model (made here, device to device: 2.5 GB each; freed with the model or by
 * sessions = 1).  One session leaves the GPU idle wherever its launch-bound decode steps wait; a second one fills the gaps: 647 FLEURS-like clips x 1.16 - 1.20 over a 64-slot session on one
 * MI355X (the plain call now plans up to 128 slots as two chains per step -- the same overlap -- so sessions add ~1 % there: they are for one GPU shared by
 * independent callers).  Units that share a norm_group stay in one session; every context involved counts as shared for the call (vox_ctx_set_shared); results are per
 * unit and do not depend on the split.  vox_get_stage_timings then reports the longest session's stage times and the call's wall time.  Smaller calls and every other
 * entry point are unchanged.  (No reference counterpart -- the reference transcribes one file at a time, bin/transcribe.rs:112-126; a host that prefers its own threads
 * uses vox_model_replicate + vox_ctx_set_shared instead.)  Q4 (GGUF) models. */
int32_t vox_model_set_sessions(vox_model* m, int32_t sessions);

/* Q4LanguageModel pieces used directly by e2e-bench / WASM (gguf/model.rs:566,665,680,711) */
int32_t vox_decoder_cache_create(vox_model* m, int32_t max_seq, vox_cache** out);   /* create_cache_preallocated */
int32_t vox_cache_free(vox_cache* c);
int32_t vox_cache_seq_len(const vox_cache* c, int32_t* out);                         /* KVCache::seq_len */
int32_t vox_cache_reset(vox_cache* c);
/* KVCache::update on layer `layer` of a pre-allocated cache (models/layers/kv_cache.rs:116-136: slice_assign of k / v [1][heads][n_rows][head_dim] at rows pos ..
 * pos + n_rows); the length shared by all layers (LayerCaches::seq_len, :242-244) becomes max(len, pos + n_rows).  heads = dec_kv_heads (decoder cache) / enc_heads. */
int32_t vox_cache_update(vox_cache* c, int32_t layer, int32_t pos, const float* k_HxNxhd, const float* v_HxNxhd, int32_t n_rows, int32_t mem_kind);
/* forget the rows from `len` on (0 <= len <= seq_len): the next forward appends at `len` again.  vox_cache_update / _truncate / _reset first settle the decoder steps of
 * this cache that have not been verified yet (see vox_forward_hidden_with_cache_ex): a hand-off timeout found then is THIS call's error (VOX_ERR_HIP) and ALL steps since
 * the last synchronisation are taken back. */
int32_t vox_cache_truncate(vox_cache* c, int32_t len);
/* Streaming encoder: Q4AudioEncoder::create_cache + Q4VoxtralModel::encode_audio_with_cache (gguf/model.rs:437-459,791-799; per layer
 * :299-317,125-174), eviction KVCache::apply_sliding_window (kv_cache.rs:176-203).  The chunk's conv output rows are run through the 32 layers
 * against the cached K / V (RoPE at the absolute stream position; the cache evicts rows older than the 750-row window by itself when a chunk
 * does not fit), then reshaped / adapted like encode_audio: floor(S_chunk / 4) rows of [dec_dim].  capacity_rows 0 = 2 * window + 512.
 * LIMIT: positions are ABSOLUTE stream positions (the reference offsets RoPE by cache.seq_len(), which restarts after an eviction); the position table
 * holds 65 536 rows = 21.8 minutes of audio per cache -- later chunks are refused (VOX_ERR_INVALID) until vox_cache_reset.  A too small `cap_rows` is refused
 * BEFORE the chunk is appended (the stream cache is untouched and the call can be repeated with a larger buffer). */
int32_t vox_encoder_cache_create(vox_model* m, int32_t capacity_rows, vox_cache** out);
int32_t vox_encoder_cache_apply_sliding_window(vox_cache* enc_cache, int32_t window);
int32_t vox_cache_abs_pos(const vox_cache* c, int32_t* out);      /* encoder cache: stream positions seen so far; decoder cache: == seq_len */
int32_t vox_encode_audio_with_cache(vox_model* m, const float* mel_128xT, int32_t T, vox_cache* enc_cache, float* out, int32_t cap_rows,
                                    int32_t* S, int32_t mem_kind);
int32_t vox_embed_tokens_from_ids(vox_model* m, const int32_t* ids, int32_t n, float* out_nxD);      /* host out */
int32_t vox_forward_hidden_with_cache(vox_model* m, const float* x_MxD, int32_t M, const float* t_embed,
                                      vox_cache* cache, float* out_MxD);                              /* host in/out */
int32_t vox_lm_head(vox_model* m, const float* hidden_MxD, int32_t M, float* logits_MxV);            /* host in/out */
/* Device-resident forms of the same surface (mem_kind as everywhere else; VOX_MEM_DEVICE copies nothing and synchronises nothing), for a caller that drives
 * decode piece by piece the way bin/e2e_bench.rs:179-224 and web/bindings.rs:357-424 do on Burn tensors:
 *     embed_tokens_from_ids -> audio_pos + text_embed -> forward_hidden_with_cache -> lm_head -> argmax(2) -> into_scalar
 * All workspaces are model-owned (no allocation per call).  A single-row forward_hidden_with_cache against a decoder cache of <= 1024 rows runs as ONE launch of the
 * persistent decode engine when that is active (vox_model_set_decode_engine) -- the launch that also computes the row's lm_head; vox_lm_head_ex / vox_lm_head_argmax
 * on the hidden buffer it handed out (`*hidden_ws`, READ-ONLY for the caller, valid until the next decoder call on the model) then return those logits / that token
 * instead of streaming the lm_head again.  Any other hidden pointer is multiplied for real.  Engine hand-off timeouts (shared GPU) surface as VOX_ERR_HIP at the next
 * synchronising call -- vox_lm_head_argmax, vox_argmax_rows, vox_ctx_synchronize, host-kind outputs -- i.e. in the step that failed when the caller reads a token per
 * step as the reference's loop does.  The library remembers which rows of the cache it has not verified yet (the engine steps since the last synchronisation and
 * anything appended behind them) and takes exactly those back: after the error vox_cache_seq_len() is the length before the failed step, REPEAT THE STEP (same
 * token, same position) and the ids are those of an undisturbed run (tests/test_gpu_fullsize.py::test_full_piecewise_surface_recovers_from_an_engine_timeout).
 * ids are always host memory (the reference passes &[i32]). */
int32_t vox_embed_tokens_from_ids_ex(vox_model* m, const int32_t* ids_host, int32_t n, float* out_nxD, int32_t mem_kind);
/* `audio_pos + text_embed` (bin/e2e_bench.rs:212, gguf/model.rs:902,946): out[i] = a[i] + b[i] on the context's stream */
int32_t vox_tensor_add(vox_ctx* ctx, const float* a, const float* b, size_t n, float* out, int32_t mem_kind);
/* out_MxD_or_null: where to copy the rows (may be NULL with VOX_MEM_DEVICE when hidden_ws_or_null is given); *hidden_ws_or_null: the model-owned device buffer holding them */
int32_t vox_forward_hidden_with_cache_ex(vox_model* m, const float* x_MxD, int32_t M, const float* t_embed, vox_cache* cache, float* out_MxD_or_null,
                                         const float** hidden_ws_or_null, int32_t mem_kind);
int32_t vox_lm_head_ex(vox_model* m, const float* hidden_MxD, int32_t M, float* logits_MxV, int32_t mem_kind);
/* `logits.argmax(2)` + the scalar read-back (bin/e2e_bench.rs:219-220; lowest index wins ties): ids_host[M]; synchronises the stream */
int32_t vox_argmax_rows(vox_ctx* ctx, const float* logits_MxV, int32_t M, int32_t V, int32_t* ids_host, int32_t mem_kind);
/* lm_head + argmax + read-back in one call: M token ids come back instead of M x 512 KB of logits; synchronises the stream */
int32_t vox_lm_head_argmax(vox_model* m, const float* hidden_MxD, int32_t M, int32_t* ids_host, int32_t mem_kind);
/* Q4VoxtralModel::generate_step_with_cache, gguf/model.rs:857-867 (text tokens only: embed -> decoder against the cache -> final norm -> lm_head) in one call;
 * token_ids[n] host, logits[n][vocab] host; the cache advances by n. */
int32_t vox_generate_step_with_cache(vox_model* m, const int32_t* token_ids, int32_t n, const float* t_embed, vox_cache* cache, float* logits_nxV);

/* Composite forwards, mel [128][T] -> logits [S][vocab] in one call (nothing leaves the device between the stages); *S = decoder positions = floor(S_enc / 4):
 *   vox_forward             Q4VoxtralModel::forward            gguf/model.rs:820-830   the audio embeddings alone are the decoder input
 *   vox_forward_streaming   Q4VoxtralModel::forward_streaming  gguf/model.rs:802-816   audio embeddings + embed(token_ids); n_ids must equal S (vox_num_audio_tokens tells it in advance)
 *   vox_forward_with_cache  Q4VoxtralModel::forward_with_cache gguf/model.rs:833-843   encode_audio_with_cache + forward_hidden_with_cache against the caller's two caches
 * mel / logits host or device per mem_kind; token ids and t_embed host. */
int32_t vox_forward(vox_model* m, const float* mel_128xT, int32_t T, const float* t_embed, float* logits_SxV, int32_t cap_rows, int32_t* S, int32_t mem_kind);
int32_t vox_forward_streaming(vox_model* m, const float* mel_128xT, int32_t T, const int32_t* token_ids, int32_t n_ids, const float* t_embed, float* logits_SxV,
                              int32_t cap_rows, int32_t* S, int32_t mem_kind);
int32_t vox_forward_with_cache(vox_model* m, const float* mel_128xT, int32_t T, const float* t_embed, vox_cache* enc_cache, vox_cache* dec_cache, float* logits_SxV,
                               int32_t cap_rows, int32_t* S, int32_t mem_kind);

/* stage timers, BenchmarkResult parity (bin/e2e_bench.rs:62-74): ms of the last transcribe call */
typedef struct { double preprocess_ms, encode_ms, decode_ms, total_ms; int32_t decode_tokens; int32_t graph_replays; } vox_timings;
int32_t vox_get_stage_timings(const vox_model* m, vox_timings* out);

/* The VOX_* measurement knobs (kernel-selection overrides used by tools/ and by the A/B tests) are read from the environment ONCE, at vox_ctx_create;
 * this re-reads them (tests that flip a knob between two calls).  Not part of the reference surface. */
int32_t vox_debug_reload_knobs(void);
/* Test hook: launch `workgroups` x 1024-thread workgroups that spin for `micros` microseconds on a side stream of the context and return at once (vox_ctx_synchronize does
 * not wait for them; vox_ctx_destroy does).  Used to test the decode engines against a GPU that is not theirs alone (tests/test_gpu_fullsize.py). */
int32_t vox_debug_occupy(vox_ctx* ctx, int32_t workgroups, int32_t micros);

/* ---- measurement hooks (bench.py roofline leg; not part of the reference surface) -------- */
/* Launch the decode-step Q4 GEMV of decoder layer `layer` (`which`: 0 qkv, 1 wo, 2 w1w3, 3 w2, 4 lm_head; 5 = the whole step as one decode-engine launch)
 * `iters` times on the ctx stream, cycling layers so weights stay HBM-cold; returns the average
 * launch duration measured with hipEvents on that stream and the algorithmic bytes per launch. */
int32_t vox_bench_decode_gemv(vox_model* m, int32_t which, int32_t iters, double* avg_us, double* bytes_per_launch,
                              const char** kernel_name);

/* The wide decode step's operators on their own (tools/wide_bench.py): operator `which` (0 q|k|v, 1 wo, 2 w1|w3, 3 w2, 4 lm_head) over `mt` = 2..4 slot groups of 16 rows,
 * `iters` launches cycling the layers; out_us[0] the GEMM launch, [1] its finishing launch, [2] both, [3] the same operator as `mt` 16-row launches back to back. */
int32_t vox_bench_wide(vox_model* m, int32_t which, int32_t mt, int32_t iters, double out_us[4]);

/* Measurement builds only (the library compiled with -DVOX_TIMELINE; VOX_ERR_UNSUPPORTED otherwise): every decode-step GEMV /
 * attention launch after _start takes the next of `n_slots` slots and each of its first `n_waves` waves stamps the 100 MHz
 * s_memrealtime counter at 4 points; _fetch copies out[n_slots][n_waves][4] back and switches the instrumentation off. */
int32_t vox_debug_timeline_start(vox_ctx* ctx, int32_t n_slots, int32_t n_waves);
int32_t vox_debug_timeline_fetch(vox_ctx* ctx, uint64_t* out, size_t cap_words, int32_t* slots_used, int32_t* meta /* [n_slots][4] {0 gemv / 1 attention, epilogue, N, K}, may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* VOXTRAL_HIP_H */
