// vox_kernels.h -- host-side launch interface of the gfx950 kernels (vox_kernels.hip).
// Everything here is internal to libvoxtral_hip.so; the public boundary is include/voxtral_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
// "done once per DEVICE" flag for per-kernel attributes (hipFuncSetAttribute applies to the current device: one vox_ctx per GPU in one process)
struct DevOnce {
    std::atomic<uint64_t> mask{0};
    bool done(int dev) const { return (mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull; }
    void set(int dev) { mask.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};
static inline int vox_current_device() { int d = 0; if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; } return d; }
#include <stdint.h>

namespace vox {

// Q4_0 weight [N][K] re-packed at load (bit-lossless) from the 18-byte AoS GGUF blocks
// (gguf/tensor.rs:98-109) into two 16-byte-aligned planes so a wavefront streams coalesced
// global_load_dwordx4:  qs[n][b] = the block's 16 nibble bytes (element i <-> low nibble of
// byte i, element i+16 <-> high nibble), sc[n][b] = the block's f16 scale.
struct Q4W {
    const uint4* qs;     // fmt 0: nibble chunks [N][nb]; fmt 1: dense bf16 weights [N][K] viewed as uint4 (8 bf16 each)
    const uint16_t* sc;  // fmt 0: IEEE f16 scale bits [N][nb]; fmt 1: unused (nullptr)
    int N, K, nb;        // nb = K / 32
    int fmt;             // WFMT_Q4_0 or WFMT_BF16 (the f32 SafeTensors path: the checkpoint is BF16 on disk, exact in bf16)
    // optional second copy in MFMA tile order (q4_tile_build_kernel): [N/16 tiles][nb/4][64 lanes] uint4 + scales; used by the
    // skinny (M <= 16) and the large-M MFMA kernels, whose B fragments then are single coalesced dwordx4 loads
    const uint4* qt = nullptr; const uint16_t* st = nullptr;
};
enum WFmt { WFMT_Q4_0 = 0, WFMT_BF16 = 1, WFMT_BF16X2 = 2, WFMT_F32 = 3 };
// WFMT_F32: a dense F32 / F16 checkpoint tensor whose values are NOT bf16-representable (models/weights.rs:16-66 accepts any): qt = the exact f32
// plane [N][K] (decode GEMV, embedding lookup), qs / sc = bf16 hi / lo planes (w ~= hi + lo to 2^-17; the MFMA GEMMs for > 4 rows).   // BF16X2: f32 weights as two dense bf16 planes, qs = hi [N][K], sc = lo [N][K] (conv stem)

enum Epi { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_ROPE_KV = 3, EPI_ARGMAX = 4, EPI_GELU = 5, EPI_SWIGLU_XF = 6, EPI_RESID_XF = 7,
           EPI_ROPE_ROWS = 8 };      // launch_q4_gemm only: store, with RoPE (interleaved pairs, rope.rs:77-141) on the first n_q columns, row m at position pos[m] (or m % rope_seq_rows, or m) -- the encoder's q|k|v operator
// (M <= 16 only) _SWIGLU_XF: SwiGLU written as XF planes; _RESID_XF: out = acc + resid as f32 AND as XF planes of out * xf_w (* xf_w2) plus
// per-workgroup partial sums of squares -- the next RMSNorm is folded into its producer and its consumer (GemmParams::ssq_part)
enum Pro { PRO_NONE = 0, PRO_RMS = 1, PRO_RMS_MUL = 2,      // RMS_MUL: RMSNorm then * mul (the cached Ada scale)
           PRO_RMS_MUL_SUM = 3 };   // RMS_MUL on x + xacc * 2^-32 (the fixed-point wo product accumulated by attn_wo_kernel); the sum is also written to x_out

// ---- fused Q4 GEMV (decode, rows of x <= 4): out[y][n] = epi( sum_k pro(x[y])[k] * W[n][k] )
struct GemvParams {
    Q4W w;
    const float* x; int x_stride;        // gridDim.y input rows
    float* out; int out_stride;          // may be null for EPI_ARGMAX / EPI_ROPE_KV (q goes to out)
    const float* bias;                   // [N] or null
    const float* resid; int resid_stride;// EPI_RESID: out = resid + acc (+ bias)
    const float* gamma; const float* mul; float eps;   // PRO_RMS: x <- (x / rms(x)) * gamma (* mul)
    long long* zero_acc; int zero_n;     // any launch: workgroup b clears its share of zero_acc[0, zero_n) (the layer's attn_wo accumulators, already consumed by w1|w3)
    const long long* xacc; float* x_out;   // PRO_RMS_MUL_SUM: int64 fixed-point (2^-32) accumulators xacc[k]; x_out[k] = x[k] + xacc[k] * 2^-32 (one workgroup per piece writes it)
    const int* pos_ptr; int pos_off;     // position = (pos_ptr ? *pos_ptr : 0) + pos_off
    const float* rope_cos; const float* rope_sin; int hd;   // tables [max_pos][hd/2]
    int n_q, n_k;                        // EPI_ROPE_KV: rows [0,n_q) q, [n_q,n_q+n_k) k, then v
    float* kcache; float* vcache; int cache_head_stride;    // [kv_head][max_seq][hd]
    float* part_val; int* part_idx;      // EPI_ARGMAX: per-workgroup (max, argmax)
    int tl_slot;                         // timeline slot (measurement builds, -DVOX_TIMELINE; assigned by the launcher, -1 = off)
};
// rows_per_wave R in {1,2,4,8}; K tiles of 2048 chosen from K. Returns hipError_t.
hipError_t launch_q4_gemv(const GemvParams& p, int n_rows_x, int pro, int epi, int R, hipStream_t s);
const char* q4_gemv_kernel_name(int K, int pro, int epi, int R);
int q4_gemv_default_R(int N, int K, int epi);
int q4_gemv_grid(int N, int R);
int q4_gemv_nwv(int K, int epi);        // waves per workgroup the launcher will use for this shape (4 / 6 / 12)
int q4_gemv_grid_k(int N, int K, int R, int epi);
int dense_gemv_grid(int N);        // same, for WFMT_BF16 weights   // workgroups launched for N rows at R rows per wave (== number of argmax partials)

// ---- Q4 GEMM on MFMA (prefill / encoder, rows of x > 4): out[M][N'] = epi( x[M][K] * W^T )
struct GemmParams {
    Q4W w;
    const float* x; int x_stride; int M;
    float* out; int out_stride;
    const float* bias;
    const float* resid; int resid_stride;
    const uint4* xf;      // optional (M <= 16): the input as XF fragment planes (see xf_store4) instead of f32 rows x
    uint16_t* xf_scratch; size_t xf_scratch_bytes;   // 17..48 rows (the 38-token prefill): room for ceil(M/16) XF tiles of K columns (64 B per column and tile);
                          // the launcher converts the f32 rows ONCE (xf_rows_kernel) instead of every wave of every workgroup splitting them again
    // fused RMSNorm, consumer side: the XF input holds x * gamma UNnormalised; row m's accumulators are scaled by
    // rstd[m] = 1/sqrt(sum_i ssq_part[i][m] / K + norm_eps) (a per-row scalar commutes with the GEMM); n_part partials of 16 rows
    const float* ssq_part; int n_part; float norm_eps;
    // EPI_RESID_XF, producer side
    uint16_t* xf_out; const float* xf_w; const float* xf_w2; float* ssq_out;
    // EPI_ROPE_KV (M <= 16, one row per sequence): columns [0, n_q) -> RoPE -> out; [n_q, n_q + n_kv*hd) -> RoPE -> K cache; rest -> V cache,
    // all at position pos[m] of sequence m's cache slice
    const int* pos; const float* rope_cos; const float* rope_sin; int hd, n_q, n_kv; float* kc; float* vc; long kv_seq_stride; int kv_head_stride;
    const int* kv_row;    // EPI_ROPE_KV, optional: row m writes into cache slice kv_row[m] instead of slice m (continuous batching: a SLOT of the step decodes whichever utterance it currently holds)
    int tl_slot;          // timeline slot (measurement builds, -DVOX_TIMELINE)
    // split-K (q4_gemm_kernel only, EPI_STORE): blockIdx.z = K slice; slice z writes its partial product to out + z * M * out_stride (bias in slice 0).
    // The consumer (launch_rms_norm_sumk) adds the slices in a fixed order -- for few-column GEMMs (the encoder's N = 1280 w2: 40 K-steps per workgroup)
    int ksplit;
    float* kz_scratch; size_t kz_scratch_bytes;      // 17..48 rows: room for the K-slice planes of q4_skinny_mt2_kernel ([ksplit][M][N] f32); null: the one-dimensional kernel
    // wide decode step (launch_q4_wide): M = 16 * wide_mt rows = wide_mt slot groups of 16; rows of consecutive groups are consecutive in out / resid / pos / kv_row,
    // the XF planes and the partial sums of squares of group g start g * (the group stride) after group 0's
    int rope_seq_rows;    // EPI_ROPE_ROWS without a position table: stacked sequences of this many rows restart at position 0 (0: row m sits at position m)
    int wide_mt; long xf_gstride /* uint4 */, xf_out_gstride /* uint16 */, ssq_part_gstride, ssq_out_gstride /* floats */;
    int wide_rows;        // > 0: q4_wide_kernel stores its K-slice planes ROW-MAJOR, [slice][wide_rows][N] f32 (rows < wide_rows only): the layout the 38-token prefill's finishing kernels read
};
// ---- wide decode step (round 6): 32 / 48 / 64 rows (2..4 slot groups of a continuous batch) through ONE weight fetch and ONE nibble -> bf16 conversion per K step.
// GEMM (q4_wide_kernel: the groups' XF planes staged through LDS once per workgroup, K split over workgroups into planes) + a finishing launch that sums the planes in a
// fixed order and applies the batched step's epilogues (EPI_ROPE_KV / EPI_RESID_XF / EPI_SWIGLU_XF); EPI_STORE with one K slice (the lm_head) stores directly.
struct WidePlan { int kz, sps, ntw; };      // K slices, K steps (128 columns) per slice, n-tiles per wave
bool q4_wide_plan(const Q4W& w, int mt, int epi, WidePlan* pl);
size_t q4_wide_planes_bytes(const Q4W& w, int mt, const WidePlan& pl);
hipError_t launch_q4_wide(const GemmParams& p, int epi, hipStream_t s);      // p.xf (+ xf_gstride), p.wide_mt, p.kz_scratch (planes); epilogue fields as launch_q4_gemm's XF step
hipError_t launch_q4_tile_build(Q4W w, uint4* qt, uint16_t* st, hipStream_t s);
hipError_t launch_q4_gemm(const GemmParams& p, int epi, hipStream_t s);
// dense f32-class GEMM on two bf16 weight planes (w.fmt == WFMT_BF16X2; the conv stem as an im2col GEMM); epi: EPI_STORE / EPI_GELU
hipError_t launch_dense2_gemm(const GemmParams& p, int epi, hipStream_t s);
hipError_t launch_transpose(const float* in, int R, int C, float* out, hipStream_t s);   // [R][C] -> [C][R]
int q4_skinny_resid_xf_parts(int N);   // number of [16]-row partial sums of squares an EPI_RESID_XF launch writes to ssq_out

// ---- small fused ops
hipError_t launch_q4_repack(const uint8_t* raw, uint4* qs, uint16_t* sc, int64_t n_blocks, int nb, int row_mul, int row_add, hipStream_t s);
hipError_t launch_q4_dequant(Q4W w, float* out, hipStream_t s);                 // diagnostics (tensor.rs:88-113)
hipError_t launch_rms_norm(const float* x, int x_stride, int rows, int dim, const float* gamma, const float* mul,
                           float eps, float* out, int out_stride, hipStream_t s);
// x[row] += sum of ksplit partial planes (fixed order; plane stride = part_stride floats), written back, then RMSNorm -> out (dim <= 4096, dim % 4 == 0)
hipError_t launch_rms_norm_sumk(float* x, int x_stride, int rows, int dim, const float* part, size_t part_stride, int ksplit, const float* gamma, float eps, float* out, int out_stride, hipStream_t s);
// pieces of the two-dimensional 17..48-row GEMM (q4_skinny_mt2_kernel) for callers that fold the finishing sum of the K-slice planes into their next kernel
int q4_skinny_mt2_plan(const Q4W& w, int M);                                                                  // K slices, 0 = not applicable
hipError_t launch_xf_rows(const float* x, int x_stride, int M, int K, uint16_t* xf, hipStream_t s);        // f32 rows -> XF tiles
hipError_t launch_q4_skinny_mt2_planes(const GemmParams& p, int KZ, hipStream_t s);                         // p.xf in, planes [KZ][M][N] -> p.kz_scratch
hipError_t launch_splitk_finish_resid(const float* planes, int KZ, int M, int N, float* x, int x_stride, hipStream_t s);
hipError_t launch_splitk_finish_swiglu_xf(const float* planes, int KZ, int M, int N, const float* bias, uint16_t* xf, hipStream_t s);      // planes -> SwiGLU -> XF tiles (K' = N / 2)
hipError_t launch_splitk_finish_rope_kv(const float* planes, int KZ, int M, int N, float* q_out, int q_stride, int n_q, int n_kv, int hd, int pos_off,
                                        const float* cos_t, const float* sin_t, float* kc, float* vc, int head_stride, hipStream_t s);           // planes -> RoPE -> q rows + KV cache
hipError_t launch_rms_norm_xf_sumk(float* x, int x_stride, int rows, int dim, const float* planes, int kz, const float* gamma, const float* mul, float eps, uint16_t* xf, hipStream_t s);
hipError_t launch_rms_norm_xf(const float* x, int x_stride, int rows, int dim, const float* gamma, const float* mul,
                              float eps, uint16_t* xf, hipStream_t s);   // rows <= 16 -> XF fragment planes
// interleaved-pair RoPE in place on columns [0, n_rot) of buf[M][stride]; row m has position pos_off + m
// (seq_rows > 0: stacked sequences of seq_rows rows each, positions restart per sequence)
// (row_pos != nullptr: ragged sequences packed back to back, row m has position pos_off + row_pos[m])
hipError_t launch_rope(float* buf, int M, int stride, int n_rot, int hd, int pos_off, const float* cos_t,
                       const float* sin_t, hipStream_t s, int seq_rows = 0, const int* row_pos = nullptr);
// copy k (columns [k_col, k_col+n_kv*hd)) and v (next n_kv*hd columns) of buf rows into the cache at pos_off+m
hipError_t launch_kv_store(const float* buf, int M, int stride, int k_col, int n_kv, int hd, int pos_off,
                           float* kcache, float* vcache, int cache_head_stride, hipStream_t s, int seq_rows = 0, long kv_seq_stride = 0);

struct AttnParams {
    const float* q; int q_stride;            // q[m][h*hd + d]
    const float* k; const float* v;          // k[kvh*kv_head_stride + j*kv_row_stride + d]
    int kv_row_stride, kv_head_stride;
    float* out; int out_stride;              // out[m][h*hd + d]
    int M, kv_len, n_heads, n_kv_heads, offset, window;   // query m at position offset+m; window<0: none
    const int* pos_ptr;                      // decode: position = *pos_ptr + offset, kv_len = position+1
    // batched decode (gridDim.y = sequences): sequence s reads pos_ptr[s] and its own q / out rows and KV-cache slice
    int pos_per_seq; int q_seq_stride, out_seq_stride; long kv_seq_stride;
    const int* kv_row;    // batched decode, optional: sequence s reads cache slice kv_row[s] instead of slice s (attn_decode_gqa_kernel only: launch_attn_decode forces that kernel)
    // stacked prefill (gridDim.z = sequences): sequence z has seq_len[z] query rows (kv_len = offset + seq_len[z]); p.M = the maximum;
    // q / out / k / v of sequence z start q_seq_stride / out_seq_stride / kv_seq_stride floats after those of z-1
    const int* seq_len;
    const int* seq_row_off;   // stacked prefill, optional (MFMA kernel only): sequence z starts seq_row_off[z] ROWS into q / k / v / out (ragged sequences packed back to back) instead of z * the sequence strides
    uint16_t* out_xf_tiles; long out_xf_tile_stride;      // attn_prefill_small_kernel, optional: the output rows as XF tiles (tile = row / 16; stride in uint16 units) instead of `out`
    uint16_t* out_xf;     // batched decode: write the output rows (row = sequence) as XF fragment planes instead of out
    long out_xf_gstride;  // ... more than 16 sequences in one launch (attn_decode_gqa_kernel): sequence s is row s % 16 of the planes out_xf + (s / 16) * out_xf_gstride (uint16 units)
    int prefer_gqa;       // batched decode: one workgroup per (KV head, sequence) serving its 4 query heads (wide batches)
    int no_xcd_remap;     // measurement knob: keep the linear (head, sequence) workgroup order in attn_decode_kernel
    int tl_slot;          // timeline slot (measurement builds, -DVOX_TIMELINE)
    int spec_zero;        // always 0 (see attn_decode_kernel: keeps the position load a vector load)
    int spec_rows;        // decode (head-major cache, kv_row_stride == hd): rows per KV head in the cache; > 0 lets the kernel request the first
                          // 160 rows before the position word has arrived (attn_decode_core SPEC).  0: indices derived from the position.
};
hipError_t launch_attn_prefill(const AttnParams& p, int hd, hipStream_t s, int n_seq = 1);     // M > 1, causal (+window)
bool attn_prefill_small_ok(const AttnParams& p, int hd, int n_seq);      // will launch_attn_prefill take the short-sequence kernel (<= 48 rows from position 0, f32; may write XF tiles: p.out_xf_tiles)?
hipError_t launch_attn_decode(const AttnParams& p, int hd, int max_seq, hipStream_t s, int n_seq = 1);  // M == 1 per sequence
// single sequence: attention + the wo linear in one launch; acc[wo.N] (int64, zero on entry) receives the product in 2^-32 fixed point (consumer: PRO_RMS_MUL_SUM)
bool attn_wo_supported(const AttnParams& p, const Q4W& wo, int hd, int max_seq);
hipError_t launch_attn_wo(const AttnParams& p, const Q4W& wo, long long* acc, int max_seq, hipStream_t s);

// gelu(conv1d k3 s2 p1): in [Cin][L] -> out; out_token_major: out[t][co] else out[co][t]
hipError_t launch_conv1d_gelu(const float* in, int Cin, int L, const float* w, const float* b, int Cout, float* out,
                              int out_token_major, hipStream_t s);

// log-mel: frames of the (virtually) padded signal zeros(left)+scale*audio+zeros(right); out layout
// [T][128] (transposed=0) or [128][T] (transposed=1). scale_ptr may be null (scale 1).
struct MelTables { const float* window; const float* cos_t; const float* sin_t; const float* fb; const int* fb_lo; const int* fb_hi; };
hipError_t launch_mel(const float* audio, long n, long left, long right, const float* scale_ptr, MelTables t,
                      float* out, int T, int transposed, hipStream_t s);
// sample-rate conversion (rubato `Fft` algorithm as one block matrix, vox_kernels.hip): the matrix builder (once per rate pair) and the block product + overlap-add
hipError_t launch_resample_matrix(const double* H, int new_len, int fft_in, int fft_out, float* At, hipStream_t s);      // H: new_len (re, im) pairs; At: [fft_in][2 fft_out]
hipError_t launch_resample(const float* x, long n_in, const float* At, int fft_in, int fft_out, int delay, float* out, long n_out, hipStream_t s);
hipError_t launch_absmax(const float* x, long n, float target_peak, float* scale_out, hipStream_t s);
// group peaks: every unit folds max|x| into its group's cell (zeroed by the caller); unit_scale[i] = target / peak of unit i's group (1 for a silent group or group < 0)
hipError_t launch_absmax_group(const float* x, long n, unsigned* group_max_cell, hipStream_t s);
hipError_t launch_group_scale(const unsigned* group_max, const int* unit_group, int n, float target, float* unit_scale, hipStream_t s);

// h[i][:] = (audio ? audio[(a_base+i)][:] : 0) + dequant(tok[ids[id_base+i]]) ; bases add *pos_ptr if given
hipError_t launch_embed(Q4W tok, const int* ids, int n, const float* audio, int D, const int* pos_ptr, int id_off,
                        int a_off, float* out, hipStream_t s);
// final argmax over per-workgroup partials (lowest index wins ties); writes tokens[*pos_ptr+tok_off], then *pos_ptr += inc
hipError_t launch_argmax_final(const float* part_val, const int* part_idx, int n_parts, int* tokens, int* pos_ptr,
                               int tok_off, int inc, hipStream_t s);
// fused decode-step tail: tokens[*pos+1] = argmax(partials); *pos += 1; h = audio[*pos] + dequant(tok[tokens[*pos]])
hipError_t launch_argmax_embed(const float* part_val, const int* part_idx, int n_parts, int* tokens, int* pos_ptr, Q4W tok,
                               const float* audio, int D, float* h, hipStream_t s);
// batched decode step helpers (n sequences, per-sequence positions pos[s], caches [seq][kv_head][max_seq][hd] per layer)
hipError_t launch_rope_kv_batch(float* qkv, int n, int stride, int n_q, int n_kv, int hd, const int* pos, const float* cos_t, const float* sin_t,
                                float* kcache, float* vcache, long seq_stride, int head_stride, hipStream_t s);
hipError_t launch_argmax_embed_batch(const float* logits, int n, int vocab, int* tokens, int tok_stride, int* pos, const int* seq_len, Q4W tok,
                                     const float* audio, long audio_seq_stride, int D, float* h, hipStream_t s,
                                     uint16_t* xf = nullptr, const float* xf_w = nullptr, float* ssq_out = nullptr,     // xf: also h * xf_w as XF planes + sum of squares,
                                     long xf_group_stride = 0, int ssq_group_stride = 0,                                  // per group of 16 sequences (strides in elements)
                                     const long* audio_off = nullptr);      // optional: sequence s reads its audio rows at audio + audio_off[s] (floats) instead of audio + s * audio_seq_stride
// continuous batching: the argmax / next-input launch of a decode step over SLOTS (argmax_embed_slots_kernel)
struct SlotStepParams {
    const float* logits; int vocab;                    // [n_slots][vocab] (unused when init)
    int* tokens; int tok_stride; const int* clip_len;  // per utterance: token rows [n_clips][tok_stride], sequence lengths
    int* slot_clip; int* slot_qpos; const int* queue; int q_stride;      // per slot: current utterance (-1: none), index into its queue, queue [n_slots][q_stride] (-1 terminated)
    int* pos; int* kv_row; int n_clips; int first_pos; // per slot: position of the last token written, cache slice (n_clips = the scratch slice of idle slots); first_pos = prefix length
    Q4W tok; const float* audio; const long* audio_off; int D;           // audio rows of utterance c start at audio + audio_off[c] (floats), row stride D
    const float* h0;                                   // [n_clips][D]: first decode input of every utterance
    float* h;                                          // [n_slots][D]: the step's input rows
    uint16_t* xf; const float* xf_w; float* ssq_out; long xf_group_stride; int ssq_group_stride;      // as launch_argmax_embed_batch
    int init;                                          // 1: no argmax, every slot takes the head of its queue
};
hipError_t launch_argmax_embed_slots(const SlotStepParams& p, int n_slots, hipStream_t s);
hipError_t launch_occupy(int workgroups, int micros, hipStream_t s);      // test hook: spin `workgroups` x 1024 threads for `micros` us
hipError_t launch_add_rows(const float* a, const float* b, float* out, long n, hipStream_t s);
hipError_t launch_gelu(float* x, long n, hipStream_t s);
hipError_t launch_argmax_rows(const float* x, int rows, int V, int* out, hipStream_t s);      // out[r] = argmax of row r (lowest index wins ties)

// ---- persistent decode-step engine (vox_engine.hip): the whole single-stream decode step -- 26 layers + final norm + tied lm_head + argmax
// partials -- as ONE launch of 256 workgroups (one per CU) x 8 waves: wave 0 streams this CU's slice of every Q4 operator, in consumption order,
// from a per-CU contiguous copy of the weights into an LDS ring with LDS-DMA (global_load_lds ... nt) and runs ahead of every dependency edge;
// wave 1 moves activations between CUs as 8-byte {value, tag} write-through granules; waves 2..7 consume the ring (VALU dot products, x in registers).
// Replaces gguf/model.rs:938-960 (one decode step) for the real Voxtral decoder geometry; everything else keeps the per-operator launches.
struct EngLayerTab { const float* attn_norm; const float* ffn_norm; const float* ada_mul; float* kc; float* vc; };   // per decoder layer; kc / vc: [n_kv][max_seq][hd]
struct EngParams {
    const unsigned char* stream; size_t cu_stride;   // per-CU weight stream (eng_stream_bytes / 256 bytes each)
    const EngLayerTab* layers; int n_layers;          // device array
    const float* h_in;                                // [D] the step's input embedding
    const float* final_norm;                          // [D]
    const int* pos_ptr; int pos_off;                  // position = *pos_ptr + pos_off
    const float* rope_cos; const float* rope_sin;     // [max_pos][hd/2]
    int max_seq, window; float eps;
    unsigned long long *H0, *H1, *G, *PW, *A, *P2;    // granule buffers: [3072] [3072] [6144] [32][3072] [9216] [24][3072]
    unsigned long long *SS0, *SS1;                    // [256] per-CU partial sums of squares of H0 / H1's raw rows
    unsigned long long* XC;                           // [256] XCC id of every workgroup (start-up exchange)
    unsigned* serial; unsigned* err;                  // launch serial (tags never repeat), first failure code (0 = ok)
    float* part_val; int* part_idx;                   // [256] per-CU argmax partials
    // flags 65536 (the product's graph-replayed steps): the launch BEGINS with the argmax over the previous launch's 256 partials (read back from part_val / part_idx),
    // writes tokens[*pos_rw + 1], runs at that position on its own input rows audio[position] + embed(token) (h_in is not read) and leaves *pos_rw = its position
    int* tokens; int* pos_rw; const uint4* tok_qs; const uint16_t* tok_sc; int tok_nb; const float* audio;
    float* logits_out;                                // optional [vocab]
    int vocab;
    unsigned long long* tl; int tl_layer;             // timeline stamps [256][32] of layer tl_layer (null: off)
    int flags;                                        // product default 641 = 1 | 128 | 512.  1: one LDS-DMA packet in flight while the CU polls memory; 128: XCD-local
                                                      // edges as plain stores through the shared L2 (placement verified per launch); 512: all-gathers swept ag_delay_ticks
                                                      // after the CU's own publish instead of probe-then-sweep.  Measurement / diagnostic: 2 loader re-reads one packet,
                                                      // 4 probe before small sweeps, 32 no LDS-DMA (wrong results), 64 loader pauses while polling, 1024 / 2048 loader depth
                                                      // 2 / 1, 4096 nt polls on XCD-local edges, 16384 FAULT INJECTION (workgroup 7 loses a publish: the timeout test); 65536: see tokens / pos_rw above
    int ag_delay_ticks;                               // flags 512: 10-ns ticks between the CU's own publish and the all-gather sweep
    int pace_ticks;                                   // loader: minimum 10-ns ticks between two packet issues inside the layers (0: none)
};
bool eng_geometry_ok(int D, int n_heads, int n_kv, int hd, int ffn, int vocab, int max_seq);
size_t eng_stream_bytes(int n_layers, int vocab);     // total bytes of the engine weight copy (all 256 CU streams)
size_t eng_state_bytes();                             // granule buffers + serial + err
// op: 0 q|k|v (fused [6144][3072]), 1 wo, 2 w1|w3 (row-interleaved), 3 w2, 4 tied lm_head; layer ignored for op 4; 5: wo for the batched engine into ITS stream (EngBParams::stream_wo)
hipError_t launch_eng_pack(const Q4W& w, int op, int layer, int n_layers, unsigned char* stream, int vocab, hipStream_t s);
void eng_state_carve(unsigned char* state, EngParams* p);      // point p's granule buffers / serial / err into a zero-initialised state block
hipError_t launch_decode_engine(const EngParams& p, hipStream_t s);
// after a launch without flags 65536 has ended: the final norm's OUTPUT of that step (what forward_hidden_with_cache returns, gguf/model.rs:676) from the launch's own last
// all-gather (H0 = h * final_norm * 512 granules, SS0 = 256 partial sums of squares) -> out[3072]
hipError_t launch_eng_hidden(const EngParams& p, float* out, hipStream_t s);
hipError_t eng_occupancy(int* blocks_per_cu);      // resident workgroups per CU the runtime grants decode_engine_kernel
int eng_lds_bytes();

// ---- persistent decode-layer engine for a group of <= 16 sequences (vox_engine_b16.hip): the 26 decoder layers of one batched decode step as ONE launch of 256 workgroups
// (one per CU) x 14 waves on the single-stream engine's packet stream.  Replaces the 26 x 5 launches of the batched step (gguf/model.rs:938-960 for B > 1); the step's tail
// (16-row lm_head GEMM, argmax / next embedding) stays launch-based and reads xf_out / ssq_out.
struct EngBParams {
    const unsigned char* stream;                      // the packet stream built by launch_eng_pack (layer part)
    const unsigned char* stream_wo;                   // wo in the XCD-group K split (launch_eng_pack op 5 per layer; engb_wo_stream_bytes): [layer][packet 2][CU][13824]
    const EngLayerTab* layers; int n_layers;          // device array; kc / vc: the GROUP's first sequence, [sequence][n_kv][max_seq][hd]
    long kv_seq_stride;                               // floats between two sequences' cache slices
    const float* h_in; int h_stride; int n_rows;      // [n_rows][D] the step's input rows (sequences n_rows..15 of the tile are zeros)
    const float* final_norm;                          // [D]
    const int* pos;                                   // [n_rows] position of every sequence
    const int* kv_row;                                // null: sequence m's cache slice is slice m of kc / vc; else slice kv_row[m] (device-resident: the wide batch's slots change utterances between launches)
    const float* rope_cos; const float* rope_sin;     // [max_pos][hd/2]
    int max_seq, window; float eps;
    unsigned char *XH0, *XH1;                         // all-gathered streams as MFMA A fragments: [96 blocks][hi, lo][64 lanes] x 16 B
    float *SS0, *SS1;                                 // [256 CUs][16] partial sums of squares
    unsigned long long* G;                            // [16][6144] q|k|v granules {value, tag}
    unsigned char* XO;                                // [32 heads][4 blocks][hi, lo][64] x 16 B attention outputs
    float* PW;                                        // [8 planes][3072][16] wo partial products (one plane per XCD group)
    unsigned char* XA;                                // [8 groups][36 blocks][hi, lo][64] x 16 B SwiGLU outputs
    float* P2;                                        // [8 planes][3072][16] w2 partial products
    unsigned *F0, *F1, *FO, *FW, *FA, *F2;            // flag words: F0 / F1 [256 CUs], 16 bytes apart (same-line write-through stores serialise); FO [32 heads][16 sequences], FA [256 CUs] (XCD-local); FW / F2: unused
                                                      // since the partial planes carry their own validity (empty = all ones)
    unsigned long long* XC; unsigned* serial; unsigned* err;
    uint16_t* xf_out; float* ssq_out;                 // the layer stack's output: XF planes of h * final_norm (xf_store4 layout) + [256][16] partial sums of squares
    unsigned long long* tl; int tl_layer;             // timeline stamps [256][32] of layer tl_layer (null: off)
    int flags;                                        // product default 2241 = 1 | 64 | 128 | 2048.  1: one LDS-DMA packet in flight while the CU's COMM wave polls; 64: none issued meanwhile (bounded 3 us); 128: plain stores on XCD-local
                                                      // edges (placement verified per launch); 1024 / 2048: loader depth 2 / 1; 32: no LDS-DMA (diagnostic, wrong results); 16384: fault injection
};
size_t engb_state_bytes();
size_t engb_wo_stream_bytes(int n_layers);            // bytes of the batched engine's wo stream (all 256 CUs)
void engb_state_carve(unsigned char* state, EngBParams* p);      // point p's edge buffers / flags / serial / err into a state block prepared by engb_state_init
hipError_t engb_state_init(unsigned char* state, hipStream_t s);      // zeros + the partial planes marked empty: before the first launch and after a failed one
hipError_t launch_decode_engine_b16(const EngBParams& p, hipStream_t s);
// TWO groups (2 x <= 16 sequences) through ONE launch, phase by phase: group B computes while group A's hand-off resolves.  Same weights / layer table / RoPE tables / flags,
// kv_row set in both, one state block (engb_state_bytes) per group; a timeout is reported in group A's err word.
hipError_t launch_decode_engine_b16x2(const EngBParams& a, const EngBParams& b, hipStream_t s);
hipError_t engb_occupancy(int* blocks_per_cu);
hipError_t engb_prepare_kernels();                    // the dynamic-LDS limits of the three launch forms (once; never inside a stream capture)
hipError_t launch_engb_ssq_fold(const float* ssq256, float* ssq16, hipStream_t s);      // [256][16] -> [16][16] partial sums of squares (fixed order)
int engb_lds_bytes();
int engb_lds_bytes2();                                // ... of the two-group launch

// ---- measurement knobs: VOX_* environment snapshot (taken at vox_ctx_create / vox_debug_reload_knobs); launch paths never call getenv
void knobs_load_once();      // vox_ctx_create: build the snapshot if it does not exist yet (thread-safe)
void knobs_reload();         // vox_debug_reload_knobs (tests): replace the snapshot
// MI355X behaviour found in round 6 (tools/repro/pk_fp32_corun.cpp, profiles/r06_pk_fp32_corun.txt): a packed-FP32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)
// whose op_sel takes the HIGH dword of SRC1 for the LOW lane returns a wrong low lane while ANOTHER wave on the CU executes MFMAs (another stream's GEMM: two contexts on
// one GPU, forked launch chains); so does v_pk_mov_b32 op_sel:[1,0] (low <- src0.high, high <- src1.low).  Straight lanes, src0 / src2 swaps and low-dword broadcasts are
// fine.  hipcc forms these encodings on its own (SLP-vectorised pair arithmetic: rope_kernel's rotation, dense_gemv_kernel's horizontal add, rms_norm_sumk_kernel's sum of
// squares, a swapped 8-byte LDS store in the batched engine), so: kernels where they appeared are compiled without packed FP32 (this attribute) or re-worded, and
// tests/test_abi_cpu.py::test_no_packed_fp32_src1_swap scans every shipped code object for them (tools/kernel_resources.py).  (Only rope_kernel's instance was ever seen to
// fail in the product -- two sessions on one GPU; the others are removed because the stand-alone reproducer fails on their encoding.)
#ifdef VOX_PK_AS_COMPILED      // measurement build `pk_as_compiled` (build.py VARIANTS): the kernels as hipcc packs them -- what the hazard costs to avoid
#define VOX_NO_PK_F32
#else
#define VOX_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#endif
const char* knob_str(const char* name);      // nullptr when unset

// ---- timeline instrumentation (measurement builds only, -DVOX_TIMELINE): every q4_gemv / attn_decode launch gets the next slot and its
// waves stamp s_memrealtime (100 MHz) at 4 points into buf[slot][wave][4]; under graph replay the captured slot is rewritten per replay.
// Returns hipErrorNotSupported in product builds.
hipError_t tl_configure(unsigned long long* buf, int n_slots, int n_waves);   // buf == nullptr: off
int tl_slots_used();
void tl_slot_meta(int slot, int out[4]);    // {0 gemv / 1 attention, epi, N, K} of the launch that took `slot`

}  // namespace vox
