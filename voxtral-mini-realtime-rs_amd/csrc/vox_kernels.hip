// vox_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for the Voxtral hot path.
//
// Design notes (see DESIGN.md for rooflines):
//  * q4_gemv_kernel  : the decode-step operator (reference: gguf/shader.wgsl:41-133, M <= 4).
//      HBM-bound.  One wave owns R consecutive weight rows; lanes split K in 16-byte Q4 chunks
//      (global_load_dwordx4, non-temporal: every weight byte is read exactly once per token).
//      All weight loads of the wave are issued before anything else; the activation vector is
//      staged once per workgroup into LDS (XOR-swizzled so the per-lane ds_read_b128 of
//      128-byte-strided chunks is bank-conflict free) with the RMSNorm(+Ada) prologue fused,
//      and the epilogue fuses bias / residual / SwiGLU / RoPE+KV-cache write / argmax partials.
//  * q4_gemm_kernel  : prefill + encoder operator (reference: gguf/shader_naive.wgsl:31-99, M > 4).
//      MFMA-bound.  v_mfma_f32_16x16x32_bf16: one MFMA K-step == one Q4_0 block, so the integer
//      weights (q-8, exact in bf16) go through the matrix core and the f16 block scale is applied
//      to the 16x16 result in f32; activations are split hi+lo bf16 (2 MFMAs) for f32-class accuracy.
//  * q4_skinny_kernel / q4_skinny_mt_kernel / q4_gemm_big_kernel : <= 16 rows (batched decode step, XF fragment planes in, straight-line
//      in-order load pipeline), 17..48 rows (the 38-token prefill), large M (stacked encoder) -- all on the tile-ordered weight copy.
//  * attn_wo_kernel  : single-stream decode, attention + wo in one launch; the per-head partial products are combined with int64
//      fixed-point atomics (order-independent => deterministic) and summed into the residual by the next GEMV's prologue.
//  * attention, conv, mel, norm, rope, resampler: wave64 shuffle / LDS kernels, f32.
// vmcnt retires loads IN ORDER: every pipeline here requests its operands in consumption order.
#include "vox_kernels.h"

#include <hip/hip_fp16.h>

#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <unordered_map>

namespace vox {

// ---- timeline instrumentation (measurement builds only) ------------------------------------------
static int g_tl_next = -1, g_tl_slots = 0;
#ifdef VOX_TIMELINE
__device__ unsigned long long* g_tl_buf = nullptr;
__device__ int g_tl_waves = 0;
__device__ __forceinline__ void tl_stamp(int slot, int wave_global, int k) {
    if (slot >= 0 && (wave_global & 3) == 0 && (threadIdx.x & 63) == 0 && g_tl_buf && wave_global < g_tl_waves)   // wave 0 of every workgroup only (keeps the overhead low)
        g_tl_buf[((size_t)slot * g_tl_waves + wave_global) * 4 + k] = wall_clock64();
}
#define VOX_TL(slot, wave, k) tl_stamp(slot, wave, k)
hipError_t tl_configure(unsigned long long* buf, int n_slots, int n_waves) {
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_tl_buf), &buf, sizeof buf); if (e != hipSuccess) return e;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_tl_waves), &n_waves, sizeof n_waves); if (e != hipSuccess) return e;
    g_tl_next = buf ? 0 : -1; g_tl_slots = buf ? n_slots : 0;
    return hipSuccess;
}
#else
#define VOX_TL(slot, wave, k) ((void)0)
hipError_t tl_configure(unsigned long long*, int, int) { return hipErrorNotSupported; }
#endif
int tl_slots_used() { return g_tl_next < 0 ? 0 : g_tl_next; }
static int g_tl_meta[4096][4];     // per slot: {0 gemv / 1 attention, epi, N, K}
static int tl_take_slot(int type, int epi, int N, int K) {
    if (g_tl_next < 0 || g_tl_next >= g_tl_slots || g_tl_next >= 4096) return -1;
    g_tl_meta[g_tl_next][0] = type; g_tl_meta[g_tl_next][1] = epi; g_tl_meta[g_tl_next][2] = N; g_tl_meta[g_tl_next][3] = K;
    return g_tl_next++;
}
void tl_slot_meta(int slot, int out[4]) { for (int i = 0; i < 4; i++) out[i] = (slot >= 0 && slot < 4096) ? g_tl_meta[slot][i] : 0; }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// DPP cross-lane (no LDS traffic, unlike __shfl_xor -> ds_bpermute). ctrl: 0xB1 quad_perm[1,0,3,2] (xor 1),
// 0x4E quad_perm[2,3,0,1] (xor 2), 0x141 row_half_mirror, 0x140 row_mirror. All lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float readlane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
// sum over the 16 lanes of each DPP row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v)); v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
// full-wave reductions, result uniform (every lane). Must be called with all 64 lanes active.
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
// sum over aligned groups of 8 lanes (result in every lane of the group)
__device__ __forceinline__ float group8_sum(float v) { v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); return v; }
__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erff(x / 1.41421356237309504880f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt_u4(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
// byte n of a dword -> f32.  Inline asm: written as (w >> 8n) & 0xFF the backend folds the nibble mask into a
// v_bfe + v_cvt_f32_ubyte0 pair (1.5x the VALU work); v_cvt_f32_ubyteN reads byte N directly.
__device__ __forceinline__ float ub0(uint32_t w) { float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(w)); return f; }
__device__ __forceinline__ float ub1(uint32_t w) { float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(w)); return f; }
__device__ __forceinline__ float ub2(uint32_t w) { float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(w)); return f; }
__device__ __forceinline__ float ub3(uint32_t w) { float f; asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(w)); return f; }

// sum_k x[k] * nibble_k for one 16-byte Q4_0 chunk (32 elements). Element i <-> low nibble of byte i,
// element 16+i <-> high nibble of byte i (gguf/tensor.rs:98-109). Offset -8 is applied by the caller
// through the chunk sum: sum x*(q-8) = sum x*q - 8*sum x.
__device__ __forceinline__ float q4_chunk_dot(const uint4 q, const float* __restrict__ x) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t lo = w[i] & 0x0F0F0F0Fu, hi = (w[i] >> 4) & 0x0F0F0F0Fu;
        s0 = fmaf(x[4 * i + 0], ub0(lo), s0);
        s1 = fmaf(x[4 * i + 1], ub1(lo), s1);
        s0 = fmaf(x[4 * i + 2], ub2(lo), s0);
        s1 = fmaf(x[4 * i + 3], ub3(lo), s1);
        s0 = fmaf(x[16 + 4 * i + 0], ub0(hi), s0);
        s1 = fmaf(x[16 + 4 * i + 1], ub1(hi), s1);
        s0 = fmaf(x[16 + 4 * i + 2], ub2(hi), s0);
        s1 = fmaf(x[16 + 4 * i + 3], ub3(hi), s1);
    }
    return s0 + s1;
}

// ------------------------------------------------------------------------------------------------
// Q4 re-pack / dequant (load time + diagnostics)
// ------------------------------------------------------------------------------------------------
// source block b = (row, c) of an [rows][nb] tensor lands at destination row (row*row_mul + row_add): row_mul/row_add
// implement the load-time fusions (wq|wk|wv concatenation, w1/w3 row interleave) without a second pass.
__global__ void q4_repack_kernel(const uint8_t* __restrict__ raw, uint4* __restrict__ qs, uint16_t* __restrict__ sc,
                                 int64_t n_blocks, int nb, int row_mul, int row_add) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint8_t* p = raw + b * 18;
    const int64_t row = b / nb, c = b % nb, dst = (row * row_mul + row_add) * nb + c;
    sc[dst] = (uint16_t)(p[0] | (p[1] << 8));
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        w[i] = (uint32_t)p[2 + 4 * i] | ((uint32_t)p[3 + 4 * i] << 8) | ((uint32_t)p[4 + 4 * i] << 16) | ((uint32_t)p[5 + 4 * i] << 24);
    qs[dst] = make_uint4(w[0], w[1], w[2], w[3]);
}
hipError_t launch_q4_repack(const uint8_t* raw, uint4* qs, uint16_t* sc, int64_t n_blocks, int nb, int row_mul, int row_add, hipStream_t s) {
    if (n_blocks <= 0) return hipSuccess;
    q4_repack_kernel<<<dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, s>>>(raw, qs, sc, n_blocks, nb, row_mul, row_add);
    return hipGetLastError();
}

__global__ void q4_dequant_kernel(Q4W w, float* __restrict__ out) {
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (int64_t)w.N * w.nb) return;
    const uint4 q = w.qs[b];
    const float d = f16_bits_to_f32(w.sc[b]);
    const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
    float* o = out + b * 32;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t by = (ww[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        o[i] = ((float)(by & 0xF) - 8.0f) * d;
        o[i + 16] = ((float)(by >> 4) - 8.0f) * d;
    }
}
hipError_t launch_q4_dequant(Q4W w, float* out, hipStream_t s) {
    int64_t nbk = (int64_t)w.N * w.nb;
    q4_dequant_kernel<<<dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, s>>>(w, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// fused Q4 GEMV
// ------------------------------------------------------------------------------------------------
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int HD, bool NT, bool LATE_V, bool SPEC = false, class Hook = NoHook>
__device__ __forceinline__ float4 attn_decode_core(const float* __restrict__ qh, const float* __restrict__ kb, const float* __restrict__ vb, int kv_row_stride,
                                                   int pos, int window, float* __restrict__ sc, float* __restrict__ red, float4* __restrict__ osum, int tl_slot, int tlw,
                                                   int spec_rows = 0, Hook after_issue = Hook());

__device__ __forceinline__ void st_pair(float* dst, float a, float b) { dst[0] = a; dst[1] = b; }

// LDS image of x: chunk c (32 floats) holds its eight 16-byte pieces at piece index j ^ ((c>>1)&7):
// lanes of one ds_read_b128 service group then hit 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int xs_piece(int c, int j) { return c * 8 + (j ^ ((c >> 1) & 7)); }

// P = passes per row group: the R rows of a group are one contiguous run of R*nb 16-byte chunks, walked 64 chunks
// (one dwordx4 per lane) at a time, so every lane is busy in every pass even when K is not a multiple of 2048
// (K = 3072: R = 2 -> exactly 3 passes).  Requires N % R == 0 and R*nb <= 64*P.
// NWV = waves per workgroup (4, 6 or 12): the total number of waves (and so the row-group -> wave mapping) is the same for every NWV,
// a bigger workgroup only shares one staged copy of the activation vector between more waves (768 / 512 / 256 workgroups re-read it).
template <int P, int R, int PRO, int EPI, int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void q4_gemv_kernel(const GemvParams p) {
    constexpr int NT = 64 * NWV;                                      // threads per workgroup
    constexpr int NX = (8 * P + R * NWV - 1) / (R * NWV);             // float4 activation pieces per thread (K/4 <= 512 P / R pieces)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.w.K, nb = p.w.nb, N = p.w.N;
    float4* xs = reinterpret_cast<float4*>(smem);  // K floats, swizzled
    float* sxs = smem + K;                         // nb chunk sums
    float* red = sxs + nb;                         // 4 * NWV floats scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int y = blockIdx.y;
    const float* __restrict__ xg = p.x + (size_t)y * p.x_stride;
    const int npieces = K >> 2;
    // persistent waves: wave w handles row groups g = w, w + n_waves, ... with the NEXT group's weight loads issued
    // before the current group is consumed, so HBM stays busy while the VALU works.
    const int rnb = R * nb, n_groups = N / R, n_waves = gridDim.x * NWV;
    const int bx = (int)blockIdx.x;
    int g = bx * NWV + wave;
    VOX_TL(p.tl_slot, blockIdx.x * NWV + wave, 0);
    if (p.zero_acc) {      // uniform; a handful of stores per workgroup
        const int per = (p.zero_n + (int)gridDim.x - 1) / (int)gridDim.x, i_end = min(((int)blockIdx.x + 1) * per, p.zero_n);
        for (int i = blockIdx.x * per + tid; i < i_end; i += NT) p.zero_acc[i] = 0;      // (a loop: a small grid has more entries per workgroup than threads)
    }

    // Every global load below is UNCONDITIONAL (indices are clamped, never predicated): a "cond ? load : 0"
    // makes hipcc branch around the load and drain vmcnt(0) per element, which serialises HBM round trips.
    // (1) activation pieces (+ norm weights) first -- VMEM returns in order, so they land first ...
    constexpr bool HAS_MUL = PRO == PRO_RMS_MUL || PRO == PRO_RMS_MUL_SUM, SUMP = PRO == PRO_RMS_MUL_SUM;
    constexpr int NPART = 2;                                          // PRO_RMS_MUL_SUM: the 4 int64 accumulators of a piece = two 16-byte loads
    float4 xp[NX], gp[PRO != PRO_NONE ? NX : 1], mp[HAS_MUL ? NX : 1], pp[SUMP ? NX : 1][SUMP ? NPART : 1];
#define VOX_XLOAD                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NX; i++) {                                                     \
        const int pc = min(tid + NT * i, npieces - 1);                                                   \
        xp[i] = reinterpret_cast<const float4*>(xg)[pc];                                                 \
        if (SUMP) { pp[i][0] = reinterpret_cast<const float4*>(p.xacc)[2 * pc]; pp[i][1] = reinterpret_cast<const float4*>(p.xacc)[2 * pc + 1]; } \
        if (PRO != PRO_NONE) gp[i] = reinterpret_cast<const float4*>(p.gamma)[pc];                       \
        if (HAS_MUL) mp[i] = reinterpret_cast<const float4*>(p.mul)[pc];                                 \
    }
#if defined(VOX_ABL_NOX)          /* measurement build: no activation loads at all (results wrong) */
#pragma unroll
    for (int i = 0; i < NX; i++) { xp[i] = make_float4(1.f, 2.f, 3.f, 4.f); gp[PRO != PRO_NONE ? i : 0] = xp[i]; mp[HAS_MUL ? i : 0] = xp[i]; if (SUMP) for (int u = 0; u < NPART; u++) pp[i][u] = make_float4(0.f, 0.f, 0.f, 0.f); }
#elif !defined(VOX_ABL_WFIRST)
    VOX_XLOAD
#endif
    // per-pass lane constants: chunk-in-group -> (row in group, chunk in row)
    int vcl[P], cp[P], rp[P]; bool okp[P];
#pragma unroll
    for (int q_ = 0; q_ < P; q_++) {
        const int v = lane + 64 * q_;
        okp[q_] = v < rnb; vcl[q_] = min(v, rnb - 1);
        rp[q_] = R == 1 ? 0 : vcl[q_] / nb; cp[q_] = vcl[q_] - rp[q_] * nb;
    }
    // (2) ... then the first row group's weights.
    uint4 qa[P], qb[P];
    uint16_t da[P], db[P];
#define VOX_WLOAD(Q_, D_, G_)                                                                           \
    _Pragma("unroll") for (int q_ = 0; q_ < P; q_++) {                                                   \
        const size_t idx = (size_t)(G_) * rnb + vcl[q_];                                                 \
        Q_[q_] = ld_nt_u4(p.w.qs + idx);                                                                 \
        D_[q_] = VOX_SCLOAD(idx);                                                                        \
    }
#ifdef VOX_ABL_NOSCALE            /* measurement build: no block-scale loads (results wrong) */
#define VOX_SCLOAD(I_) ((uint16_t)0x3c00)
#else
#define VOX_SCLOAD(I_) __builtin_nontemporal_load(p.w.sc + (I_))
#endif
    // No weight request before the activation vector is staged: the vector is a cross-XCD read that otherwise queues behind the weight bursts of
    // the workgroups that started first (3 x 3 back-to-back pairs, tools/gemv_ablate.py: q|k|v 6.75 -> 6.20 us, w2 7.9 -> 7.6, wo 4.75 -> 4.65;
    // w1|w3 and lm_head neutral).
#ifdef VOX_ABL_XFIRST_RESID_ONLY      /* measurement build: the earlier setting (only wo / w2 wait for the vector) */
    constexpr bool XFIRST = PRO == PRO_NONE && EPI == EPI_RESID;
#elif defined(VOX_ABL_XFIRST_NO_SWIGLU)  /* measurement build: w1|w3 keeps the weights-with-vector order */
    constexpr bool XFIRST = EPI != EPI_SWIGLU;
#else
    constexpr bool XFIRST = true;
#endif
    if (!XFIRST) { VOX_WLOAD(qa, da, min(g, n_groups - 1)) }
#if defined(VOX_ABL_WFIRST) && !defined(VOX_ABL_NOX)       /* measurement build: weights issued BEFORE the activation loads */
    VOX_XLOAD
#endif
#undef VOX_XLOAD

    // (3) prologue on the activation vector (RMSNorm (+Ada multiplier) fused), staged to LDS.  The row scale 1/rms is a scalar, so it
    // commutes with the dot products: x * gamma is staged UNnormalised right away and every row result is multiplied by rstd in
    // the epilogue -- the sum-of-squares reduction no longer sits (with its own barrier) in front of the staging.
    if (SUMP) {      // x = resid + acc * 2^-32 (attn_wo_kernel's fixed-point sums); piece pc of the result is written back by workgroup pc % gridDim.x
#pragma unroll
        for (int i = 0; i < NX; i++) {
            float4 a = xp[i];
            const float4 lo = pp[i][0], hi = pp[i][1];
            // int64 with 32 fractional bits -> f32 with ONE rounding (the integer and fraction halves converted separately cancel for small negative sums);
            // the power-of-two scale afterwards is exact
            const auto fx = [](float l, float h_) { return __ll2float_rn((long long)(((unsigned long long)__float_as_uint(h_) << 32) | (unsigned long long)__float_as_uint(l))) * (1.0f / 4294967296.0f); };
            a.x += fx(lo.x, lo.y); a.y += fx(lo.z, lo.w); a.z += fx(hi.x, hi.y); a.w += fx(hi.z, hi.w);
            xp[i] = a;
            const int pc = tid + NT * i;
            if (pc < npieces && pc % (int)gridDim.x == (int)blockIdx.x) reinterpret_cast<float4*>(p.x_out)[pc] = a;
        }
    }
    if (PRO != PRO_NONE) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++)
            if (tid + NT * i < npieces) ss += xp[i].x * xp[i].x + xp[i].y * xp[i].y + xp[i].z * xp[i].z + xp[i].w * xp[i].w;
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
    }
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int pc = tid + NT * i;            // pieces come in whole groups of 8 lanes (K % 32 == 0)
        float4 v = xp[i];
        if (PRO != PRO_NONE) {
            const float4 gm = gp[i];
            v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
            if (HAS_MUL) {
                const float4 m = mp[i];
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
            }
        }
        const float s8 = group8_sum((v.x + v.y) + (v.z + v.w));     // all lanes active (DPP)
        if (pc < npieces) {
            const int c = pc >> 3, j = pc & 7;
            if (j == 0) sxs[c] = s8;
            xs[xs_piece(c, j)] = v;
        }
    }
    __syncthreads();
    if (XFIRST) { VOX_WLOAD(qa, da, min(g, n_groups - 1)) }
    VOX_TL(p.tl_slot, blockIdx.x * NWV + wave, 1);
    // burn RmsNorm divides by sqrt(mean(x^2) + eps); we multiply by the reciprocal
    float ssq = 0.f;
    if (PRO != PRO_NONE) {      // fixed order: deterministic
#pragma unroll
        for (int w = 0; w < NWV; w += 4) ssq += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
    }
    const float rstd = PRO != PRO_NONE ? 1.0f / sqrtf(ssq / (float)K + p.eps) : 1.0f;

    float best = -INFINITY; int best_i = 0x7fffffff;   // EPI_ARGMAX running (max, first index) of this wave
    constexpr bool ROPE = EPI == EPI_ROPE_KV;
    const int pos = ROPE ? (p.pos_ptr ? *p.pos_ptr : 0) + p.pos_off : 0;

    // (4)+(5) consume one row group from registers (Q_, D_) and run its epilogue
#define VOX_GROUP(Q_, D_, G_)                                                                                          \
    {                                                                                                                  \
        const int row0 = (G_) * R;                                                                                     \
        float acc[R];                                                                                                  \
        _Pragma("unroll") for (int r = 0; r < R; r++) acc[r] = 0.f;                                                    \
        _Pragma("unroll") for (int q_ = 0; q_ < P; q_++) {                                                             \
            if (okp[q_]) {                                                                                             \
                float xv[32];                                                                                          \
                _Pragma("unroll") for (int j = 0; j < 8; j++) {                                                        \
                    const float4 v = xs[xs_piece(cp[q_], j)];                                                          \
                    xv[4 * j + 0] = v.x; xv[4 * j + 1] = v.y; xv[4 * j + 2] = v.z; xv[4 * j + 3] = v.w;                \
                }                                                                                                      \
                const float val = f16_bits_to_f32(D_[q_]) * (VOX_DOT(Q_[q_], xv) - 8.0f * sxs[cp[q_]]);                \
                _Pragma("unroll") for (int r = 0; r < R; r++) acc[r] += (R == 1 || rp[q_] == r) ? val : 0.f;           \
            }                                                                                                          \
        }                                                                                                              \
        VOX_REDUCE                                                                                                     \
        if (PRO != PRO_NONE) { _Pragma("unroll") for (int r = 0; r < R; r++) acc[r] *= rstd; }                          \
        if (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_GELU) {                                                 \
            _Pragma("unroll") for (int r = 0; r < R; r++) {                                                            \
                const int n = row0 + r;                                                                                \
                if (lane == r) {                                                                                       \
                    float v = acc[r];                                                                                  \
                    if (p.bias) v += p.bias[n];                                                                        \
                    if (EPI == EPI_RESID) v = v + p.resid[(size_t)y * p.resid_stride + n];                             \
                    if (EPI == EPI_GELU) v = gelu_f(v);                                                                \
                    p.out[(size_t)y * p.out_stride + n] = v;                                                           \
                }                                                                                                      \
            }                                                                                                          \
        } else if (EPI == EPI_SWIGLU) { /* rows interleaved at load: 2i = w1 row i (gate), 2i+1 = w3 row i (up) */    \
            _Pragma("unroll") for (int r = 0; r + 1 < R; r += 2) {                                                     \
                const int n = row0 + r;                                                                                \
                if (lane == (r >> 1)) p.out[(size_t)y * p.out_stride + (n >> 1)] = silu_f(acc[r]) * acc[r + 1];        \
            }                                                                                                          \
        } else if (ROPE) { /* [wq|wk|wv]: RoPE pairs (rope.rs:99-141), k/v into the cache slot */                      \
            const int hd = p.hd, half = hd >> 1;                                                                       \
            _Pragma("unroll") for (int r = 0; r + 1 < R; r += 2) {                                                     \
                const int n = row0 + r;                                                                                \
                if (lane == (r >> 1)) {                                                                                \
                    const float a = acc[r], b = acc[r + 1];                                                            \
                    if (n < p.n_q + p.n_k) {                                                                           \
                        const int dd = n % hd;                                                                         \
                        const float c = p.rope_cos[(size_t)pos * half + (dd >> 1)], sn = p.rope_sin[(size_t)pos * half + (dd >> 1)]; \
                        const float ra = a * c - b * sn, rb = a * sn + b * c;                                          \
                        if (n < p.n_q) st_pair(p.out + n, ra, rb);                                                 \
                        else {                                                                                         \
                            const int kn = n - p.n_q, kh = kn / hd;                                                    \
                            st_pair(p.kcache + (size_t)kh * p.cache_head_stride + (size_t)pos * hd + dd, ra, rb);  \
                        }                                                                                              \
                    } else {                                                                                           \
                        const int vn = n - p.n_q - p.n_k, vh = vn / hd, dd = vn % hd;                                  \
                        st_pair(p.vcache + (size_t)vh * p.cache_head_stride + (size_t)pos * hd + dd, a, b);        \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
        } else if (EPI == EPI_ARGMAX) {                                                                                \
            _Pragma("unroll") for (int r = 0; r < R; r++) {                                                            \
                const int n = row0 + r;                                                                                \
                if (p.out && lane == r) p.out[(size_t)y * p.out_stride + n] = acc[r];                                  \
                if (acc[r] > best || (acc[r] == best && n < best_i)) { best = acc[r]; best_i = n; }                    \
            }                                                                                                          \
        }                                                                                                              \
    }
#ifdef VOX_ABL_NOCONSUME
#define VOX_DOT(Q_, X_) (__uint_as_float(((Q_).x ^ (Q_).y ^ (Q_).z ^ (Q_).w) & 0x3fffffffu) + (X_)[0])
#else
#define VOX_DOT(Q_, X_) q4_chunk_dot(Q_, X_)
#endif
#ifdef VOX_ABL_NOREDUCE
#define VOX_REDUCE _Pragma("unroll") for (int r = 0; r < R; r++) acc[r] = readlane_f(acc[r], 0);
#else
#define VOX_REDUCE _Pragma("unroll") for (int r = 0; r < R; r++) acc[r] = wave_sum(acc[r]);
#endif

    // software-pipelined, unrolled by two so the double buffer needs no register copies
    while (g < n_groups) {
        VOX_WLOAD(qb, db, min(g + n_waves, n_groups - 1))     // prefetch (clamped: harmless re-read at the tail)
        VOX_GROUP(qa, da, g)
        VOX_TL(p.tl_slot, blockIdx.x * NWV + wave, 2);      // (rewritten every other group: the last odd group's completion)
        g += n_waves;
        if (g >= n_groups) break;
        VOX_WLOAD(qa, da, min(g + n_waves, n_groups - 1))
        VOX_GROUP(qb, db, g)
        g += n_waves;
    }
#undef VOX_WLOAD
#undef VOX_SCLOAD
#undef VOX_GROUP
#undef VOX_DOT
#undef VOX_REDUCE
    VOX_TL(p.tl_slot, blockIdx.x * NWV + wave, 3);
    if (EPI == EPI_ARGMAX) {
        if (lane == 0) { red[NWV + wave] = best; reinterpret_cast<int*>(red)[2 * NWV + wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = red[NWV]; int bidx = reinterpret_cast<int*>(red)[2 * NWV];
            for (int wv = 1; wv < NWV; wv++) {
                const float v = red[NWV + wv]; const int ii = reinterpret_cast<int*>(red)[2 * NWV + wv];
                if (v > bv || (v == bv && ii < bidx)) { bv = v; bidx = ii; }
            }
            p.part_val[(size_t)y * gridDim.x + blockIdx.x] = bv;
            p.part_idx[(size_t)y * gridDim.x + blockIdx.x] = bidx;
        }
    }
}

// raise the dynamic-LDS limit once per kernel (not on every launch: launches may be inside a graph capture)
template <class Kern>
static hipError_t ensure_dyn_lds(Kern kern, size_t lds, DevOnce* done) {
    if (lds <= 48 * 1024) return hipSuccess;
    const int dev = vox_current_device();
    if (done->done(dev)) return hipSuccess;
    // (the limit covers static + dynamic LDS: a kernel that also declares __shared__ arrays may only ask for the remainder)
    hipFuncAttributes fa; size_t stat = 0;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)) == hipSuccess) stat = fa.sharedSizeBytes; else (void)hipGetLastError();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - std::min<size_t>(stat, 96 * 1024)));
    if (e == hipSuccess) done->set(dev);
    return e;
}

static hipError_t launch_dense_gemv(const GemvParams& p, int ny, int pro, int epi, hipStream_t s);
// Measurement knobs (VOX_*): the environment is read ONCE -- at vox_ctx_create, or by vox_debug_reload_knobs() for the tests that flip a knob at run time -- into a
// table; no launch path calls getenv.  Lookups are read-only between reloads (a reload while another thread launches is the caller's race, as with setenv itself).
// The table is an IMMUTABLE snapshot: it is built once (first vox_ctx_create / first knob_str) and afterwards only ever REPLACED as a whole by vox_debug_reload_knobs (tests;
// documented as not thread-safe against running launches).  Launch paths hold c_str() pointers into it, so a second vox_ctx_create on another thread -- one context per
// GPU in one process -- must not clear and rebuild it (round-3 advisor finding: data race + use-after-free).
static std::unordered_map<std::string, std::string>* g_knob_snap = nullptr;
static std::once_flag g_knob_once;
static std::unordered_map<std::string, std::string>* knobs_build() {
    auto* t = new std::unordered_map<std::string, std::string>();
    for (char** e = environ; e && *e; ++e)
        if (!strncmp(*e, "VOX_", 4)) { const char* eq = strchr(*e, '='); if (eq) t->emplace(std::string(*e, eq - *e), std::string(eq + 1)); }
    return t;
}
void knobs_load_once() { std::call_once(g_knob_once, [] { g_knob_snap = knobs_build(); }); }
void knobs_reload() {      // tests only: the old snapshot is leaked on purpose (pointers into it may still be held)
    knobs_load_once();
    g_knob_snap = knobs_build();
}
const char* knob_str(const char* name) {
    knobs_load_once();
    const auto* t = g_knob_snap; auto it = t->find(name);
    return it == t->end() ? nullptr : it->second.c_str();
}
static int env_int(const char* name) { const char* v = knob_str(name); return v ? atoi(v) : 0; }

// instantiated (R, P) pairs; P = ceil(R * nb / 64)
static bool gemv_has(int R, int P) {
    switch (R) {
    case 1: return P == 1 || P == 2 || P == 3 || P == 5;
    case 2: return P == 1 || P == 2 || P == 3 || P == 4 || P == 5 || P == 9;
    case 4: return P == 1 || P == 2 || P == 6;
    default: return false;
    }
}
static inline int passes_for(int K, int R) { return (R * (K / 32) + 63) / 64; }

// rows per wave. Prefers the R that fills every pass exactly (R*nb % 64 == 0). Tuning knobs (measurement only):
// VOX_GEMV_R / VOX_GEMV_R_PAIR / VOX_GEMV_R_ARGMAX override the choice when the (R, P) pair is instantiated.
int q4_gemv_default_R(int N, int K, int epi) {
    const bool pair = (epi == EPI_SWIGLU || epi == EPI_ROPE_KV);
    const int e = env_int(epi == EPI_ARGMAX ? "VOX_GEMV_R_ARGMAX" : pair ? "VOX_GEMV_R_PAIR" : "VOX_GEMV_R");
    if (e && N % e == 0 && gemv_has(e, passes_for(K, e)) && (!pair || e % 2 == 0)) return e;
    const int nb = K / 32;
    const int order_exact[3] = {pair ? 2 : 1, pair ? 4 : 2, 4};
    for (int i = 0; i < 3; i++) {   // exact fit, unless it needs so many passes that the register file halves occupancy
        const int R = order_exact[i];
        if (N % R == 0 && (R * nb) % 64 == 0 && passes_for(K, R) <= 6 && gemv_has(R, passes_for(K, R))) return R;
    }
    if (!pair && gemv_has(1, passes_for(K, 1))) return 1;
    const int order_any[3] = {2, pair ? 4 : 1, 4};
    for (int i = 0; i < 3; i++) { const int R = order_any[i]; if ((!pair || R % 2 == 0) && N % R == 0 && gemv_has(R, passes_for(K, R))) return R; }
    return 0;   // no instantiation covers this K
}

// number of workgroups for a GEMV over N rows with R rows per wave: every wave gets an equal whole number of row
// groups where possible, at most ~3 workgroups per CU stay resident and stream (persistent waves).
int dense_gemv_grid(int N);
// waves per workgroup of the decode GEMV: 4 (768 workgroups), 6 (512) or 12 (256, one per CU) -- same total waves, fewer copies of the
// staged activation vector.  VOX_GEMV_NWV overrides (measurement knob).
// Round-2 measurements (profiles/r02_decode_knobs.txt): 12-wave workgroups (one copy of x per CU instead of three) win where the staged
// vector is long or the kernel is long enough to amortise the bigger barrier -- wo (K = 4096) 4.93 -> 4.68 us, w2 (K = 9216) 8.31 -> 7.89 us,
// w1|w3 9.58 -> 9.25 us -- and lose on q|k|v (RoPE / cache epilogue; 6.08 -> 6.16) and lm_head (38.9 -> 41.1).
int q4_gemv_nwv(int K, int epi) {
    const int e = env_int("VOX_GEMV_NWV");
    if (e == 4 || e == 6 || e == 12) return e;
    if (epi == EPI_ARGMAX || epi == EPI_ROPE_KV) return 4;
    return (K >= 4096 || epi == EPI_SWIGLU) ? 12 : 4;
}
static int q4_gemv_grid_w(int N, int R, int nwv) {
    const int n_groups = N / R;
    int target = env_int("VOX_GEMV_WGS"); if (target <= 0) target = 768;
    target = target * 4 / nwv;                      // the same number of waves for every workgroup size
    int wgs = (n_groups + nwv - 1) / nwv;
    if (wgs > target) {
        const int iters = (n_groups + nwv * target - 1) / (nwv * target);
        wgs = (n_groups + nwv * iters - 1) / (nwv * iters);
    }
    return wgs < 1 ? 1 : wgs;
}
int q4_gemv_grid(int N, int R) { return q4_gemv_grid_w(N, R, 4); }

static bool gemv_fat_shape(int K, int R) { const int P = passes_for(K, R); return (R == 2 && P == 3) || (R == 1 && (P == 2 || P == 5)); }
int q4_gemv_grid_k(int N, int K, int R, int epi) { return q4_gemv_grid_w(N, R, gemv_fat_shape(K, R) ? q4_gemv_nwv(K, epi) : 4); }

// (Round 2 also built a fused q|k|v GEMV + attention launch -- write-through q / k / v stores, per-head arrival counters, the last-arriving workgroup runs the head.  It was
// bit-identical to two launches and SLOWER (22.5 us per layer against 6.8 + 1.9 + 4.6, profiles/r02_fused_attention_timeline.txt) and spilled 93 VGPRs; removed in round 5.)
template <int P, int R, int PRO, int EPI, int NWV>
static hipError_t gemv_launch_w(const GemvParams& p, int ny, hipStream_t s) {
    dim3 grid(q4_gemv_grid_w(p.w.N, R, NWV), ny);
    size_t lds = (size_t)(p.w.K + p.w.nb + 4 * NWV) * sizeof(float);
    auto kern = q4_gemv_kernel<P, R, PRO, EPI, NWV>;
    static DevOnce attr_done;
    hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
    if (e != hipSuccess) return e;
    kern<<<grid, dim3(64 * NWV), lds, s>>>(p);
    return hipGetLastError();
}
// fat workgroups are instantiated for the decode-step shapes only (R*nb % 64 == 0 passes: (R, P) = (2,3) (1,2) (1,5) (1,3) (2,2))
template <int P, int R, int PRO, int EPI>
static hipError_t gemv_launch_t(const GemvParams& p, int ny, hipStream_t s) {
    constexpr bool fat = (R == 2 && P == 3) || (R == 1 && (P == 2 || P == 5));      // == gemv_fat_shape(K, R)
    if (fat) {
        const int nwv = q4_gemv_nwv(p.w.K, EPI);
        if (nwv == 12) return gemv_launch_w<P, R, PRO, EPI, fat ? 12 : 4>(p, ny, s);
        if (nwv == 6) return gemv_launch_w<P, R, PRO, EPI, fat ? 6 : 4>(p, ny, s);
    }
    return gemv_launch_w<P, R, PRO, EPI, 4>(p, ny, s);
}

template <int P, int R>
static hipError_t gemv_dispatch_pe(const GemvParams& p, int ny, int pro, int epi, hipStream_t s) {
#define VOX_CASE(P_, E_) if (pro == P_ && epi == E_) return gemv_launch_t<P, R, P_, E_>(p, ny, s)
    VOX_CASE(PRO_NONE, EPI_STORE); VOX_CASE(PRO_NONE, EPI_RESID); VOX_CASE(PRO_NONE, EPI_GELU);
    VOX_CASE(PRO_RMS, EPI_STORE); VOX_CASE(PRO_RMS, EPI_ARGMAX);
    if ((pro == PRO_RMS_MUL || pro == PRO_RMS_MUL_SUM) && p.mul == nullptr) return hipErrorInvalidValue;
    if (pro == PRO_RMS_MUL_SUM && (!p.xacc || !p.x_out || ny != 1)) return hipErrorInvalidValue;
    if (R >= 2) {
        constexpr int R2 = R >= 2 ? R : 2;
        if (pro == PRO_RMS && epi == EPI_SWIGLU) return gemv_launch_t<P, R2, PRO_RMS, EPI_SWIGLU>(p, ny, s);
        if (pro == PRO_RMS_MUL && epi == EPI_SWIGLU) return gemv_launch_t<P, R2, PRO_RMS_MUL, EPI_SWIGLU>(p, ny, s);
        if (pro == PRO_RMS_MUL_SUM && epi == EPI_SWIGLU && P == 3 && R == 2) return gemv_launch_t<3, 2, PRO_RMS_MUL_SUM, EPI_SWIGLU>(p, ny, s);
        if (pro == PRO_NONE && epi == EPI_SWIGLU) return gemv_launch_t<P, R2, PRO_NONE, EPI_SWIGLU>(p, ny, s);
        if (pro == PRO_RMS && epi == EPI_ROPE_KV) return gemv_launch_t<P, R2, PRO_RMS, EPI_ROPE_KV>(p, ny, s);
    }
#undef VOX_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_q4_gemv(const GemvParams& p_in, int ny, int pro, int epi, int R, hipStream_t s) {
    GemvParams p = p_in; p.tl_slot = tl_take_slot(0, epi, p.w.N, p.w.K);
    if (p.w.fmt == WFMT_BF16 || p.w.fmt == WFMT_F32) return launch_dense_gemv(p, ny, pro, epi, s);
    if (p.w.K % 32 || p.w.K <= 0 || p.w.N <= 0 || R <= 0 || p.w.N % R) return hipErrorInvalidValue;
    const int P = passes_for(p.w.K, R);
#define VOX_RP(R_, P_) if (R == R_ && P == P_) return gemv_dispatch_pe<P_, R_>(p, ny, pro, epi, s)
    VOX_RP(1, 1); VOX_RP(1, 2); VOX_RP(1, 3); VOX_RP(1, 5);
    VOX_RP(2, 1); VOX_RP(2, 2); VOX_RP(2, 3); VOX_RP(2, 4); VOX_RP(2, 5); VOX_RP(2, 9);
    VOX_RP(4, 1); VOX_RP(4, 2); VOX_RP(4, 6);
#undef VOX_RP
    return hipErrorInvalidValue;   // K beyond the instantiated range (this model family: K <= 9216)
}

const char* q4_gemv_kernel_name(int K, int pro, int epi, int R) {
    static thread_local char buf[96];
    snprintf(buf, sizeof buf, "q4_gemv_kernel<P=%d,R=%d,PRO=%d,EPI=%d,NWV=%d>", passes_for(K, R), R, pro, epi, gemv_fat_shape(K, R) ? q4_gemv_nwv(K, epi) : 4);
    return buf;
}

// ------------------------------------------------------------------------------------------------
// dense bf16 GEMV (the f32 SafeTensors path, models/layers/*.rs with burn::nn::Linear; decode rows <= 4).
// Weights are the checkpoint's BF16 values (exact), activations and accumulation f32.  HBM-bound like the Q4 GEMV
// (6.9 GB per token); low register use -> 8 waves/SIMD hide the latency, no explicit pipelining needed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8_bf16(const uint4 w, const float4 xa, const float4 xb) {
    float s0 = __uint_as_float(w.x << 16) * xa.x, s1 = __uint_as_float(w.x & 0xFFFF0000u) * xa.y;
    s0 = fmaf(__uint_as_float(w.y << 16), xa.z, s0); s1 = fmaf(__uint_as_float(w.y & 0xFFFF0000u), xa.w, s1);
    s0 = fmaf(__uint_as_float(w.z << 16), xb.x, s0); s1 = fmaf(__uint_as_float(w.z & 0xFFFF0000u), xb.y, s1);
    s0 = fmaf(__uint_as_float(w.w << 16), xb.z, s0); s1 = fmaf(__uint_as_float(w.w & 0xFFFF0000u), xb.w, s1);
    return s0 + s1;
}

__device__ __forceinline__ float dot8_f32(const uint4 we, const uint4 wo, const float4 xa, const float4 xb) {   // 8 consecutive f32 weights . x
    float s0 = __uint_as_float(we.x) * xa.x, s1 = __uint_as_float(we.y) * xa.y;
    s0 = fmaf(__uint_as_float(we.z), xa.z, s0); s1 = fmaf(__uint_as_float(we.w), xa.w, s1);
    s0 = fmaf(__uint_as_float(wo.x), xb.x, s0); s1 = fmaf(__uint_as_float(wo.y), xb.y, s1);
    s0 = fmaf(__uint_as_float(wo.z), xb.z, s0); s1 = fmaf(__uint_as_float(wo.w), xb.w, s1);
    return s0 + s1;
}
// F32W: the weights are the exact f32 plane (WFMT_F32: 32 bytes per 8 weights instead of 16)
template <int PRO, int EPI, int F32W = 0>
__global__ __launch_bounds__(256) VOX_NO_PK_F32 void dense_gemv_kernel(const GemvParams p) {      // (VOX_NO_PK_F32: hipcc's horizontal add was a src1-swapped v_pk_add_f32, vox_kernels.h)
    constexpr int NX = 10, R = 2;                  // K <= 10240; one wave = one row pair
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = p.w.K, N = p.w.N, nc = K >> 3;   // 16-byte chunks (8 bf16) per row
    float4* xe = reinterpret_cast<float4*>(smem);  // x[8c .. 8c+3]
    float4* xo = xe + nc;                          // x[8c+4 .. 8c+7]   (split so lane-consecutive reads are conflict-free)
    float* red = smem + K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, y = blockIdx.y;
    const float* __restrict__ xg = p.x + (size_t)y * p.x_stride;
    const int npieces = K >> 2;
    float4 xp[NX];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int pc = tid + 256 * i;
        xp[i] = reinterpret_cast<const float4*>(xg)[min(pc, npieces - 1)];
        if (PRO != PRO_NONE && pc < npieces) ss += (xp[i].x * xp[i].x + xp[i].y * xp[i].y) + (xp[i].z * xp[i].z + xp[i].w * xp[i].w);
    }
    float rms = 1.0f;
    if (PRO != PRO_NONE) {
        ss = wave_sum(ss);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        rms = 1.0f / sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + p.eps);
    }
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const int pc = tid + 256 * i;
        if (pc < npieces) {
            float4 v = xp[i];
            if (PRO != PRO_NONE) {
                const float4 gm = reinterpret_cast<const float4*>(p.gamma)[pc];
                v.x = (v.x * rms) * gm.x; v.y = (v.y * rms) * gm.y; v.z = (v.z * rms) * gm.z; v.w = (v.w * rms) * gm.w;
                if (PRO == PRO_RMS_MUL) {
                    const float4 m = reinterpret_cast<const float4*>(p.mul)[pc];
                    v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                }
            }
            if (pc & 1) xo[pc >> 1] = v; else xe[pc >> 1] = v;
        }
    }
    __syncthreads();
    float best = -INFINITY; int best_i = 0x7fffffff;
    const int pos = (EPI == EPI_ROPE_KV) ? (p.pos_ptr ? *p.pos_ptr : 0) + p.pos_off : 0;
    const int n_groups = N / R, n_waves = gridDim.x * 4;
    for (int g = blockIdx.x * 4 + wave; g < n_groups; g += n_waves) {
        const int row0 = g * R;
        const uint4* w0 = (F32W ? p.w.qt : p.w.qs) + (size_t)row0 * nc * (F32W ? 2 : 1);
        const uint4* w1 = w0 + nc * (F32W ? 2 : 1);
        float acc[R] = {0.f, 0.f};
        for (int c0 = 0; c0 < nc; c0 += 128) {
            const int ca = c0 + lane, cb = ca + 64, cca = min(ca, nc - 1), ccb = min(cb, nc - 1);
            const float4 xea = xe[cca], xoa = xo[cca], xeb = xe[ccb], xob = xo[ccb];
            const float ma = ca < nc ? 1.0f : 0.0f, mb = cb < nc ? 1.0f : 0.0f;
            if (F32W) {
                const uint4 a0e = ld_nt_u4(w0 + 2 * cca), a0o = ld_nt_u4(w0 + 2 * cca + 1), a1e = ld_nt_u4(w0 + 2 * ccb), a1o = ld_nt_u4(w0 + 2 * ccb + 1);
                const uint4 b0e = ld_nt_u4(w1 + 2 * cca), b0o = ld_nt_u4(w1 + 2 * cca + 1), b1e = ld_nt_u4(w1 + 2 * ccb), b1o = ld_nt_u4(w1 + 2 * ccb + 1);
                acc[0] += ma * dot8_f32(a0e, a0o, xea, xoa) + mb * dot8_f32(a1e, a1o, xeb, xob);
                acc[1] += ma * dot8_f32(b0e, b0o, xea, xoa) + mb * dot8_f32(b1e, b1o, xeb, xob);
            } else {
                const uint4 a0 = ld_nt_u4(w0 + cca), a1 = ld_nt_u4(w0 + ccb), b0 = ld_nt_u4(w1 + cca), b1 = ld_nt_u4(w1 + ccb);
                acc[0] += ma * dot8_bf16(a0, xea, xoa) + mb * dot8_bf16(a1, xeb, xob);
                acc[1] += ma * dot8_bf16(b0, xea, xoa) + mb * dot8_bf16(b1, xeb, xob);
            }
        }
        acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]);
        if (EPI == EPI_STORE || EPI == EPI_RESID || EPI == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int n = row0 + r;
                if (lane == r) {
                    float v = acc[r];
                    if (p.bias) v += p.bias[n];
                    if (EPI == EPI_RESID) v = v + p.resid[(size_t)y * p.resid_stride + n];
                    if (EPI == EPI_GELU) v = gelu_f(v);
                    p.out[(size_t)y * p.out_stride + n] = v;
                }
            }
        } else if (EPI == EPI_SWIGLU) {
            if (lane == 0) p.out[(size_t)y * p.out_stride + (row0 >> 1)] = silu_f(acc[0]) * acc[1];
        } else if (EPI == EPI_ROPE_KV) {
            const int hd = p.hd, half = hd >> 1, n = row0;
            if (lane == 0) {
                const float a = acc[0], b = acc[1];
                if (n < p.n_q + p.n_k) {
                    const int dd = n % hd;
                    const float c = p.rope_cos[(size_t)pos * half + (dd >> 1)], sn = p.rope_sin[(size_t)pos * half + (dd >> 1)];
                    const float ra = a * c - b * sn, rb = a * sn + b * c;
                    if (n < p.n_q) { p.out[n] = ra; p.out[n + 1] = rb; }
                    else {
                        const int kn = n - p.n_q, kh = kn / hd;
                        float* dst = p.kcache + (size_t)kh * p.cache_head_stride + (size_t)pos * hd + dd;
                        dst[0] = ra; dst[1] = rb;
                    }
                } else {
                    const int vn = n - p.n_q - p.n_k, vh = vn / hd, dd = vn % hd;
                    float* dst = p.vcache + (size_t)vh * p.cache_head_stride + (size_t)pos * hd + dd;
                    dst[0] = a; dst[1] = b;
                }
            }
        } else if (EPI == EPI_ARGMAX) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int n = row0 + r;
                if (p.out && lane == r) p.out[(size_t)y * p.out_stride + n] = acc[r];
                if (acc[r] > best || (acc[r] == best && n < best_i)) { best = acc[r]; best_i = n; }
            }
        }
    }
    if (EPI == EPI_ARGMAX) {
        if (lane == 0) { red[4 + wave] = best; reinterpret_cast<int*>(red)[8 + wave] = best_i; }
        __syncthreads();
        if (tid == 0) {
            float bv = red[4]; int bidx = reinterpret_cast<int*>(red)[8];
            for (int wv = 1; wv < 4; wv++) {
                const float v = red[4 + wv]; const int ii = reinterpret_cast<int*>(red)[8 + wv];
                if (v > bv || (v == bv && ii < bidx)) { bv = v; bidx = ii; }
            }
            p.part_val[(size_t)y * gridDim.x + blockIdx.x] = bv;
            p.part_idx[(size_t)y * gridDim.x + blockIdx.x] = bidx;
        }
    }
}

int dense_gemv_grid(int N) { const int wgs = (N / 2 + 3) / 4; return wgs > 2048 ? 2048 : (wgs < 1 ? 1 : wgs); }

template <int PRO, int EPI>
static hipError_t dense_launch_t(const GemvParams& p, int ny, hipStream_t s) {
    const size_t lds = (size_t)(p.w.K + 16) * sizeof(float);
    if (p.w.fmt == WFMT_F32) dense_gemv_kernel<PRO, EPI, 1><<<dim3(dense_gemv_grid(p.w.N), ny), dim3(256), lds, s>>>(p);
    else dense_gemv_kernel<PRO, EPI, 0><<<dim3(dense_gemv_grid(p.w.N), ny), dim3(256), lds, s>>>(p);
    return hipGetLastError();
}
static hipError_t launch_dense_gemv(const GemvParams& p, int ny, int pro, int epi, hipStream_t s) {
    if (p.w.K % 32 || p.w.K > 10240 || p.w.N % 2) return hipErrorInvalidValue;
    if ((pro == PRO_RMS_MUL || pro == PRO_RMS_MUL_SUM) && p.mul == nullptr) return hipErrorInvalidValue;
    if (pro == PRO_RMS_MUL_SUM) return hipErrorInvalidValue;      // Q4 decode step only
#define VOX_CASE(P_, E_) if (pro == P_ && epi == E_) return dense_launch_t<P_, E_>(p, ny, s)
    VOX_CASE(PRO_NONE, EPI_STORE); VOX_CASE(PRO_NONE, EPI_RESID); VOX_CASE(PRO_NONE, EPI_GELU); VOX_CASE(PRO_NONE, EPI_SWIGLU);
    VOX_CASE(PRO_RMS, EPI_STORE); VOX_CASE(PRO_RMS, EPI_ARGMAX); VOX_CASE(PRO_RMS, EPI_SWIGLU); VOX_CASE(PRO_RMS_MUL, EPI_SWIGLU);
    VOX_CASE(PRO_RMS, EPI_ROPE_KV);
#undef VOX_CASE
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// Q4 GEMM on MFMA (v_mfma_f32_16x16x32_bf16)
// ------------------------------------------------------------------------------------------------
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__device__ __forceinline__ uint32_t bf16_rne_bits(float x) {   // round-to-nearest-even, finite inputs
    const uint32_t u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// 8 floats -> packed bf16 hi plane + bf16 lo plane (x ~= hi + lo, |err| <= 2^-17 |x|).  gfx950 has a hardware
// round-to-nearest-even pack (v_cvt_pk_bf16_f32): a pair costs 2 cvt + 2 unpack + 2 sub instead of ~14 integer ops.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    union { bf16x2_t b; uint32_t u; } c; c.b = __builtin_convertvector(v, bf16x2_t); return c.u;
}
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(a, b);
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
__device__ __forceinline__ void split_bf16x8(const float4 a, const float4 b, uint4& hi, uint4& lo) {
    split_pair(a.x, a.y, hi.x, lo.x); split_pair(a.z, a.w, hi.y, lo.y);
    split_pair(b.x, b.y, hi.z, lo.z); split_pair(b.z, b.w, hi.w, lo.w);
}
// one dword of a Q4 chunk (bytes 4g..4g+3) -> 8 exact bf16 integers (q-8):
// slots 0..3 = low nibbles (elements 4g..4g+3), slots 4..7 = high nibbles (elements 16+4g..16+4g+3)
__device__ __forceinline__ uint32_t pack_bf16_exact(float a, float b) {   // a, b small integers: exact in bf16
    return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xFFFF0000u);
}
__device__ __forceinline__ uint4 q4_dword_to_bf16x8(uint32_t w) {
    const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu;
    uint4 o;
    o.x = pack_bf16_exact(ub0(lo) - 8.0f, ub1(lo) - 8.0f);
    o.y = pack_bf16_exact(ub2(lo) - 8.0f, ub3(lo) - 8.0f);
    o.z = pack_bf16_exact(ub0(hi) - 8.0f, ub1(hi) - 8.0f);
    o.w = pack_bf16_exact(ub2(hi) - 8.0f, ub3(hi) - 8.0f);
    return o;
}
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) {
    union { uint4 u; bf16x8 b; } c; c.u = v; return c.b;
}

// one dword of a Q4 chunk (8 nibbles) -> 8 bf16 operands 128+q (bits 0x4300|q) in the K-slot order
// {4g, 4g+2, 16+4g, 16+4g+2, 4g+1, 4g+3, 16+4g+1, 16+4g+3}: 2 v_and + 1 shift + 4 v_perm_b32 (no inline asm: an asm VALU
// result feeding an MFMA is invisible to the hazard recognizer)
__device__ __forceinline__ uint4 q4_dword_to_bf16x8_biased(uint32_t w) {
    const uint32_t lo = w & 0x0F0F0F0Fu, hi = (w >> 4) & 0x0F0F0F0Fu, c = 0x43434343u;
    return make_uint4(__builtin_amdgcn_perm(lo, c, 0x00060004u), __builtin_amdgcn_perm(hi, c, 0x00060004u),
                      __builtin_amdgcn_perm(lo, c, 0x00070005u), __builtin_amdgcn_perm(hi, c, 0x00070005u));
}

// ---- XF: activations of a <= 16-row batch stored ONCE as MFMA A-fragments (bf16 hi + lo planes), so the batched-decode
// GEMMs load their operands with plain coalesced dwordx4 and spend no VALU on conversion (the conversion used to be redone
// by every wave of every workgroup).  Plane layout: uint4 [K/128 (q)][4 (j)][64 lanes]; lane = 16*g + row holds block
// 4q+j elements in the K-slot order of the bit-trick B fragment {4g, 4g+2, 16+4g, 16+4g+2, 4g+1, 4g+3, 16+4g+1, 16+4g+3};
// the lo plane follows the hi plane (K/128 * 256 uint4 later).  Producers call xf_store4 with k % 4 == 0.
__device__ __forceinline__ void xf_store4(uint16_t* __restrict__ xf, int K, int row, int k, float4 v) {
    const int q = k >> 7, j = (k >> 5) & 3, e = k & 31, half = e >> 4, g = (e & 15) >> 2;
    const size_t base = ((size_t)((q * 4 + j) * 64 + g * 16 + row)) * 8 + 2 * half;
    const size_t plane = (size_t)(K >> 7) * 256 * 8;
    uint32_t hi, lo;
    split_pair(v.x, v.z, hi, lo); *reinterpret_cast<uint32_t*>(xf + base) = hi; *reinterpret_cast<uint32_t*>(xf + plane + base) = lo;
    split_pair(v.y, v.w, hi, lo); *reinterpret_cast<uint32_t*>(xf + base + 4) = hi; *reinterpret_cast<uint32_t*>(xf + plane + base + 4) = lo;
}

// Workgroup tile: 64 rows of x (4 MFMA m-tiles) x 64 weight rows (wave w owns n-tile w).
// A operand = activations (rows m, k = 8*(lane>>4)..+7), B operand = integer weights (cols n = lane&15),
// D[m][n]: lane holds n = lane&15, m = 4*(lane>>4) + reg.  The K order inside one MFMA follows the Q4 chunk:
// lane group g = lane>>4 contributes elements {4g..4g+3, 16+4g..16+4g+3} of the block.
template <int EPI, int FMT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void q4_gemm_k32_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(16))) uint4 lds[2][2][256];   // [buffer][hi/lo][mt*64 + lane]
    const int K = p.w.K, nb = p.w.nb, N = p.w.N, M = p.M;
    (void)K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    // staging role: thread -> (row sm, group sg)
    const int sm = tid >> 2, sg = tid & 3;
    const bool srow_ok = (m0 + sm) < M;
    const float* xrow = p.x + (size_t)(srow_ok ? (m0 + sm) : 0) * p.x_stride;
    const int slot = ((sm >> 4) * 4 + sg) * 16 + (sm & 15);   // == mt*64 + (g*16 + mi)
    // MFMA role: weight row for this lane
    const int wn = n0 + wave * 16 + (lane & 15), wg = lane >> 4;
    const bool wrow_ok = wn < N;
    // FMT Q4_0: one dword of nibbles per lane per block (+ f16 scale). FMT BF16: 8 bf16 weights (16 B) per lane per 32-k step,
    // natural k order (k = 32b + 8g .. +7), no scale.
    const uint32_t* wq = reinterpret_cast<const uint32_t*>(p.w.qs) + (size_t)(wrow_ok ? wn : 0) * nb * 4 + wg;
    const uint16_t* ws = FMT == WFMT_Q4_0 ? p.w.sc + (size_t)(wrow_ok ? wn : 0) * nb : nullptr;
    const uint4* wd16 = p.w.qs + (size_t)(wrow_ok ? wn : 0) * nb * 4 + wg;
    const int xoff_a = FMT == WFMT_Q4_0 ? 4 * sg : 8 * sg, xoff_b = FMT == WFMT_Q4_0 ? 16 + 4 * sg : 8 * sg + 4;

    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define VOX_GLOAD(B_)                                                                              \
    xa = *reinterpret_cast<const float4*>(xrow + 32 * (B_) + xoff_a);      /* unconditional loads (row clamped)  */  \
    xb = *reinterpret_cast<const float4*>(xrow + 32 * (B_) + xoff_b);                                               \
    if (FMT == WFMT_Q4_0) { wd = wq[(size_t)(B_) * 4]; wsc = ws[(B_)]; }   /* out-of-range rows/cols read row 0, never stored */ \
    else { wdv = wd16[(size_t)(B_) * 4]; }
#define VOX_STAGE(BUF_)                                                                            \
    { uint4 hi_, lo_; split_bf16x8(xa, xb, hi_, lo_); lds[(BUF_)][0][slot] = hi_; lds[(BUF_)][1][slot] = lo_; }
    // two-deep register pipeline: loads for block b+2 are issued while block b is on the MFMAs and block b+1
    // (loaded one iteration ago, so already landed) is converted and staged to the other LDS buffer.
    float4 xa, xb; uint32_t wd = 0; uint16_t wsc = 0; uint4 wdv = make_uint4(0, 0, 0, 0);
    VOX_GLOAD(0)
    VOX_STAGE(0)
    uint32_t cur_wd = wd; uint16_t cur_sc = wsc; uint4 cur_wdv = wdv;
    { const int b1 = min(1, nb - 1); VOX_GLOAD(b1) }
    __syncthreads();
    for (int b = 0; b < nb; b++) {
        const int buf = b & 1;
        const float4 sxa = xa, sxb = xb; const uint32_t swd = wd; const uint16_t ssc = wsc; const uint4 swdv = wdv;   // block b+1 (in flight since last iteration)
        { const int b2 = min(b + 2, nb - 1); VOX_GLOAD(b2) }   // unconditional (clamped): a branch here forces phi copies + vmcnt waits
        const bf16x8 bw = as_bf16x8(FMT == WFMT_Q4_0 ? q4_dword_to_bf16x8(cur_wd) : cur_wdv);
        const float d = FMT == WFMT_Q4_0 ? f16_bits_to_f32(cur_sc) : 1.0f;
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            const bf16x8 ah = as_bf16x8(lds[buf][0][mt * 64 + lane]);
            const bf16x8 al = as_bf16x8(lds[buf][1][mt * 64 + lane]);
            f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw, t, 0, 0, 0);
            acc[mt][0] = fmaf(d, t[0], acc[mt][0]); acc[mt][1] = fmaf(d, t[1], acc[mt][1]);
            acc[mt][2] = fmaf(d, t[2], acc[mt][2]); acc[mt][3] = fmaf(d, t[3], acc[mt][3]);
        }
        if (b + 1 < nb) {
            uint4 hi_, lo_; split_bf16x8(sxa, sxb, hi_, lo_); lds[buf ^ 1][0][slot] = hi_; lds[buf ^ 1][1][slot] = lo_;
            cur_wd = swd; cur_sc = ssc; cur_wdv = swdv;
        }
        __syncthreads();
    }
#undef VOX_GLOAD
#undef VOX_STAGE
    // epilogue: lane holds D[m = m0 + mt*16 + 4*(lane>>4) + r][n = wn]
    const float bias = (p.bias && wrow_ok) ? p.bias[wn] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + mt * 16 + 4 * (lane >> 4) + r;
            float v = acc[mt][r] + bias;
            if (EPI == EPI_SWIGLU) {
                const float other = dpp_mov<0xB1>(v);        // lane^1; rows interleaved: even n = gate, odd n = up
                if (m < M && wrow_ok && !(wn & 1)) p.out[(size_t)m * p.out_stride + (wn >> 1)] = silu_f(v) * other;
            } else if (m < M && wrow_ok) {
                if (EPI == EPI_RESID) v = v + p.resid[(size_t)m * p.resid_stride + wn];
                if (EPI == EPI_GELU) v = gelu_f(v);
                p.out[(size_t)m * p.out_stride + wn] = v;
            }
        }
}

// ---- main MFMA GEMM: K step of 128 (four Q4_0 blocks / four 32-k dense steps per barrier), workgroup tile
// (16*MT rows) x (64*NT cols); wave w owns n-tiles [w*NT, w*NT+NT) x all MT m-tiles.  Per barrier a wave issues
// 4*NT*MT*2 MFMAs against 8*MT ds_read_b128 -- 4x the work per barrier of the K-32 kernel above (kept as the
// fallback for K % 128 != 0).  Same operand conventions: A = activations (hi+lo bf16 through LDS in fragment order),
// B = integer Q4 weights straight from global (one dword per lane per block) or dense bf16 (16 B per lane per step).
// TB: Q4 weights come from the tile-ordered copy (one coalesced dwordx4 per lane per n-tile per K step = the lane's dword of the four blocks)
// instead of four strided dword loads from the row planes -- used for the 33..48-row prefill tile (MT = 3), where the weight stream is the bound.
template <int MT, int NT, int EPI, int FMT, int TB = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void q4_gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 glds[];      // [buf 2][hi/lo 2][j 4][MT][64]
    constexpr int PLANE = 4 * MT * 64, BUF = 2 * PLANE;
    const int nb = p.w.nb, N = p.w.N, M = p.M, nq = nb >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * (16 * MT), n0 = blockIdx.x * (64 * NT);
    const int KSP = p.ksplit > 1 ? p.ksplit : 1, kz = blockIdx.z;                               // split-K: this workgroup's K steps [qb, qe)
    const int qb = (int)((long)nq * kz / KSP), qe = (int)((long)nq * (kz + 1) / KSP);
    // staging role: fragment f = tid + 256*i -> (row sm = f>>4, block j = (f>>2)&3, group g = f&3): the 16 lanes of a row group read 256 contiguous bytes of their row.
    // Round 5: fragment (g, m) of block j lives at slot g*16 + (m ^ (4j + g)) of its (j, m-tile) block instead of g*16 + m -- the 16 writers of a row (same m, all (j, g))
    // used to hit slots 16 and 128 apart = the same four banks, a 16-way conflict on every ds_write_b128 (PMC: 11 conflict cycles per LDS instruction); XOR-ed with 16
    // distinct values they cover all 64 banks, and the readers (lane (g, m) reads its own permuted slot) still sweep 16 consecutive 16-byte slots per lane group.
    // (Remapping the THREADS to consecutive slots instead -- 16 rows per load instruction -- removed the conflicts too and was 3 % slower: profiles/r05_gemm_staging_ab.txt.)
    const float* xrow[MT]; int slot[MT], xo_a[MT], xo_b[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) {
        const int f = tid + 256 * i, sm = f >> 4, j = (f >> 2) & 3, g = f & 3;
        xrow[i] = p.x + (size_t)min(m0 + sm, M - 1) * p.x_stride;
#ifdef VOX_GEMM_OLD_STAGING      /* measurement build (build.py gemm_oldstage): the pre-round-5 layout, for the same-box A/B in profiles/r05_gemm_staging_ab.txt */
        slot[i] = (j * MT + (sm >> 4)) * 64 + g * 16 + (sm & 15);
#else
        slot[i] = (j * MT + (sm >> 4)) * 64 + g * 16 + ((sm & 15) ^ (4 * j + g));
#endif
        xo_a[i] = 32 * j + (FMT == WFMT_Q4_0 ? 4 * g : 8 * g); xo_b[i] = 32 * j + (FMT == WFMT_Q4_0 ? 16 + 4 * g : 8 * g + 4);
    }
    // MFMA role
    const int wg = lane >> 4;
    int rdl[4];      // where this lane's A fragment of block j sits inside its 64-slot block (see the staging map)
#pragma unroll
    for (int j = 0; j < 4; j++) {
#ifdef VOX_GEMM_OLD_STAGING
        rdl[j] = lane;
#else
        rdl[j] = (lane & 48) | ((lane & 15) ^ (4 * j + wg));
#endif
    }
    const uint32_t* wq[NT]; const uint16_t* ws[NT]; const uint4* wd16[NT]; const uint4* wqt[NT]; int wn[NT]; bool wok[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        wn[t] = n0 + (wave * NT + t) * 16 + (lane & 15); wok[t] = wn[t] < N;
        const size_t row = (size_t)min(wn[t], N - 1);
        wq[t] = reinterpret_cast<const uint32_t*>(p.w.qs) + row * nb * 4 + wg;
        ws[t] = FMT == WFMT_Q4_0 ? p.w.sc + row * nb : nullptr;
        wd16[t] = p.w.qs + row * nb * 4 + wg;
        const size_t T = (size_t)min((n0 >> 4) + wave * NT + t, ((N + 15) >> 4) - 1);
        wqt[t] = TB ? p.w.qt + T * nq * 64 + lane : nullptr;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < MT; i++) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 xa[MT], xb[MT];
    uint32_t wd[NT][4]; uint2 wsc[NT]; uint4 wdv[NT][4];
#define VOX_GLOAD(Q_)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                           \
        xa[i] = *reinterpret_cast<const float4*>(xrow[i] + 128 * (Q_) + xo_a[i]);                              \
        xb[i] = *reinterpret_cast<const float4*>(xrow[i] + 128 * (Q_) + xo_b[i]);                              \
    }                                                                                                          \
    _Pragma("unroll") for (int t = 0; t < NT; t++) {                                                           \
        if (FMT == WFMT_Q4_0) {                                                                                \
            if (TB) { const uint4 w4 = ld_nt_u4(wqt[t] + (size_t)64 * (Q_)); wd[t][0] = w4.x; wd[t][1] = w4.y; wd[t][2] = w4.z; wd[t][3] = w4.w; } \
            else { _Pragma("unroll") for (int j = 0; j < 4; j++) wd[t][j] = wq[t][(size_t)(4 * (Q_) + j) * 4]; } \
            wsc[t] = *reinterpret_cast<const uint2*>(ws[t] + 4 * (Q_));                                        \
        } else {                                                                                               \
            _Pragma("unroll") for (int j = 0; j < 4; j++) wdv[t][j] = wd16[t][(size_t)(4 * (Q_) + j) * 4];     \
        }                                                                                                      \
    }
#define VOX_STAGE(BUF_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                           \
        uint4 hi_, lo_; split_bf16x8(sxa[i], sxb[i], hi_, lo_);                                                \
        glds[(BUF_) * BUF + slot[i]] = hi_; glds[(BUF_) * BUF + PLANE + slot[i]] = lo_;                        \
    }
    VOX_GLOAD(qb)
    float4 sxa[MT], sxb[MT];
#pragma unroll
    for (int i = 0; i < MT; i++) { sxa[i] = xa[i]; sxb[i] = xb[i]; }
    VOX_STAGE(0)
    uint32_t cwd[NT][4]; uint2 csc[NT]; uint4 cwdv[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++) { csc[t] = wsc[t];
#pragma unroll
        for (int j = 0; j < 4; j++) { cwd[t][j] = wd[t][j]; cwdv[t][j] = wdv[t][j]; } }
    { const int q1 = min(qb + 1, qe - 1); VOX_GLOAD(q1) }
    __syncthreads();
    for (int q = qb; q < qe; q++) {
        const int buf = (q - qb) & 1;
#pragma unroll
        for (int i = 0; i < MT; i++) { sxa[i] = xa[i]; sxb[i] = xb[i]; }          // k-step q+1 (in flight since last iteration)
        uint32_t nwd[NT][4]; uint2 nsc[NT]; uint4 nwdv[NT][4];
#pragma unroll
        for (int t = 0; t < NT; t++) { nsc[t] = wsc[t];
#pragma unroll
            for (int j = 0; j < 4; j++) { nwd[t][j] = wd[t][j]; nwdv[t][j] = wdv[t][j]; } }
        { const int q2 = min(q + 2, qe - 1); VOX_GLOAD(q2) }                      // unconditional (clamped) prefetch
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const bf16x8 bw = as_bf16x8(FMT == WFMT_Q4_0 ? q4_dword_to_bf16x8(cwd[t][j]) : cwdv[t][j]);
                float d = 1.0f;
                if (FMT == WFMT_Q4_0) {
                    const uint32_t pr = (j & 2) ? csc[t].y : csc[t].x;
                    d = f16_bits_to_f32((uint16_t)((j & 1) ? (pr >> 16) : (pr & 0xFFFFu)));
                }
#pragma unroll
                for (int i = 0; i < MT; i++) {
                    const bf16x8 ah = as_bf16x8(glds[buf * BUF + (j * MT + i) * 64 + rdl[j]]);
                    const bf16x8 al = as_bf16x8(glds[buf * BUF + PLANE + (j * MT + i) * 64 + rdl[j]]);
                    f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw, tt, 0, 0, 0);
                    acc[t][i][0] = fmaf(d, tt[0], acc[t][i][0]); acc[t][i][1] = fmaf(d, tt[1], acc[t][i][1]);
                    acc[t][i][2] = fmaf(d, tt[2], acc[t][i][2]); acc[t][i][3] = fmaf(d, tt[3], acc[t][i][3]);
                }
            }
        }
        if (q + 1 < qe) {
            VOX_STAGE(buf ^ 1)
#pragma unroll
            for (int t = 0; t < NT; t++) { csc[t] = nsc[t];
#pragma unroll
                for (int j = 0; j < 4; j++) { cwd[t][j] = nwd[t][j]; cwdv[t][j] = nwdv[t][j]; } }
        }
        __syncthreads();
    }
#undef VOX_GLOAD
#undef VOX_STAGE
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const float bias = (p.bias && wok[t] && kz == 0) ? p.bias[wn[t]] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + i * 16 + 4 * (lane >> 4) + r;
                float v = acc[t][i][r] + bias;
                if (EPI == EPI_SWIGLU) {
                    const float other = dpp_mov<0xB1>(v);        // lane^1; rows interleaved: even n = gate, odd n = up
                    if (m < M && wok[t] && !(wn[t] & 1)) p.out[(size_t)m * p.out_stride + (wn[t] >> 1)] = silu_f(v) * other;
                } else if (m < M && wok[t]) {
                    if (EPI == EPI_RESID) v = v + p.resid[(size_t)m * p.resid_stride + wn[t]];
                    if (EPI == EPI_GELU) v = gelu_f(v);
                    (p.out + (size_t)kz * p.M * p.out_stride)[(size_t)m * p.out_stride + wn[t]] = v;
                }
            }
    }
}

// ---- skinny MFMA GEMM for M <= 16 rows (batched decode: one row per sequence).  HBM-bound: the weights are streamed once
// for the whole batch.  No LDS on the operand path, no barriers until the final reduction:
//  * a wave owns NTW n-tiles (16 weight rows each) and every KS-th 128-wide K step (split-K across the KS waves of the
//    workgroup, combined through LDS in a fixed order -> deterministic);
//  * weights: one global_load_dwordx4 per lane per n-tile per K step -- lane (n = l&15, g = l>>4) fetches the whole 16-byte
//    block 4q+g of row n (1 KB contiguous per wave-load from the row-major plane), then a 4x4 dword transpose across the four
//    16-lane rows (2x v_permlane16_swap + 2x v_permlane32_swap) leaves it with dword g of blocks 4q..4q+3 = its MFMA B fragments;
//  * activations: f32 rows read straight from global/L2 in fragment order and split hi+lo bf16 in registers, reused by the NTW tiles.
// Tile layout (second copy of a Q4 weight, built at load for the batched-decode linears): for n-tile T (16 rows) and K step q
// (4 blocks) qt[(T*nq + q)*64 + lane] is exactly the uint4 the lane needs -- dword g = lane>>4 of blocks 4q..4q+3 of row
// 16T + (lane&15) -- so a wave-load is 1 KB contiguous and consecutive K steps are contiguous (pure streaming, no transpose);
// st[((T*nq + q)*16 + n)*4 + j] is the f16 scale of block 4q+j of row 16T+n.
__global__ void q4_tile_build_kernel(Q4W w, uint4* __restrict__ qt, uint16_t* __restrict__ st, int n_tiles) {
    const int nq = w.nb >> 2;
    const long total = (long)n_tiles * nq * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63); const long tq = i >> 6; const int q = (int)(tq % nq), T = (int)(tq / nq);
        const int g = lane >> 4, n = T * 16 + (lane & 15);
        uint32_t d[4] = {0x88888888u, 0x88888888u, 0x88888888u, 0x88888888u};
        if (n < w.N) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(w.qs) + ((size_t)n * w.nb + 4 * q) * 4 + g;
#pragma unroll
            for (int j = 0; j < 4; j++) d[j] = src[j * 4];
        }
        qt[i] = make_uint4(d[0], d[1], d[2], d[3]);
        if (g == 0) {
#pragma unroll
            for (int j = 0; j < 4; j++) st[(tq * 16 + (lane & 15)) * 4 + j] = n < w.N ? w.sc[(size_t)n * w.nb + 4 * q + j] : (uint16_t)0;
        }
    }
}
hipError_t launch_q4_tile_build(Q4W w, uint4* qt, uint16_t* st, hipStream_t s) {
    if (w.nb % 4) return hipErrorInvalidValue;
    const int n_tiles = (w.N + 15) / 16;
    q4_tile_build_kernel<<<dim3(2048), dim3(256), 0, s>>>(w, qt, st, n_tiles);
    return hipGetLastError();
}

// STEPS > 0 (batched decode step, tile-ordered weights + XF activations): every wave owns exactly STEPS K-steps and runs them as straight-line
// code with the loads issued IN CONSUMPTION ORDER, D steps ahead (see the main loop).
// NTW == 3 (w1|w3 of the real model: 2304 n-tiles = 768 workgroups of 4 waves = exactly 3 per CU): 256-thread workgroups, <= 168 VGPRs so that the
// three workgroups of a CU are resident together (with NTW = 4 the 576 workgroups needed 256 VGPRs -> 512 slots -> a second round of 64).
template <int NTW, int EPI, int TILED, int XIN, int PRO, int STEPS = 0>
__global__ __launch_bounds__(NTW == 3 ? 256 : 512, NTW == 3 ? 3 : 1) void q4_skinny_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float sred[];      // [KS][NTW][64][4]
    __shared__ float s_rstd[16]; __shared__ float s_pp[32 * 16];
    const int nb = p.w.nb, N = p.w.N, M = p.M, nq = nb >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, KS = blockDim.x >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int nbase = blockIdx.x * (16 * NTW);
    const int tlw = wave == 0 ? (int)blockIdx.x * 4 : 1; (void)tlw;      // timeline builds: wave 0 of every workgroup stamps (tl_stamp keeps indices % 4 == 0)
    VOX_TL(p.tl_slot, tlw, 0);
    const float* xrow = p.x + (size_t)min(li, M - 1) * p.x_stride;
    const uint4* wq[NTW]; const uint16_t* ws[NTW];
    const int n_tiles = (N + 15) >> 4;
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        if (TILED) {
            const size_t T = (size_t)min(blockIdx.x * NTW + t, n_tiles - 1);
            wq[t] = p.w.qt + T * nq * 64 + lane; ws[t] = p.w.st + (T * nq * 16 + li) * 4;
        } else {
            const size_t row = (size_t)min(nbase + t * 16 + li, N - 1);
            wq[t] = p.w.qs + row * nb + g; ws[t] = p.w.sc + row * nb;
        }
    }
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint4 wv[NTW], wvn[NTW]; uint2 sv[NTW], svn[NTW]; float4 xa[4], xb[4];      // XIN: xa = hi fragments, xb = lo fragments (bit patterns)
    const uint4* xfh = p.xf + lane; const uint4* xfl = p.xf + (size_t)nq * 256 + lane;
#define VOX_WLOAD(WV_, SV_, Q_)                                                                            \
    _Pragma("unroll") for (int t = 0; t < NTW; t++) {                                                      \
        WV_[t] = ld_nt_u4(wq[t] + (TILED ? 64 : 4) * (Q_)); SV_[t] = *reinterpret_cast<const uint2*>(ws[t] + (TILED ? 64 : 4) * (Q_)); }
#define VOX_XLOAD(Q_)                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                        \
        if (XIN) {                                                                                         \
            xa[j] = *reinterpret_cast<const float4*>(xfh + ((Q_) * 4 + j) * 64);                           \
            xb[j] = *reinterpret_cast<const float4*>(xfl + ((Q_) * 4 + j) * 64);                           \
        } else {                                                                                           \
            xa[j] = *reinterpret_cast<const float4*>(xrow + 128 * (Q_) + 32 * j + 4 * g);                  \
            xb[j] = *reinterpret_cast<const float4*>(xrow + 128 * (Q_) + 32 * j + 16 + 4 * g); } }
// B fragments by bit tricks: bf16 bits 0x4300 | q are exactly 128 + q, so one dword of 8 nibbles becomes 8 bf16 operands
// in 7 VALU ops (v_and_or_b32 on nibble pairs 16 bits apart).  sum_k x_k (q_k - 8) = sum_k x_k (128 + q_k) - 136 sum_k x_k, and
// -136 sum_k x_k per (row, block) comes from one extra MFMA pair against a B of bf16(-136), shared by the wave's NTW tiles and
// already in the accumulator layout.  K-slot order of a lane group g: {4g, 4g+2, 16+4g, 16+4g+2, 4g+1, 4g+3, 16+4g+1, 16+4g+3}.
#define VOX_SSTEP(WV_, SV_, XA_, XB_)                                                                      \
    {                                                                                                      \
        const bf16x8 m136 = as_bf16x8(make_uint4(0xC308C308u, 0xC308C308u, 0xC308C308u, 0xC308C308u));     \
        uint32_t dw[NTW][4];                                                                               \
        _Pragma("unroll") for (int t = 0; t < NTW; t++) {                                                  \
            dw[t][0] = WV_[t].x; dw[t][1] = WV_[t].y; dw[t][2] = WV_[t].z; dw[t][3] = WV_[t].w;            \
            if (!TILED) { /* 4x4 dword transpose across the four 16-lane rows */                           \
                auto s01 = __builtin_amdgcn_permlane16_swap(WV_[t].x, WV_[t].y, false, false);             \
                auto s23 = __builtin_amdgcn_permlane16_swap(WV_[t].z, WV_[t].w, false, false);             \
                auto u02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);                 \
                auto u13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);                 \
                dw[t][0] = u02[0]; dw[t][1] = u13[0]; dw[t][2] = u02[1]; dw[t][3] = u13[1];                \
            }                                                                                              \
        }                                                                                                  \
        /* block j outermost: one correction accumulator (4 registers) is live at a time, not four */       \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                    \
            uint4 ah, al;                                                                                  \
            if (XIN) {                                                                                     \
                ah = make_uint4(__float_as_uint(XA_[j].x), __float_as_uint(XA_[j].y), __float_as_uint(XA_[j].z), __float_as_uint(XA_[j].w)); \
                al = make_uint4(__float_as_uint(XB_[j].x), __float_as_uint(XB_[j].y), __float_as_uint(XB_[j].z), __float_as_uint(XB_[j].w)); \
            } else {                                                                                       \
                split_pair(XA_[j].x, XA_[j].z, ah.x, al.x); split_pair(XB_[j].x, XB_[j].z, ah.y, al.y);    \
                split_pair(XA_[j].y, XA_[j].w, ah.z, al.z); split_pair(XB_[j].y, XB_[j].w, ah.w, al.w);    \
            }                                                                                              \
            const f32x4 sx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
            const f32x4 cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), m136, sx, 0, 0, 0);    \
            _Pragma("unroll") for (int t = 0; t < NTW; t++) {                                              \
                const bf16x8 bw = as_bf16x8(q4_dword_to_bf16x8_biased(dw[t][j]));                          \
                const uint32_t sc2 = (j >> 1) ? SV_[t].y : SV_[t].x;                                       \
                const float d = f16_bits_to_f32((uint16_t)((j & 1) ? (sc2 >> 16) : (sc2 & 0xFFFFu)));      \
                f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), bw, cs, 0, 0, 0);        \
                tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), bw, tt, 0, 0, 0);              \
                acc[t] = __builtin_elementwise_fma((f32x4){d, d, d, d}, tt, acc[t]);                        \
            }                                                                                              \
        }                                                                                                  \
    }
    // fused RMSNorm: the producers' partial sums of squares (thread -> (row = tid & 15, chunk = tid >> 4); up to 12 partials per thread), and the
    // epilogue operands of the tile this wave will finish (t = wave: residual / norm weight / position -- not a dependent round trip behind the
    // split-K barrier; PMC: the batched-decode kernels sat parked in s_waitcnt for > 50 % of their cycles).  Unconditional clamped loads (a
    // branch around a load makes hipcc drain vmcnt).  vmcnt retires IN ORDER and these words were written by the previous launches on other XCDs
    // (memory round trips): the straight-line path requests them BEHIND its first K step's operands, so that step does not wait for them.
    float pv[12]; const int prow = tid & 15, pch = tid >> 4, nch = blockDim.x >> 4;
    float pre_res[4] = {0.f, 0.f, 0.f, 0.f}; float pre_xw = 0.f; int pre_pos[4] = {0, 0, 0, 0}; int pre_row[4] = {0, 0, 0, 0};
#define VOX_PRELOADS                                                                                       \
    {                                                                                                      \
        if (PRO) {                                                                                         \
            _Pragma("unroll") for (int u = 0; u < 12; u++) {                                               \
                const int pi = pch + u * nch;                                                              \
                const float v = p.ssq_part[(size_t)min(pi, p.n_part - 1) * 16 + prow];                     \
                pv[u] = pi < p.n_part ? v : 0.f;                                                           \
            }                                                                                              \
        }                                                                                                  \
        const int n0_ = min(nbase + min(wave, NTW - 1) * 16 + li, N - 1);                                  \
        if (EPI == EPI_RESID_XF || EPI == EPI_RESID) {                                                     \
            _Pragma("unroll") for (int r = 0; r < 4; r++) pre_res[r] = p.resid[(size_t)min(4 * g + r, M - 1) * p.resid_stride + n0_]; \
        }                                                                                                  \
        if (EPI == EPI_RESID_XF) { pre_xw = p.xf_w[n0_]; if (p.xf_w2) pre_xw *= p.xf_w2[n0_]; }            \
        if (EPI == EPI_ROPE_KV) {                                                                          \
            _Pragma("unroll") for (int r = 0; r < 4; r++) pre_pos[r] = p.pos[min(4 * g + r, M - 1)];       \
            _Pragma("unroll") for (int r = 0; r < 4; r++) pre_row[r] = p.kv_row ? p.kv_row[min(4 * g + r, M - 1)] : min(4 * g + r, M - 1); \
        }                                                                                                  \
    }
    constexpr bool STRAIGHT = STEPS > 0 && TILED && XIN;
    if (!STRAIGHT) VOX_PRELOADS
    // wave's K steps: [q, qend) step qs -- a contiguous range when tiled (pure streaming), interleaved otherwise
    const int per = (nq + KS - 1) / KS;
    int q = TILED ? wave * per : wave;
    const int qend = TILED ? min(q + per, nq) : nq, qs = TILED ? 1 : KS;
    if (STRAIGHT) {
        // vmcnt retires loads IN ORDER: waiting for a young L2-hit load (an activation fragment) also waits for every older HBM load.  The legacy
        // loop issued the NEXT step's weights before the CURRENT step's fragments, so every step paid a full HBM round trip (timeline:
        // ~2 us per K step, waves parked > 50 %); a ring that refilled weights out of consumption order was slower still (round-2 experiments,
        // profiles/r02_batch16_ring_prefetch.txt).  Here: w(s), x(s) of steps 0..D-1 first, then after step s is multiplied, w(s+D), x(s+D) -- the
        // issue order IS the consumption order, D steps of latency cover; fully unrolled (STEPS is exact: the launcher only picks this
        // instantiation when every wave of the workgroup owns STEPS steps), so there is no control flow around any load.
        constexpr int D = NTW >= 3 ? 2 : 3, DB = STEPS <= 0 ? 1 : (D < STEPS ? D : STEPS);      // (DB = 1 keeps the dead STEPS = 0 instantiation legal)
        uint4 wr[DB][NTW]; uint2 sr[DB][NTW]; float4 xr[DB][2][4];
#define VOX_ISSUE(B_, Q_)                                                                                  \
        { VOX_WLOAD(wr[B_], sr[B_], (Q_))                                                                  \
          _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                  \
              xr[B_][0][j] = *reinterpret_cast<const float4*>(xfh + ((Q_) * 4 + j) * 64);                  \
              xr[B_][1][j] = *reinterpret_cast<const float4*>(xfl + ((Q_) * 4 + j) * 64); } }
        VOX_ISSUE(0, q)
        __builtin_amdgcn_sched_barrier(0);
        VOX_PRELOADS
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 1; u < DB; u++) VOX_ISSUE(u, q + u)
#pragma unroll
        for (int st = 0; st < STEPS; st++) {
            __builtin_amdgcn_sched_barrier(0);
            VOX_SSTEP(wr[st % DB], sr[st % DB], xr[st % DB][0], xr[st % DB][1])
            // the norm partials are older than every load still in flight: fold them now (12 registers less in the rest of the loop)
            if (st == 0 && PRO) s_pp[pch * 16 + prow] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7])) + ((pv[8] + pv[9]) + (pv[10] + pv[11]));
            if (st + DB < STEPS) VOX_ISSUE(st % DB, q + st + DB)          // compile-time condition
            if (st == 0) { VOX_TL(p.tl_slot, tlw, 1); }                   // first K step multiplied
        }
#undef VOX_ISSUE
    } else if (q < qend) {
        VOX_WLOAD(wv, sv, q)
        for (;;) {
            // the weight prefetch for the NEXT step is issued (and pinned) before anything of this step: hipcc otherwise sinks
            // it next to its consumers to save registers and exposes the HBM latency on every step
            { const int qn = min(q + qs, qend - 1); VOX_WLOAD(wvn, svn, qn) }
            VOX_XLOAD(q)
            __builtin_amdgcn_sched_barrier(0);
            VOX_SSTEP(wv, sv, xa, xb)
            VOX_TL(p.tl_slot, tlw, 1);                     // (rewritten every other step: completion of the last even step)
            q += qs; if (q >= qend) break;
            { const int qn = min(q + qs, qend - 1); VOX_WLOAD(wv, sv, qn) }
            VOX_XLOAD(q)
            __builtin_amdgcn_sched_barrier(0);
            VOX_SSTEP(wvn, svn, xa, xb)
            q += qs; if (q >= qend) break;
        }
    }
#undef VOX_WLOAD
#undef VOX_XLOAD
#undef VOX_SSTEP
#undef VOX_PRELOADS
    VOX_TL(p.tl_slot, tlw, 2);                             // K loop done
    // split-K combine (fixed order) + epilogue: wave t finishes tile t
#pragma unroll
    for (int t = 0; t < NTW; t++)
        *reinterpret_cast<float4*>(sred + ((size_t)(wave * NTW + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    if (PRO && !STRAIGHT) s_pp[pch * 16 + prow] = ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7])) + ((pv[8] + pv[9]) + (pv[10] + pv[11]));
    __syncthreads();
    if (PRO) {
        if (tid < 16) { float a = 0.f; for (int cch = 0; cch < nch; cch++) a += s_pp[cch * 16 + tid]; s_rstd[tid] = 1.0f / sqrtf(a / (float)p.w.K + p.norm_eps); }
        __syncthreads();
    }
    for (int t = wave; t < NTW; t += KS) {      // (KS may be smaller than NTW for short K: a wave then finishes several tiles)
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < KS; w++) {
            const float4 v = *reinterpret_cast<const float4*>(sred + ((size_t)(w * NTW + t) * 64 + lane) * 4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        if (PRO) { const float4 rs = *reinterpret_cast<const float4*>(s_rstd + 4 * g); sum.x *= rs.x; sum.y *= rs.y; sum.z *= rs.z; sum.w *= rs.w; }
        const int n = nbase + t * 16 + li; const bool nok = n < N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        const float vals[4] = {sum.x + bias, sum.y + bias, sum.z + bias, sum.w + bias};
        float xw = 0.f;
        const bool pre = t == wave;                         // operands prefetched at kernel start
        if (EPI == EPI_RESID_XF && nok) { if (pre) xw = pre_xw; else { xw = p.xf_w[n]; if (p.xf_w2) xw *= p.xf_w2[n]; } }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = 4 * g + r;
            float v = vals[r];
            if (EPI == EPI_SWIGLU) {
                const float other = dpp_mov<0xB1>(v);
                if (m < M && nok && !(n & 1)) p.out[(size_t)m * p.out_stride + (n >> 1)] = silu_f(v) * other;
            } else if (EPI == EPI_SWIGLU_XF) {
                // activation k = n/2 of row m straight into the XF planes of the next GEMM (K' = N/2): lanes li = 0,2,..,14 hold
                // k = k0..k0+7; a K-slot pair is (k, k+2), i.e. lanes li and li+4
                const float a = silu_f(v) * dpp_mov<0xB1>(v);
                const float b = __shfl(a, lane + 4, 64);
                const int k = n >> 1;
                if (m < M && nok && !(n & 1) && !(k & 2)) {
                    uint32_t hi, lo; split_pair(a, b, hi, lo);
                    const int K2 = N >> 1, qq = k >> 7, jj = (k >> 5) & 3, e = k & 31, half = e >> 4, gg = (e & 15) >> 2, tt = e & 3;
                    const size_t base = ((size_t)((qq * 4 + jj) * 64 + gg * 16 + m)) * 8 + 2 * half + 4 * (tt & 1);
                    uint16_t* xo = reinterpret_cast<uint16_t*>(p.out);
                    *reinterpret_cast<uint32_t*>(xo + base) = hi; *reinterpret_cast<uint32_t*>(xo + (size_t)(K2 >> 7) * 256 * 8 + base) = lo;
                }
            } else if (EPI == EPI_RESID_XF) {
                // residual stream as f32 + the next RMSNorm folded in: XF planes of v * gamma (K-slot pair (k, k+2) = lanes li, li+2)
                // and this workgroup's partial sum of squares per row
                const bool ok = m < M && nok;
                if (ok) { v = v + (pre ? pre_res[r] : p.resid[(size_t)m * p.resid_stride + n]); p.out[(size_t)m * p.out_stride + n] = v; } else v = 0.f;
                const float ss = row16_sum(v * v);
                if (li == 0) p.ssq_out[(size_t)blockIdx.x * 16 + m] = ss;
                const float a = v * xw, b = dpp_mov<0x4E>(a);
                if (ok && !(li & 2)) {
                    uint32_t hi, lo; split_pair(a, b, hi, lo);
                    const int qq = n >> 7, jj = (n >> 5) & 3, e = n & 31, half = e >> 4, gg = (e & 15) >> 2, tt = e & 3;
                    const size_t base = ((size_t)((qq * 4 + jj) * 64 + gg * 16 + m)) * 8 + 2 * half + 4 * (tt & 1);
                    *reinterpret_cast<uint32_t*>(p.xf_out + base) = hi; *reinterpret_cast<uint32_t*>(p.xf_out + (size_t)(N >> 7) * 256 * 8 + base) = lo;
                }
            } else if (EPI == EPI_ROPE_KV) {
                const float other = dpp_mov<0xB1>(v);          // the pair partner (interleaved RoPE pairs, rope.rs:77-141)
                if (m < M && nok) {
                    const int ps = pre ? pre_pos[r] : p.pos[m], kd = p.n_kv * p.hd;
                    const int cr = pre ? pre_row[r] : (p.kv_row ? p.kv_row[m] : m);      // cache slice of row m
                    if (n < p.n_q + kd) {
                        const int dd = n % p.hd;
                        const size_t ti = (size_t)ps * (p.hd >> 1) + (dd >> 1);
                        const float c = p.rope_cos[ti], sn = p.rope_sin[ti];
                        const float o = (n & 1) ? other * sn + v * c : v * c - other * sn;
                        if (n < p.n_q) p.out[(size_t)m * p.out_stride + n] = o;
                        else p.kc[(size_t)cr * p.kv_seq_stride + (size_t)((n - p.n_q) / p.hd) * p.kv_head_stride + (size_t)ps * p.hd + dd] = o;
                    } else {
                        const int vn = n - p.n_q - kd;
                        p.vc[(size_t)cr * p.kv_seq_stride + (size_t)(vn / p.hd) * p.kv_head_stride + (size_t)ps * p.hd + (vn % p.hd)] = v;
                    }
                }
            } else if (m < M && nok) {
                if (EPI == EPI_RESID) v = v + (pre ? pre_res[r] : p.resid[(size_t)m * p.resid_stride + n]);
                if (EPI == EPI_GELU) v = gelu_f(v);
                p.out[(size_t)m * p.out_stride + n] = v;
            }
        }
    }
    VOX_TL(p.tl_slot, tlw, 3);
}

// ---- WIDE decode step (round 6): 32 / 48 / 64 rows -- MT slot groups of a continuous batch -- as ONE GEMM per operator.  The forked per-group chains
// of q4_skinny_kernel fetch, convert (nibble -> bf16, block scales) and correct (-136 sum x) every weight tile once PER GROUP, and every workgroup pulls its group's whole
// XF block through its L1: four concurrent chains move ~1.3 GB per layer through the L1s for 65 MB of weights and the step cost grows linearly with the groups
// (1.84 / 2.08 / 2.75 / 3.40 ms for 1..4).  Here:
//  * workgroup = 4 waves, wave w owns NTW n-tiles of the workgroup's n-range; all four walk the SAME K slice (blockIdx.y: K steps [z * sps, +sps)), so each K step's
//    activations -- the MT groups' bf16 hi + lo A fragments, MT x 8 KB -- are fetched from L2 ONCE per workgroup into a double-buffered LDS stage (every thread moves
//    2 MT uint4: global -> registers one step ahead -> ds_write_b128, all 1 KB-contiguous per wave: conflict-free) and read back by all waves as MFMA operands;
//  * a wave converts its weight dwords to B fragments once per K step and multiplies them with the MT groups' fragments: same exact-integer arithmetic as the skinny
//    kernel (bf16 128 + q operands, -136 sum x as the first MFMA pair of every (group, block), f16 block scale on the f32 result);
//  * weights come through a 4-deep register ring issued in consumption order (vmcnt retires in order: the A loads of step s + 2 are requested before the weights of
//    step s + 4, see the skinny kernel's notes);
//  * K is split ACROSS workgroups (the N = 3072 operators have 192 n-tiles: split-K inside a workgroup is what makes every workgroup re-read the whole XF block): slice
//    z stores its accumulators as plane z, in accumulator layout ([z][group][n-tile][lane] float4: 1 KB per wave store), and wide_finish_kernel sums the planes in a
//    fixed order (deterministic) and applies the step's epilogue.  One K slice (the lm_head: 8192 n-tiles) stores the logits directly.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16x8 v4_as_bf16x8(u32x4_t v) { union { u32x4_t u; bf16x8 b; } c; c.u = v; return c.b; }
// Run-time step loop, unrolled by two (register buffers with static indices), every load unconditional with a clamped step index: a branch around a load makes hipcc
// drain vmcnt at the join.  (Tried and dropped: the slice's steps as straight-line code per step count -- hipcc hoists the address arithmetic and the loads of all steps
// to the top, 1 - 13 KB of scratch per lane, 10 x slower.)
template <int MT, int NTW, bool DIRECT>
__global__ __launch_bounds__(256, 2) void q4_wide_kernel(const GemmParams p, const int sps, float* __restrict__ planes) {
    constexpr int SPS = 0;
    unsigned long long* tlb = (SPS == 0 && (p.ksplit & 8)) ? reinterpret_cast<unsigned long long*>(const_cast<float*>(p.bias)) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define VOX_WTL(I_) { if (SPS == 0 && tlb && threadIdx.x == 0) tlb[I_] = __builtin_amdgcn_s_memrealtime(); }
    VOX_WTL(0)
    const int abl = p.ksplit;      // measurement only (tools/wide_bench.py, VOX_WIDE_ABL): 1 no MFMA work, 2 no A loads past the prologue, 4 no weight loads past it (results wrong)
    extern __shared__ __attribute__((aligned(16))) u32x4_t wlds[];      // [2 stages][MT][hi, lo][4 j][64 lanes]
    __shared__ float s_rstd[MT * 16]; __shared__ float s_pp[16 * 16];
    __shared__ __attribute__((aligned(16))) float s_cs[2][MT][4][4][4];      // [stage][group][block j][lane group g][4 rows]: the -136 sum(x) correction of every (row, block) of the step
    constexpr int STAGE = MT * 8 * 64, NA = 2 * MT;      // uint4 per stage; uint4 per thread and stage
    const int N = p.w.N, nq = p.w.nb >> 2, n_tiles = (N + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int tile0 = ((int)blockIdx.x * 4 + wave) * NTW, q0 = (int)blockIdx.y * sps;
    const uint4* wq[NTW]; const uint16_t* ws[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const size_t T = (size_t)min(tile0 + t, n_tiles - 1);
        wq[t] = p.w.qt + (T * nq + q0) * 64 + lane; ws[t] = p.w.st + ((T * nq + q0) * 16 + li) * 4;
    }
    // this thread's share of an A stage: uint4 i = tid + 256 u of the stage, line = i >> 6 = (group, hi / lo, j)
    const u32x4_t* asrc[NA];      // (native vector types below: arrays of the uint4 STRUCT with conditional writes end up in scratch)
#pragma unroll
    for (int u = 0; u < NA; u++) {
        const int i = tid + 256 * u, line = i >> 6, mt = line >> 3, hl = (line >> 2) & 1, j = line & 3;
        asrc[u] = reinterpret_cast<const u32x4_t*>(p.xf) + (size_t)mt * p.xf_gstride + (size_t)hl * nq * 256 + ((size_t)q0 * 4 + j) * 64 + (i & 63);
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int WD = MT * NTW >= 16 ? 1 : 2;      // weight register ring: two steps ahead, one at MT 4 x NTW 4 (the registers; a step is >= 2 us there: one step covers an HBM round trip)
    u32x4_t areg[1][NA]; u32x4_t wr[WD][NTW]; u32x2_t sr[WD][NTW];      // A: one step ahead (an L2 / MALL hit: ~1.5 us against a ~2 us step), weights two (generic loop) or four steps
#define VOX_AISSUE(B_, S_) { _Pragma("unroll") for (int u = 0; u < NA; u++) areg[0][u] = asrc[u][(size_t)(S_) * 256]; }
#define VOX_WISSUE(B_, S_) { _Pragma("unroll") for (int t = 0; t < NTW; t++) { wr[B_][t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wq[t] + 64 * (S_))); sr[B_][t] = *reinterpret_cast<const u32x2_t*>(ws[t] + 64 * (S_)); } }
    VOX_AISSUE(0, 0) VOX_WISSUE(0, 0)
    __builtin_amdgcn_sched_barrier(0);
    VOX_WTL(1)
    if (WD == 2) { const int s1 = min(1, sps - 1); VOX_WISSUE(1, s1) }
    __builtin_amdgcn_sched_barrier(0);
    if (DIRECT) {      // fused RMSNorm, consumer side (as q4_skinny_kernel PRO): rstd of every row from the producer's partial sums of squares; requested behind the first operands
        const int prow = tid & 15, pch = tid >> 4;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float a = 0.f;
            for (int pi = pch; pi < p.n_part; pi += 16) a += p.ssq_part[(size_t)mt * p.ssq_part_gstride + (size_t)pi * 16 + prow];
            s_pp[pch * 16 + prow] = a;
            __syncthreads();
            if (tid < 16) { float b = 0.f; for (int c = 0; c < 16; c++) b += s_pp[c * 16 + tid]; s_rstd[mt * 16 + tid] = 1.0f / sqrtf(b / (float)p.w.K + p.norm_eps); }
            __syncthreads();
        }
    }
    const bf16x8 m136 = as_bf16x8(make_uint4(0xC308C308u, 0xC308C308u, 0xC308C308u, 0xC308C308u));
#define VOX_WSTEP(K_, S_)                                                                                          \
    {                                                                                                              \
        u32x4_t* stg = wlds + ((K_) & 1) * STAGE;                                                                    \
        /* the uint4 this thread stages are exactly the A fragments (group u / 2, hi / lo u & 1) of block j = wave for its lane: the correction -136 sum_k x_k of every  \
           (row, block) -- one MFMA pair against a constant B, the same for every n-tile -- is computed ONCE per workgroup here, from registers, and published with the stage */ \
        _Pragma("unroll") for (int mt = 0; mt < MT; mt++) {                                                        \
            f32x4 c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v4_as_bf16x8(areg[0][2 * mt]), m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v4_as_bf16x8(areg[0][2 * mt + 1]), m136, c_, 0, 0, 0); \
            if (li == 0) *reinterpret_cast<f32x4*>(&s_cs[(K_) & 1][mt][wave][g][0]) = c_;                          \
        }                                                                                                          \
        _Pragma("unroll") for (int u = 0; u < NA; u++) stg[tid + 256 * u] = areg[0][u];                     \
        if ((S_) == 0) VOX_WTL(2)                                                                                  \
        __syncthreads();      /* stage (K_ & 1) holds step S_; every wave is done with step S_ - 1 (the other stage) */ \
        if ((S_) == 0) VOX_WTL(3)                                                                                  \
        uint32_t dw[NTW][4]; u32x2_t sv[NTW];                                                                        \
        _Pragma("unroll") for (int t = 0; t < NTW; t++) { dw[t][0] = wr[(K_) % WD][t].x; dw[t][1] = wr[(K_) % WD][t].y; dw[t][2] = wr[(K_) % WD][t].z; dw[t][3] = wr[(K_) % WD][t].w; sv[t] = sr[(K_) % WD][t]; } \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        { const int sa = min((S_) + 1, sps - 1), sn = min((S_) + WD, sps - 1); if (!(abl & 2)) VOX_AISSUE(0, sa) if (!(abl & 4)) VOX_WISSUE((K_) % WD, sn) }      /* in consumption order: A one step ahead, the weights two */ \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        if (!(abl & 1)) {                                                                                          \
            /* groups in sets of MH (all MT, or two at MT 4 x NTW 2, which sits at the 256-register limit): the set's chains are interleaved -- every MFMA's input comes \
               from MH * NTW MFMAs earlier, not from the one before it -- and, where the registers allow (PF), block j + 1's fragments are requested before block j is multiplied */ \
            constexpr int MH = MT * NTW >= 12 ? 1 : (MT * NTW >= 8 ? MT / 2 : MT), NH = MT / MH; constexpr bool PF = NH == 1;      \
            u32x4_t fr[PF ? 2 : 1][MH][2]; f32x4 csr[PF ? 2 : 1][MH];                                              \
            if (PF) { _Pragma("unroll") for (int mt = 0; mt < MH; mt++) { fr[0][mt][0] = stg[((mt * 2 + 0) * 4 + 0) * 64 + lane]; fr[0][mt][1] = stg[((mt * 2 + 1) * 4 + 0) * 64 + lane]; \
                                                                           csr[0][mt] = *reinterpret_cast<const f32x4*>(&s_cs[(K_) & 1][mt][0][g][0]); } } \
            _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                        \
                bf16x8 bw[NTW]; float d[NTW];                                                                      \
                _Pragma("unroll") for (int t = 0; t < NTW; t++) {                                                  \
                    bw[t] = as_bf16x8(q4_dword_to_bf16x8_biased(dw[t][j]));                                        \
                    const uint32_t sc2 = (j >> 1) ? sv[t].y : sv[t].x;                                             \
                    d[t] = f16_bits_to_f32((uint16_t)((j & 1) ? (sc2 >> 16) : (sc2 & 0xFFFFu)));                   \
                }                                                                                                  \
                _Pragma("unroll") for (int h = 0; h < NH; h++) {                                                   \
                    constexpr int dummy_ = 0; (void)dummy_;                                                        \
                    const int cur = PF ? (j & 1) : 0;                                                              \
                    if (!PF) { _Pragma("unroll") for (int mt = 0; mt < MH; mt++) { const int mg_ = h * MH + mt; fr[0][mt][0] = stg[((mg_ * 2 + 0) * 4 + j) * 64 + lane]; fr[0][mt][1] = stg[((mg_ * 2 + 1) * 4 + j) * 64 + lane]; \
                                                                                    csr[0][mt] = *reinterpret_cast<const f32x4*>(&s_cs[(K_) & 1][mg_][j][g][0]); } } \
                    else if (j < 3) { _Pragma("unroll") for (int mt = 0; mt < MH; mt++) { fr[(j + 1) & 1][mt][0] = stg[((mt * 2 + 0) * 4 + j + 1) * 64 + lane]; fr[(j + 1) & 1][mt][1] = stg[((mt * 2 + 1) * 4 + j + 1) * 64 + lane]; \
                                                                                           csr[(j + 1) & 1][mt] = *reinterpret_cast<const f32x4*>(&s_cs[(K_) & 1][mt][j + 1][g][0]); } } \
                    f32x4 tt[MH][NTW];                                                                             \
                    _Pragma("unroll") for (int mt = 0; mt < MH; mt++)                                              \
                        _Pragma("unroll") for (int t = 0; t < NTW; t++) tt[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v4_as_bf16x8(fr[cur][mt][0]), bw[t], csr[cur][mt], 0, 0, 0); \
                    _Pragma("unroll") for (int mt = 0; mt < MH; mt++)                                              \
                        _Pragma("unroll") for (int t = 0; t < NTW; t++) tt[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v4_as_bf16x8(fr[cur][mt][1]), bw[t], tt[mt][t], 0, 0, 0); \
                    _Pragma("unroll") for (int mt = 0; mt < MH; mt++)                                              \
                        _Pragma("unroll") for (int t = 0; t < NTW; t++) acc[h * MH + mt][t] = __builtin_elementwise_fma((f32x4){d[t], d[t], d[t], d[t]}, tt[mt][t], acc[h * MH + mt][t]); \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
    for (int s0 = 0; s0 < sps; s0 += 2) {
        VOX_WSTEP(0, s0)
        if (s0 == 0) VOX_WTL(4)
        if (s0 + 1 < sps) VOX_WSTEP(1, s0 + 1)
        if (s0 == 0) VOX_WTL(5)
    }
    VOX_WTL(6)
#undef VOX_WSTEP
#undef VOX_AISSUE
#undef VOX_WISSUE
    if (DIRECT) {      // one K slice: out[row][n] = rstd[row] * acc (the lm_head's logits)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const float4 rs = *reinterpret_cast<const float4*>(s_rstd + mt * 16 + 4 * g);
            const float rr[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const int n = (tile0 + t) * 16 + li;
                if (tile0 + t < n_tiles && n < N) {
#pragma unroll
                    for (int r = 0; r < 4; r++) p.out[(size_t)(mt * 16 + 4 * g + r) * p.out_stride + n] = acc[mt][t][r] * rr[r];
                }
            }
        }
    } else if (p.wide_rows > 0) {      // the prefill's finishing kernels read [slice][row][N]
        const int z = blockIdx.y, R = p.wide_rows;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int t = 0; t < NTW; t++) {
                const int n = (tile0 + t) * 16 + li;
                if (tile0 + t < n_tiles && n < N) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { const int row = mt * 16 + 4 * g + r; if (row < R) planes[((size_t)z * R + row) * N + n] = acc[mt][t][r]; }
                }
            }
    } else {
        const int z = blockIdx.y;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int t = 0; t < NTW; t++)
                if (tile0 + t < n_tiles) {
                    float* dst = planes + ((((size_t)z * MT + mt) * n_tiles + tile0 + t) * 64 + lane) * 4;
                    // WRITE-THROUGH plane stores: plain stores leave 6 - 14 MB dirty in the XCD L2s, which the kernel boundary then writes back before the finishing launch
                    // may start (MI355X_MICROARCH.md: boundary + B / 6 TB/s; publish-large: write-through wins) -- measured 102.5 -> 96.3 us per layer of GEMM + finishing
                    // launches at four groups (VOX_WIDE_ABL=16: plain stores)
                    if (!(p.ksplit & 16)) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(acc[mt][t]) : "memory");
                    else *reinterpret_cast<float4*>(dst) = make_float4(acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]);
                }
    }
    VOX_WTL(7)
#undef VOX_WTL
}
// finishing launch of the wide step: wave = one (group, n-tile): the K-slice planes summed in slice order (all loads in flight), then q4_skinny_kernel's epilogue for
// that tile, on the group's rows / XF planes / partial sums of squares.  grid (ceil(n_tiles / 4), MT).
template <int EPI, int PRO>
__global__ __launch_bounds__(256) void wide_finish_kernel(const GemmParams p, const float* __restrict__ planes, const int KZ) {
    __shared__ float s_rstd[16]; __shared__ float s_pp[16 * 16];
    const int N = p.w.N, n_tiles = (N + 15) >> 4, MT = gridDim.y, mt = blockIdx.y, M = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int T = min((int)blockIdx.x * 4 + wave, n_tiles - 1); const bool tile_ok = (int)blockIdx.x * 4 + wave < n_tiles;
    const float* src = planes + (((size_t)mt * n_tiles + T) * 64 + lane) * 4; const size_t zstride = (size_t)MT * n_tiles * 256;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z0 = 0; z0 < KZ; z0 += 8) {      // eight plane loads in flight (a run-time trip count alone made the compiler issue load - wait - add)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4*>(src + (size_t)min(z0 + u, KZ - 1) * zstride);
#pragma unroll
        for (int u = 0; u < 8; u++) if (z0 + u < KZ) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
    // the group's operands: rows 16 mt .. 16 mt + 15 of the row-major buffers, its own XF planes and partial sums
    const int row0 = mt * 16;
    if (PRO) {
        const int prow = tid & 15, pch = tid >> 4; float a = 0.f;
        const float* sp = p.ssq_part + (size_t)mt * p.ssq_part_gstride + prow;
        for (int p0 = 0; p0 < p.n_part; p0 += 16 * 12) {      // twelve partials per thread in flight (unconditional clamped loads)
            float pv[12];
#pragma unroll
            for (int u = 0; u < 12; u++) { const int pi = p0 + pch + 16 * u; const float v = sp[(size_t)min(pi, p.n_part - 1) * 16]; pv[u] = pi < p.n_part ? v : 0.f; }
            a += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7])) + ((pv[8] + pv[9]) + (pv[10] + pv[11]));
        }
        s_pp[pch * 16 + prow] = a;
        __syncthreads();
        if (tid < 16) { float b = 0.f; for (int c = 0; c < 16; c++) b += s_pp[c * 16 + tid]; s_rstd[tid] = 1.0f / sqrtf(b / (float)p.w.K + p.norm_eps); }
        __syncthreads();
        const float4 rs = *reinterpret_cast<const float4*>(s_rstd + 4 * g); sum.x *= rs.x; sum.y *= rs.y; sum.z *= rs.z; sum.w *= rs.w;
    }
    if (!tile_ok) return;      // (after the barriers)
    const int n = T * 16 + li; const bool nok = n < N;
    const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
    const float vals[4] = {sum.x + bias, sum.y + bias, sum.z + bias, sum.w + bias};
    float xw = 0.f;
    if (EPI == EPI_RESID_XF && nok) { xw = p.xf_w[n]; if (p.xf_w2) xw *= p.xf_w2[n]; }
    uint16_t* xf_out = EPI == EPI_RESID_XF ? p.xf_out + (size_t)mt * p.xf_out_gstride : nullptr;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int m = 4 * g + r, mg = row0 + m;      // row inside the group / in the row-major buffers
        float v = vals[r];
        if (EPI == EPI_SWIGLU_XF) {
            const float a = silu_f(v) * dpp_mov<0xB1>(v);
            const float b = __shfl(a, lane + 4, 64);
            const int k = n >> 1;
            if (m < M && nok && !(n & 1) && !(k & 2)) {
                uint32_t hi, lo; split_pair(a, b, hi, lo);
                const int K2 = N >> 1, qq = k >> 7, jj = (k >> 5) & 3, e = k & 31, half = e >> 4, gg = (e & 15) >> 2, tt = e & 3;
                const size_t base = ((size_t)((qq * 4 + jj) * 64 + gg * 16 + m)) * 8 + 2 * half + 4 * (tt & 1);
                uint16_t* xo = reinterpret_cast<uint16_t*>(p.out) + (size_t)mt * p.xf_out_gstride;
                *reinterpret_cast<uint32_t*>(xo + base) = hi; *reinterpret_cast<uint32_t*>(xo + (size_t)(K2 >> 7) * 256 * 8 + base) = lo;
            }
        } else if (EPI == EPI_RESID_XF) {
            const bool ok = m < M && nok;
            if (ok) { v = v + p.resid[(size_t)mg * p.resid_stride + n]; p.out[(size_t)mg * p.out_stride + n] = v; } else v = 0.f;
            const float ss = row16_sum(v * v);
            if (li == 0) p.ssq_out[(size_t)mt * p.ssq_out_gstride + (size_t)T * 16 + m] = ss;
            const float a = v * xw, b = dpp_mov<0x4E>(a);
            if (ok && !(li & 2)) {
                uint32_t hi, lo; split_pair(a, b, hi, lo);
                const int qq = n >> 7, jj = (n >> 5) & 3, e = n & 31, half = e >> 4, gg = (e & 15) >> 2, tt = e & 3;
                const size_t base = ((size_t)((qq * 4 + jj) * 64 + gg * 16 + m)) * 8 + 2 * half + 4 * (tt & 1);
                *reinterpret_cast<uint32_t*>(xf_out + base) = hi; *reinterpret_cast<uint32_t*>(xf_out + (size_t)(N >> 7) * 256 * 8 + base) = lo;
            }
        } else if (EPI == EPI_ROPE_KV) {
            const float other = dpp_mov<0xB1>(v);
            if (m < M && nok) {
                const int ps = p.pos[mg], kd = p.n_kv * p.hd;
                const int cr = p.kv_row ? p.kv_row[mg] : mg;
                if (n < p.n_q + kd) {
                    const int dd = n % p.hd;
                    const size_t ti = (size_t)ps * (p.hd >> 1) + (dd >> 1);
                    const float c = p.rope_cos[ti], sn = p.rope_sin[ti];
                    const float o = (n & 1) ? other * sn + v * c : v * c - other * sn;
                    if (n < p.n_q) p.out[(size_t)mg * p.out_stride + n] = o;
                    else p.kc[(size_t)cr * p.kv_seq_stride + (size_t)((n - p.n_q) / p.hd) * p.kv_head_stride + (size_t)ps * p.hd + dd] = o;
                } else {
                    const int vn = n - p.n_q - kd;
                    p.vc[(size_t)cr * p.kv_seq_stride + (size_t)(vn / p.hd) * p.kv_head_stride + (size_t)ps * p.hd + (vn % p.hd)] = v;
                }
            }
        } else if (m < M && nok) {
            p.out[(size_t)mg * p.out_stride + n] = v;
        }
    }
}
// plan of one wide GEMM: n-tiles per wave, K slices.  Enough workgroups to fill the chip twice over where the shape allows (2 workgroups of 4 waves per CU), slices of
// whole K steps.  VOX_WIDE_FORCE="N:ntw:kz" overrides one weight shape (measurement knob).
bool q4_wide_plan(const Q4W& w, int mt, int epi, WidePlan* pl) {
    if (mt < 2 || mt > 4 || w.fmt != WFMT_Q4_0 || !w.qt || !w.st || w.nb % 4 || w.N % 16) return false;
    if (epi != EPI_STORE && epi != EPI_ROPE_KV && epi != EPI_RESID_XF && epi != EPI_SWIGLU_XF) return false;
    const int nq = w.nb / 4, tiles = w.N / 16;
    int ntw = tiles >= 384 ? 2 : 1, kz = 1;
    if (epi == EPI_STORE && tiles >= 4096 && !knob_str("VOX_WIDE_LM_NTW2")) ntw = 4;      // the lm_head (8192 n-tiles): 512 workgroups = ONE round of resident workgroups instead of two, half the activation re-reads
    if (epi != EPI_STORE) {
        const int ranges = (tiles + 4 * ntw - 1) / (4 * ntw);
        for (int d = 1; d <= nq; d++) if (nq % d == 0) { kz = d; if ((long)ranges * d >= 320) break; }
        if (kz > 24) { for (int d = 24; d >= 1; d--) if (nq % d == 0) { kz = d; break; } }
    }
    if (const char* f = knob_str("VOX_WIDE_FORCE")) { int fn = 0, fw = 0, fk = 0; if (sscanf(f, "%d:%d:%d", &fn, &fw, &fk) == 3 && fn == w.N && (fw == 1 || fw == 2 || (fw == 4 && epi == EPI_STORE)) && fk >= 1 && fk <= 24 && nq % fk == 0 && (epi != EPI_STORE || fk == 1)) { ntw = fw; kz = fk; } }
    pl->ntw = ntw; pl->kz = kz; pl->sps = nq / kz;
    return true;
}
size_t q4_wide_planes_bytes(const Q4W& w, int mt, const WidePlan& pl) { return pl.kz > 1 || true ? (size_t)pl.kz * mt * ((w.N + 15) / 16) * 256 * sizeof(float) : 0; }
template <int MT, int NTW, bool DIRECT>
static hipError_t wide_launch(const GemmParams& p, const WidePlan& pl, hipStream_t s) {
    const int tiles = (p.w.N + 15) / 16;
    dim3 grid((tiles + 4 * NTW - 1) / (4 * NTW), pl.kz);
    const size_t lds = (size_t)2 * MT * 8 * 64 * sizeof(uint4);
    auto kern = q4_wide_kernel<MT, NTW, DIRECT>; static DevOnce done;
    hipError_t e = ensure_dyn_lds(kern, lds, &done); if (e != hipSuccess) return e;
    if (knob_str("VOX_WIDE_DEBUG")) { static bool said = false; if (!said) { said = true; int occ = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 256, lds);
        hipFuncAttributes fa{}; (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)); fprintf(stderr, "[wide] MT %d NTW %d direct %d: grid %u x %u, %d steps per slice, lds %zu + %zu static, %d regs, occupancy %d workgroups / CU\n", MT, NTW, (int)DIRECT, grid.x, grid.y, pl.sps, lds, (size_t)fa.sharedSizeBytes, fa.numRegs, occ); } }
    kern<<<grid, dim3(256), lds, s>>>(p, pl.sps, p.kz_scratch);
    return hipGetLastError();
}
hipError_t launch_q4_wide(const GemmParams& p, int epi_stage, hipStream_t s) {      // epi | 0x100: the GEMM launch only, epi | 0x200: the finishing launch only (measurement hook)
    const int epi = epi_stage & 0xff; const bool only_gemm = (epi_stage & 0x100) != 0, only_finish = (epi_stage & 0x200) != 0;
    WidePlan pl;
    const int MT = p.wide_mt;
    if (!p.xf || p.M != 16 * MT || !q4_wide_plan(p.w, MT, epi, &pl)) return hipErrorInvalidValue;
    const bool direct = epi == EPI_STORE;
    GemmParams pa = p; pa.ksplit = env_int("VOX_WIDE_ABL");
    if (direct) { if (!p.out || !p.ssq_part || p.n_part < 1) return hipErrorInvalidValue; }
    else if (!p.kz_scratch || p.kz_scratch_bytes < q4_wide_planes_bytes(p.w, MT, pl)) return hipErrorInvalidValue;
    hipError_t e = hipErrorInvalidValue;
#define VOX_W(M_, N_) if (MT == M_ && pl.ntw == N_) e = direct ? wide_launch<M_, N_, true>(pa, pl, s) : wide_launch<M_, N_, false>(pa, pl, s);
    if (only_finish) e = hipSuccess; else { VOX_W(2, 1) VOX_W(2, 2) VOX_W(3, 1) VOX_W(3, 2) VOX_W(4, 1) VOX_W(4, 2)
        if (direct && pl.ntw == 4) { if (MT == 2) e = wide_launch<2, 4, true>(pa, pl, s); else if (MT == 3) e = wide_launch<3, 4, true>(pa, pl, s); else if (MT == 4) e = wide_launch<4, 4, true>(pa, pl, s); } }
#undef VOX_W
    if (e != hipSuccess || direct || only_gemm) return e;
    dim3 fgrid(((p.w.N + 15) / 16 + 3) / 4, MT);
    switch (epi) {
    case EPI_ROPE_KV: if (!p.ssq_part || !p.pos || !p.kc || !p.vc) return hipErrorInvalidValue; wide_finish_kernel<EPI_ROPE_KV, 1><<<fgrid, dim3(256), 0, s>>>(p, p.kz_scratch, pl.kz); break;
    case EPI_SWIGLU_XF: if (!p.ssq_part) return hipErrorInvalidValue; wide_finish_kernel<EPI_SWIGLU_XF, 1><<<fgrid, dim3(256), 0, s>>>(p, p.kz_scratch, pl.kz); break;
    case EPI_RESID_XF: if (!p.xf_out || !p.xf_w || !p.ssq_out || !p.resid) return hipErrorInvalidValue; wide_finish_kernel<EPI_RESID_XF, 0><<<fgrid, dim3(256), 0, s>>>(p, p.kz_scratch, pl.kz); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- skinny MFMA GEMM for 17..48 rows (the 38-token decoder prefill, gguf/model.rs:908-923): MT m-tiles of 16 rows share ONE weight fetch.
// The 32 x 128 MFMA kernel walks the weights once per 16-row m-tile (three times for M = 38: 0.5 TB/s, profiles/r01_bench_kernel_stats.csv);
// here a wave owns NTW n-tiles and a contiguous range of 128-wide K steps (split-K over the 4 waves of the workgroup, combined through LDS in
// a fixed order), converts its tile-ordered weight dwords to bf16 B fragments ONCE per K step and reuses them for the MT activation tiles
// (f32 rows read from L2, split hi + lo bf16 in registers).  Same arithmetic as q4_skinny_kernel: exact integer weights through the matrix
// core, f16 block scale applied to the f32 result, -136 * sum(x) through a constant-B MFMA pair.
// XIN: the rows arrive as MT XF tiles (bf16 hi + lo MFMA A-fragments, converted once by xf_rows_kernel): no conversion VALU here.  The f32 form
// spent 680 of its ~750 VALU instructions per K step splitting the same 48 x 128 block in every wave of every workgroup (PMC: VALU-bound,
// profiles/r02_pmc_prefill.txt).
__global__ __launch_bounds__(256) void xf_rows_kernel(const float* __restrict__ x, int x_stride, int M, int K, uint16_t* __restrict__ xf, int n_tiles) {
    const int k4 = K >> 2;
    const long total = (long)n_tiles * 16 * k4, tile_stride = (long)2 * (K >> 7) * 256 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / k4), k = (int)(i % k4) * 4;
        const float4 v = row < M ? *reinterpret_cast<const float4*>(x + (size_t)row * x_stride + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        xf_store4(xf + (size_t)(row >> 4) * tile_stride, K, row & 15, k, v);
    }
}
template <int MT, int NTW, int EPI, bool XIN = false>
__global__ __launch_bounds__(256) void q4_skinny_mt_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float sred[];      // [KS 4][MT][NTW][64][4]
    const int nb = p.w.nb, N = p.w.N, M = p.M, nq = nb >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int KS = 4;
    const int g = lane >> 4, li = lane & 15;
    const int nbase = blockIdx.x * (16 * NTW), n_tiles = (N + 15) >> 4;
    const float* xrow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) xrow[mt] = p.x + (size_t)min(mt * 16 + li, M - 1) * p.x_stride;
    const uint4* wq[NTW]; const uint16_t* ws[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const size_t T = (size_t)min(blockIdx.x * NTW + t, n_tiles - 1);
        wq[t] = p.w.qt + T * nq * 64 + lane; ws[t] = p.w.st + (T * nq * 16 + li) * 4;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int per = (nq + KS - 1) / KS, q0 = wave * per, q1 = min(q0 + per, nq);
    const bf16x8 m136 = as_bf16x8(make_uint4(0xC308C308u, 0xC308C308u, 0xC308C308u, 0xC308C308u));
    uint4 wv[NTW]; uint2 sv[NTW];
    if (q0 < q1) {
#pragma unroll
        for (int t = 0; t < NTW; t++) { wv[t] = ld_nt_u4(wq[t] + 64 * q0); sv[t] = *reinterpret_cast<const uint2*>(ws[t] + 64 * q0); }
    }
    for (int q = q0; q < q1; q++) {
        uint4 wn[NTW]; uint2 sn[NTW];
        const int qn = min(q + 1, q1 - 1);
#pragma unroll
        for (int t = 0; t < NTW; t++) { wn[t] = ld_nt_u4(wq[t] + 64 * qn); sn[t] = *reinterpret_cast<const uint2*>(ws[t] + 64 * qn); }   // next step's weights in flight first
        // B fragments + block scales of this K step, once for all m-tiles
        bf16x8 bw[NTW][4]; float dsc[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const uint32_t dw[4] = {wv[t].x, wv[t].y, wv[t].z, wv[t].w}; const uint32_t sc2[2] = {sv[t].x, sv[t].y};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bw[t][j] = as_bf16x8(q4_dword_to_bf16x8_biased(dw[j]));
                dsc[t][j] = f16_bits_to_f32((uint16_t)((j & 1) ? (sc2[j >> 1] >> 16) : (sc2[j >> 1] & 0xFFFFu)));
            }
        }
        // (requesting tile mt + 1's rows before tile mt is multiplied -- a register ping-pong -- measured SLOWER: 7.1 vs 4.8 ms per prefill)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            float4 xa[4], xb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (XIN) {
                    const uint4* xt = p.xf + (size_t)mt * 2 * nq * 256 + lane;
                    xa[j] = *reinterpret_cast<const float4*>(xt + (q * 4 + j) * 64);                       // hi fragments (bit patterns)
                    xb[j] = *reinterpret_cast<const float4*>(xt + (size_t)nq * 256 + (q * 4 + j) * 64);    // lo fragments
                } else {
                    xa[j] = *reinterpret_cast<const float4*>(xrow[mt] + 128 * q + 32 * j + 4 * g);
                    xb[j] = *reinterpret_cast<const float4*>(xrow[mt] + 128 * q + 32 * j + 16 + 4 * g);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint4 ah, al;
                if (XIN) {
                    ah = make_uint4(__float_as_uint(xa[j].x), __float_as_uint(xa[j].y), __float_as_uint(xa[j].z), __float_as_uint(xa[j].w));
                    al = make_uint4(__float_as_uint(xb[j].x), __float_as_uint(xb[j].y), __float_as_uint(xb[j].z), __float_as_uint(xb[j].w));
                } else {
                    split_pair(xa[j].x, xa[j].z, ah.x, al.x); split_pair(xb[j].x, xb[j].z, ah.y, al.y);
                    split_pair(xa[j].y, xa[j].w, ah.z, al.z); split_pair(xb[j].y, xb[j].w, ah.w, al.w);
                }
                f32x4 cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), m136, cs, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(ah), bw[t][j], cs, 0, 0, 0);
                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(al), bw[t][j], tt, 0, 0, 0);
                    const float d = dsc[t][j];
                    acc[mt][t] = __builtin_elementwise_fma((f32x4){d, d, d, d}, tt, acc[mt][t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NTW; t++) { wv[t] = wn[t]; sv[t] = sn[t]; }
    }
    // split-K combine (fixed order) + epilogue: wave w finishes (m-tile, n-tile) pairs w, w + 4, ...
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < NTW; t++)
            *reinterpret_cast<float4*>(sred + ((size_t)((wave * MT + mt) * NTW + t) * 64 + lane) * 4) = make_float4(acc[mt][t][0], acc[mt][t][1], acc[mt][t][2], acc[mt][t][3]);
    __syncthreads();
    for (int pr = wave; pr < MT * NTW; pr += KS) {
        const int mt = pr / NTW, t = pr % NTW;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < KS; w++) {
            const float4 v = *reinterpret_cast<const float4*>(sred + ((size_t)((w * MT + mt) * NTW + t) * 64 + lane) * 4);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const int n = nbase + t * 16 + li; const bool nok = n < N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
        const float vals[4] = {sum.x + bias, sum.y + bias, sum.z + bias, sum.w + bias};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = mt * 16 + 4 * g + r;
            float v = vals[r];
            if (EPI == EPI_SWIGLU) {
                const float other = dpp_mov<0xB1>(v);        // lane^1; rows interleaved: even n = gate, odd n = up
                if (m < M && nok && !(n & 1)) p.out[(size_t)m * p.out_stride + (n >> 1)] = silu_f(v) * other;
            } else if (m < M && nok) {
                if (EPI == EPI_RESID) v = v + p.resid[(size_t)m * p.resid_stride + n];
                if (EPI == EPI_GELU) v = gelu_f(v);
                p.out[(size_t)m * p.out_stride + n] = v;
            }
        }
    }
}
// ---- 17..48 rows, TWO-DIMENSIONAL decomposition (round 3): the kernel above gives every workgroup one n-tile and ALL of K, so each of the 192..576 workgroups
// re-reads the whole XF activation block from L2 (221 MB of L2 -> L1 for the 10.6 MB q|k|v, 354 MB for w1|w3: the launches are L2-bandwidth-bound at 12 TB/s,
// profiles/r02_pmc_prefill.txt).  Here a workgroup = 4 * NTW n-tiles (one per wave) x ONE K slice (blockIdx.y of p.ksplit): its four waves walk the SAME K steps, so
// the XF lines they request are the same lines (one L2 read per workgroup and step), and the activation traffic drops by the number of K slices.  Slice z writes its
// partial sums to plane z; splitk_finish_kernel adds the planes in a fixed order and applies the epilogue.
template <int MT, int NTW>
__global__ __launch_bounds__(256) void q4_skinny_mt2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 xlds[];      // [buf 2][fragment MT * 8][64 lanes]: the XF block of one K step, shared by the four waves
    constexpr int NF = MT * 8, FPW = NF / 4;                          // fragments per step (m-tile, block j, hi / lo), per wave
    const int nb = p.w.nb, N = p.w.N, M = p.M, nq = nb >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int n_tiles = (N + 15) >> 4, tile0 = (blockIdx.x * 4 + wave) * NTW;
    const int KZ = p.ksplit > 1 ? p.ksplit : 1, kz = blockIdx.y;
    const int q0 = (int)((long)nq * kz / KZ), q1 = (int)((long)nq * (kz + 1) / KZ);
    const uint4* wq[NTW]; const uint16_t* ws[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const size_t T = (size_t)min(tile0 + t, n_tiles - 1);
        wq[t] = p.w.qt + T * nq * 64 + lane; ws[t] = p.w.st + (T * nq * 16 + li) * 4;
    }
    // this wave's share of a step's XF block: fragments f = wave + 4 i -> (m-tile f / 8, block (f % 8) / 2, hi / lo f % 2)
    const uint4* xsrc[FPW];
#pragma unroll
    for (int i = 0; i < FPW; i++) { const int f = wave + 4 * i, mt = f >> 3, j = (f & 7) >> 1, hl = f & 1; xsrc[i] = p.xf + (size_t)mt * 2 * nq * 256 + (size_t)hl * nq * 256 + j * 64 + lane; }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < NTW; t++) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 m136 = as_bf16x8(make_uint4(0xC308C308u, 0xC308C308u, 0xC308C308u, 0xC308C308u));
    uint4 wv[NTW]; uint2 sv[NTW]; uint4 xr[FPW];
#pragma unroll
    for (int t = 0; t < NTW; t++) { wv[t] = ld_nt_u4(wq[t] + 64 * q0); sv[t] = *reinterpret_cast<const uint2*>(ws[t] + 64 * q0); }
#pragma unroll
    for (int i = 0; i < FPW; i++) xr[i] = xsrc[i][(size_t)q0 * 256];
#pragma unroll
    for (int i = 0; i < FPW; i++) xlds[(wave + 4 * i) * 64 + lane] = xr[i];
    __syncthreads();
    for (int q = q0; q < q1; q++) {
        const int buf = (q - q0) & 1;
        uint4 wn[NTW]; uint2 sn[NTW];
        const int qn = min(q + 1, q1 - 1);
#pragma unroll
        for (int t = 0; t < NTW; t++) { wn[t] = ld_nt_u4(wq[t] + 64 * qn); sn[t] = *reinterpret_cast<const uint2*>(ws[t] + 64 * qn); }   // next step's weights and activations in flight first
#pragma unroll
        for (int i = 0; i < FPW; i++) xr[i] = xsrc[i][(size_t)qn * 256];
        bf16x8 bw[NTW][4]; float dsc[NTW][4];
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const uint32_t dw[4] = {wv[t].x, wv[t].y, wv[t].z, wv[t].w}; const uint32_t sc2[2] = {sv[t].x, sv[t].y};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bw[t][j] = as_bf16x8(q4_dword_to_bf16x8_biased(dw[j]));
                dsc[t][j] = f16_bits_to_f32((uint16_t)((j & 1) ? (sc2[j >> 1] >> 16) : (sc2[j >> 1] & 0xFFFFu)));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bf16x8 ah = as_bf16x8(xlds[(buf * NF + mt * 8 + j * 2) * 64 + lane]), al = as_bf16x8(xlds[(buf * NF + mt * 8 + j * 2 + 1) * 64 + lane]);
                f32x4 cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, m136, cs, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bw[t][j], cs, 0, 0, 0);
                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bw[t][j], tt, 0, 0, 0);
                    const float d = dsc[t][j];
                    acc[mt][t] = __builtin_elementwise_fma((f32x4){d, d, d, d}, tt, acc[mt][t]);
                }
            }
        }
        if (q + 1 < q1) {
#pragma unroll
            for (int i = 0; i < FPW; i++) xlds[((buf ^ 1) * NF + wave + 4 * i) * 64 + lane] = xr[i];
        }
#pragma unroll
        for (int t = 0; t < NTW; t++) { wv[t] = wn[t]; sv[t] = sn[t]; }
        __syncthreads();
    }
    float* plane = p.out + (size_t)kz * M * N;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < NTW; t++) {
            const int n = (tile0 + t) * 16 + li;
            if (tile0 + t < n_tiles && n < N) {
#pragma unroll
                for (int r = 0; r < 4; r++) { const int m = mt * 16 + 4 * g + r; if (m < M) plane[(size_t)m * N + n] = acc[mt][t][r]; }
            }
        }
}
// a += planes[0][i] + planes[1][i] + ... (k ascending: the fixed order every split-K consumer uses).  A `for (k < kz)` loop with a run-time trip count
// issues one load, waits, adds: kz dependent round trips per element (the 11 us of the summing norm, 8.7 us of the encoder's).  Here eight loads are
// in flight per round (unconditional, clamped index; the add is predicated), so the sum costs ONE round trip for kz <= 8.
template <class V>
__device__ __forceinline__ void plane_sum_add(V& a, const V* __restrict__ p0, size_t plane_stride_v, int kz) {
    for (int k0 = 0; k0 < kz; k0 += 8) {
        V t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = p0[(size_t)min(k0 + u, kz - 1) * plane_stride_v];
#pragma unroll
        for (int u = 0; u < 8; u++) if (k0 + u < kz) a = a + t[u];
    }
}
// sum of the K-slice planes (fixed order) + bias, then the GEMM's epilogue
template <int EPI>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ planes, int KZ, int M, int N, const float* __restrict__ bias,
                                                           const float* __restrict__ resid, int resid_stride, float* __restrict__ out, int out_stride) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / N), n = (int)(i % N);
        float v = planes[i];
        if (KZ > 1) plane_sum_add(v, planes + total + i, (size_t)total, KZ - 1);
        if (bias) v += bias[n];
        if (EPI == EPI_SWIGLU) {
            const float other = dpp_mov<0xB1>(v);        // lane ^ 1: rows interleaved, even n = gate, odd n = up (N is even, so a pair never straddles a row)
            if (!(n & 1)) out[(size_t)m * out_stride + (n >> 1)] = silu_f(v) * other;
        } else {
            if (EPI == EPI_RESID) v = v + resid[(size_t)m * resid_stride + n];
            if (EPI == EPI_GELU) v = gelu_f(v);
            out[(size_t)m * out_stride + n] = v;
        }
    }
}
template <int MT, int NTW>
static hipError_t skinny_mt2_launch(const GemmParams& p_in, int epi, int KZ, hipStream_t s) {
    GemmParams p = p_in; p.ksplit = KZ; p.out = p_in.kz_scratch;
    const int n_tiles = (p.w.N + 15) / 16;
    q4_skinny_mt2_kernel<MT, NTW><<<dim3((n_tiles + 4 * NTW - 1) / (4 * NTW), KZ), dim3(256), (size_t)2 * MT * 8 * 64 * 16, s>>>(p);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    const long total = (long)p.M * p.w.N; const int blocks = (int)std::min<long>((total + 255) / 256, 2048);
    switch (epi) {
    case EPI_STORE: splitk_finish_kernel<EPI_STORE><<<blocks, 256, 0, s>>>(p_in.kz_scratch, KZ, p.M, p.w.N, p.bias, nullptr, 0, p_in.out, p_in.out_stride); break;
    case EPI_RESID: splitk_finish_kernel<EPI_RESID><<<blocks, 256, 0, s>>>(p_in.kz_scratch, KZ, p.M, p.w.N, p.bias, p.resid, p.resid_stride, p_in.out, p_in.out_stride); break;
    case EPI_GELU: splitk_finish_kernel<EPI_GELU><<<blocks, 256, 0, s>>>(p_in.kz_scratch, KZ, p.M, p.w.N, p.bias, nullptr, 0, p_in.out, p_in.out_stride); break;
    case EPI_SWIGLU: splitk_finish_kernel<EPI_SWIGLU><<<blocks, 256, 0, s>>>(p_in.kz_scratch, KZ, p.M, p.w.N, p.bias, nullptr, 0, p_in.out, p_in.out_stride); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
template <int MT, int NTW>
static hipError_t skinny_mt_launch(const GemmParams& p, int epi, hipStream_t s) {
    dim3 grid((p.w.N + 16 * NTW - 1) / (16 * NTW));
    const size_t lds = (size_t)4 * MT * NTW * 64 * 4 * sizeof(float);
    if (p.xf) {
        switch (epi) {
        case EPI_STORE: q4_skinny_mt_kernel<MT, NTW, EPI_STORE, true><<<grid, dim3(256), lds, s>>>(p); break;
        case EPI_RESID: q4_skinny_mt_kernel<MT, NTW, EPI_RESID, true><<<grid, dim3(256), lds, s>>>(p); break;
        case EPI_GELU: q4_skinny_mt_kernel<MT, NTW, EPI_GELU, true><<<grid, dim3(256), lds, s>>>(p); break;
        case EPI_SWIGLU: q4_skinny_mt_kernel<MT, NTW, EPI_SWIGLU, true><<<grid, dim3(256), lds, s>>>(p); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (epi) {
    case EPI_STORE: q4_skinny_mt_kernel<MT, NTW, EPI_STORE><<<grid, dim3(256), lds, s>>>(p); break;
    case EPI_RESID: q4_skinny_mt_kernel<MT, NTW, EPI_RESID><<<grid, dim3(256), lds, s>>>(p); break;
    case EPI_GELU: q4_skinny_mt_kernel<MT, NTW, EPI_GELU><<<grid, dim3(256), lds, s>>>(p); break;
    case EPI_SWIGLU: q4_skinny_mt_kernel<MT, NTW, EPI_SWIGLU><<<grid, dim3(256), lds, s>>>(p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// The pieces of the two-dimensional 17..48-row GEMM for callers that fold the finishing sum into their next kernel (the decoder prefill):
// round 6: the prefill's 17..48-row GEMMs run on q4_wide_kernel (the rows' XF tiles staged through LDS once per workgroup, corrections once per workgroup, the tiles'
// MFMA chains interleaved; row-major planes for the same finishing kernels) with its own K split; VOX_PREFILL_WIDE=0: q4_skinny_mt2_kernel (the round-3 form)
static bool prefill_wide_plan(const Q4W& w, int M, WidePlan* pl) {
    const char* e = knob_str("VOX_PREFILL_WIDE");
    return !(e && e[0] == '0') && env_int("VOX_SKINNY_MT2") == 0 && M > 16 && M <= 48 && w.N % 16 == 0 && q4_wide_plan(w, (M + 15) / 16, EPI_ROPE_KV, pl);
}
int q4_skinny_mt2_plan(const Q4W& w, int M) {      // K slices the 2-D kernel would use for this operator, 0 = not applicable
    if (w.fmt != WFMT_Q4_0 || !w.qt || !w.st || w.nb % 4 || w.N % 2 || M <= 16 || M > 48 || env_int("VOX_SKINNY_MT2") < 0) return 0;
    { WidePlan pl; if (prefill_wide_plan(w, M, &pl)) return pl.kz; }
    const int tiles = (w.N + 15) / 16, nq = w.nb / 4, wg1 = (tiles + 3) / 4;
    int KZ = std::min(std::min(8, nq), std::max(1, (384 + wg1 - 1) / wg1));
    { const int e = env_int("VOX_SKINNY_MT2"); if (e > 0) KZ = std::min(e, nq); }
    return KZ;
}
hipError_t launch_xf_rows(const float* x, int x_stride, int M, int K, uint16_t* xf, hipStream_t s) {      // f32 rows -> ceil(M / 16) XF tiles
    const int mt = (M + 15) / 16; const long total = (long)mt * 16 * (K >> 2);
    if (K % 128) return hipErrorInvalidValue;
    xf_rows_kernel<<<dim3((unsigned)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, s>>>(x, x_stride, M, K, xf, mt);
    return hipGetLastError();
}
hipError_t launch_q4_skinny_mt2_planes(const GemmParams& p_in, int KZ, hipStream_t s) {      // p.xf = the input tiles; writes planes [KZ][M][N] to p.kz_scratch, no epilogue
    GemmParams p = p_in; p.ksplit = KZ; p.out = p_in.kz_scratch;
    const int mt = (p.M + 15) / 16, n_tiles = (p.w.N + 15) / 16;
    if (!p.xf || !p.kz_scratch || (size_t)KZ * p.M * p.w.N * 4 > p.kz_scratch_bytes || KZ < 1) return hipErrorInvalidValue;
    { WidePlan pl;
      if (prefill_wide_plan(p.w, p.M, &pl) && pl.kz == KZ) {
          GemmParams q = p_in; q.wide_mt = mt; q.wide_rows = p.M; q.M = 16 * mt; q.xf_gstride = (long)2 * (p.w.nb / 4) * 256;
          return launch_q4_wide(q, EPI_ROPE_KV | 0x100, s);      // the GEMM launch alone: the caller's finishing kernels read the planes
      } }
    if (mt == 2) q4_skinny_mt2_kernel<2, 1><<<dim3((n_tiles + 3) / 4, KZ), dim3(256), (size_t)2 * 2 * 8 * 64 * 16, s>>>(p);
    else if (mt == 3) q4_skinny_mt2_kernel<3, 1><<<dim3((n_tiles + 3) / 4, KZ), dim3(256), (size_t)2 * 3 * 8 * 64 * 16, s>>>(p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
// Finishing steps of the 38-token prefill folded into what used to follow them (one launch instead of two / three; same arithmetic, same order):
//  * w1|w3: sum of the planes -> SiLU(gate) * up -> straight into the XF tiles of w2's input (was: f32 rows, then xf_rows_kernel)
__global__ __launch_bounds__(256) void splitk_finish_swiglu_xf_kernel(const float* __restrict__ planes, int KZ, int M, int N, const float* __restrict__ bias, uint16_t* __restrict__ xf) {
    const int K2 = N >> 1, k4 = K2 >> 2;
    const long total = (long)M * k4; const size_t plane = (size_t)M * N, tile_stride = (size_t)2 * (K2 >> 7) * 256 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / k4), k = (int)(i % k4) * 4, n = 2 * k;
        const float4* p0 = reinterpret_cast<const float4*>(planes + (size_t)row * N + n);
        float4 a = p0[0], b = p0[1];
        if (KZ > 1) { plane_sum_add(a, p0 + (plane >> 2), plane >> 2, KZ - 1); plane_sum_add(b, p0 + (plane >> 2) + 1, plane >> 2, KZ - 1); }
        if (bias) { const float4 ba = *reinterpret_cast<const float4*>(bias + n), bb = *reinterpret_cast<const float4*>(bias + n + 4); a = a + ba; b = b + bb; }
        xf_store4(xf + (size_t)(row >> 4) * tile_stride, K2, row & 15, k, make_float4(silu_f(a.x) * a.y, silu_f(a.z) * a.w, silu_f(b.x) * b.y, silu_f(b.z) * b.w));
    }
}
//  * q|k|v: sum of the planes -> RoPE on the q and k pairs (rope_kernel's formulas) -> q rows to the f32 buffer the attention reads, k / v rows straight into
//    the cache (was: f32 rows, rope_kernel in place, kv_store_kernel).  One sequence; row m sits at position pos_off + m.
__global__ __launch_bounds__(256) void splitk_finish_rope_kv_kernel(const float* __restrict__ planes, int KZ, int M, int N, float* __restrict__ q_out, int q_stride, int n_q, int n_kv, int hd,
                                                                   int pos_off, const float* __restrict__ cos_t, const float* __restrict__ sin_t, float* __restrict__ kc, float* __restrict__ vc,
                                                                   int head_stride) {
    const int n4 = N >> 2, kd = n_kv * hd;
    const long total = (long)M * n4; const size_t plane = (size_t)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i % n4) * 4;
        const float4* p0 = reinterpret_cast<const float4*>(planes + (size_t)m * N + n);
        float4 v = p0[0];
        if (KZ > 1) plane_sum_add(v, p0 + (plane >> 2), plane >> 2, KZ - 1);
        const int pos = pos_off + m;
        if (n < n_q + kd) {
            const size_t ti = (size_t)pos * (hd >> 1) + ((n % hd) >> 1);
            const float c0 = cos_t[ti], s0 = sin_t[ti], c1 = cos_t[ti + 1], s1 = sin_t[ti + 1];
            v = make_float4(v.x * c0 - v.y * s0, v.x * s0 + v.y * c0, v.z * c1 - v.w * s1, v.z * s1 + v.w * c1);
        }
        if (n < n_q) *reinterpret_cast<float4*>(q_out + (size_t)m * q_stride + n) = v;
        else {
            const int cidx = n < n_q + kd ? n - n_q : n - n_q - kd;
            float* dst = (n < n_q + kd ? kc : vc) + (size_t)(cidx / hd) * head_stride + (size_t)pos * hd + (cidx % hd);
            *reinterpret_cast<float4*>(dst) = v;
        }
    }
}
hipError_t launch_splitk_finish_swiglu_xf(const float* planes, int KZ, int M, int N, const float* bias, uint16_t* xf, hipStream_t s) {
    if (KZ < 1 || M < 1 || M > 48 || N % 256) return hipErrorInvalidValue;
    const long total = (long)M * (N >> 3);
    splitk_finish_swiglu_xf_kernel<<<(int)std::min<long>((total + 255) / 256, 2048), 256, 0, s>>>(planes, KZ, M, N, bias, xf);
    return hipGetLastError();
}
hipError_t launch_splitk_finish_rope_kv(const float* planes, int KZ, int M, int N, float* q_out, int q_stride, int n_q, int n_kv, int hd, int pos_off,
                                        const float* cos_t, const float* sin_t, float* kc, float* vc, int head_stride, hipStream_t s) {
    if (KZ < 1 || M < 1 || N % 4 || hd % 4 || n_q % hd || N != n_q + 2 * n_kv * hd) return hipErrorInvalidValue;
    const long total = (long)M * (N >> 2);
    splitk_finish_rope_kv_kernel<<<(int)std::min<long>((total + 255) / 256, 2048), 256, 0, s>>>(planes, KZ, M, N, q_out, q_stride, n_q, n_kv, hd, pos_off, cos_t, sin_t, kc, vc, head_stride);
    return hipGetLastError();
}
hipError_t launch_splitk_finish_resid(const float* planes, int KZ, int M, int N, float* x, int x_stride, hipStream_t s) {      // x += sum of the planes
    const long total = (long)M * N;
    splitk_finish_kernel<EPI_RESID><<<(int)std::min<long>((total + 255) / 256, 2048), 256, 0, s>>>(planes, KZ, M, N, nullptr, x, x_stride, x, x_stride);
    return hipGetLastError();
}
// 17..48 rows, tile-ordered Q4 weights, K % 128 == 0, 16-byte aligned f32 rows
static hipError_t launch_q4_skinny_mt(const GemmParams& p_in, int epi, hipStream_t s) {
    GemmParams p = p_in;
    const int mt = (p.M + 15) / 16, tiles = (p.w.N + 15) / 16;
    if (!p.xf && p.xf_scratch && p.xf_scratch_bytes >= (size_t)mt * p.w.K * 64 && p.w.K % 128 == 0 && !env_int("VOX_SKINNY_MT_NO_XF")) {
        // rows -> XF tiles once per GEMM (a ~3 us launch), then the conversion-free kernel
        const long total = (long)mt * 16 * (p.w.K >> 2);
        xf_rows_kernel<<<dim3((unsigned)std::min<long>((total + 255) / 256, 1024)), dim3(256), 0, s>>>(p.x, p.x_stride, p.M, p.w.K, p.xf_scratch, mt);
        p.xf = reinterpret_cast<const uint4*>(p.xf_scratch);
    }
    int ntw = tiles >= 512 ? 2 : 1;                           // two n-tiles per wave only while >= 256 workgroups remain
    { const int e = env_int("VOX_SKINNY_MT_NTW"); if (e == 1 || e == 2) ntw = e; }
    // two-dimensional form (q4_skinny_mt2_kernel): K slices so that >= 384 workgroups remain; needs the XF tiles and room for the planes.  VOX_SKINNY_MT2=-1: off
    if (p.xf && p.kz_scratch && p.w.N % 2 == 0 && env_int("VOX_SKINNY_MT2") >= 0) {
        if (!env_int("VOX_SKINNY_MT_NTW")) ntw = 1;      // one n-tile per wave here (w1|w3 with two: 3.32 vs 3.23 ms per prefill, profiles/r03_prefill_2d_kernel.txt)
        const int nq = p.w.nb / 4, wg1 = (tiles + 4 * ntw - 1) / (4 * ntw);
        int KZ = std::min(std::min(8, nq), std::max(1, (384 + wg1 - 1) / wg1));
        { const int e = env_int("VOX_SKINNY_MT2"); if (e > 0) KZ = std::min(e, nq); }
        if ((size_t)KZ * p.M * p.w.N * 4 <= p.kz_scratch_bytes) {
#define VOX_MT2(M_, N_) if (mt == M_ && ntw == N_) return skinny_mt2_launch<M_, N_>(p, epi, KZ, s)
            VOX_MT2(2, 1); VOX_MT2(2, 2); VOX_MT2(3, 1); VOX_MT2(3, 2);
#undef VOX_MT2
        }
    }
#define VOX_MTN(M_, N_) if (mt == M_ && ntw == N_) return skinny_mt_launch<M_, N_>(p, epi, s)
    VOX_MTN(2, 1); VOX_MTN(2, 2); VOX_MTN(3, 1); VOX_MTN(3, 2);
#undef VOX_MTN
    return hipErrorInvalidValue;
}

// split_pair with its two subtractions kept SCALAR: packed (v_pk_add_f32) they need (a, b) in a register pair -- for the slotted GEMM's staging that was a round of v_mov
// from freshly loaded registers at the end of every K step, each behind an s_waitcnt on a load issued moments earlier
__device__ __forceinline__ void split_pair_np(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(hi << 16); asm("" : "+v"(ra));
    const float rb = b - __uint_as_float(hi & 0xFFFF0000u);
    lo = cvt_pk_bf16(ra, rb);
}
// acc += d * t as FOUR v_fma_f32 (VOX_GEMM_BIG_PKFMA: as the f32x4 elementwise fma, which hipcc lowers to two v_pk_fma_f32): beside MFMAs a packed f32 VALU instruction
// costs ~22 cycles more than the two plain FMAs it replaces (MI355X_MICROARCH.md, per-instruction constants), and plain v_fma_f32 is 2 cycles per wave on gfx950.  The empty
// asm keeps the SLP vectoriser from re-packing the four scalars.
__device__ __forceinline__ f32x4 scale_fma(float d, f32x4 t, f32x4 acc) {
#ifdef VOX_GEMM_BIG_PKFMA
    return __builtin_elementwise_fma((f32x4){d, d, d, d}, t, acc);
#else
    // (d opaque as an f32 register: with the f16 -> f32 conversion visible hipcc fuses it into v_fma_mix_f32, which measured as the most expensive instruction of the K
    // step -- removing the 256 of them took a third off the kernel, profiles/r05_gemm_big_slots.txt)
    float dd = d; asm("" : "+v"(dd));
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; r++) { float a = fmaf(dd, t[r], acc[r]); asm("" : "+v"(a)); o[r] = a; }
    return o;
#endif
}

typedef __amdgpu_buffer_rsrc_t vsrd_t;
typedef unsigned vu32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned vu32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vsrd_t vmake_srd(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float4 vbuf_f4(vsrd_t sd, unsigned voff, unsigned soff) { const vu32x4_t v = __builtin_bit_cast(vu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)voff, (int)soff, 0)); return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
__device__ __forceinline__ uint4 vbuf_u4(vsrd_t sd, unsigned voff, unsigned soff) { const vu32x4_t v = __builtin_bit_cast(vu32x4_t, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)voff, (int)soff, 0)); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 vbuf_u2(vsrd_t sd, unsigned voff, unsigned soff) { const vu32x2_t v = __builtin_bit_cast(vu32x2_t, __builtin_amdgcn_raw_buffer_load_b64(sd, (int)voff, (int)soff, 0)); return make_uint2(v.x, v.y); }

// ---- large-M MFMA GEMM (batched encoder / long prefill): workgroup tile (64*WGM) x (64*WGN), four waves in a WGM x WGN grid,
// every wave owns a 64 x 64 output block = 4 m-tiles x 4 n-tiles (16 accumulators), K step 128 (four Q4 blocks).
//  * A (activations): f32 rows -> bf16 hi+lo planes in LDS, MFMA-fragment order, written 1 KB-contiguous per wave
//    (conflict-free ds_write_b128 / ds_read_b128); the next K step's rows are already in flight in registers.
//  * B (weights): the tile-ordered copy -- one coalesced dwordx4 per lane per n-tile per K step IS the lane's four B
//    fragments; nibbles -> bf16 by the bit trick 0x4300|q = 128+q, and the -136*sum(x) correction enters as the initial
//    accumulator through one ones-MFMA pair per (m-tile, block), shared by the four n-tiles (uses the same hi+lo values,
//    so the split error cancels exactly).
//  * per (m-tile, n-tile, block): 2 MFMAs + 2 packed FMAs for the f16 block scale.
// ------------------------------------------------------------------------------------------------
template <int WGM, int WGN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void q4_gemm_big_kernel(const GemmParams p) {
    constexpr int BM = 64 * WGM, MTB = BM / 16;                     // rows / m-tiles per workgroup
    constexpr int PAIRS = 4 * MTB, NU = PAIRS / 4;                  // (block j, m-tile i) fragments groups; per-wave share
    extern __shared__ __attribute__((aligned(16))) uint4 blds[];    // [hi/lo][j 4][MTB][64]
    constexpr int PLANE = 4 * MTB * 64;
    const int nb = p.w.nb, N = p.w.N, M = p.M, nq = nb >> 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * (64 * WGN);
    const int n_tiles = (N + 15) >> 4;
#if !defined(VOX_GEMM_BIG_SERIAL)
    // Addressing = buffer loads: ONE 32-bit byte offset per lane for the A rows (wave w stages rows 16 w + li of the workgroup's 64, K slots 32 u + 4 g: pair (j = u, i = w)),
    // one for the weight tiles and one for their scales, the n-tile as a wave-uniform SGPR offset -- 64-bit per-lane pointers (4 rows + 4 tiles + 4 scale rows = 24 VGPRs)
    // left the pipelined loop below 10 VGPRs short.  Out-of-range tiles / K slots read zeros (buffer bounds), the epilogue's guards drop their results.
    static_assert(WGM == 1, "the staging map below is written for 64-row workgroups");
    const vsrd_t xsrd = vmake_srd(p.x, (unsigned)min((size_t)0xFFFFFFF0u, ((size_t)(M - 1) * p.x_stride + (size_t)p.w.K) * 4));
    const vsrd_t qsrd = vmake_srd(p.w.qt, (unsigned)((size_t)n_tiles * nq * 1024)), ssrd = vmake_srd(p.w.st, (unsigned)((size_t)n_tiles * nq * 128));
    const unsigned xo = (unsigned)(((size_t)min(m0 + 16 * wave + li, M - 1) * p.x_stride + 4 * g) * 4);
    const unsigned qo = (unsigned)lane * 16u, so = (unsigned)li * 8u;
    const int tile0 = __builtin_amdgcn_readfirstlane((n0 >> 4) + wn * 4);
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xa[NU], xb[NU]; uint4 bw[4], bwn[4]; uint2 bs[4], bsn[4];
#define VOX_ALOAD(Q_)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < NU; u++) {                                                           \
        xa[u] = vbuf_f4(xsrd, xo + (unsigned)(Q_) * 512u + (unsigned)u * 128u, 0);                             \
        xb[u] = vbuf_f4(xsrd, xo + (unsigned)(Q_) * 512u + (unsigned)u * 128u + 64u, 0); }
#define VOX_BLOAD(W_, S_, Q_)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 4; t++) {                                                            \
        W_[t] = vbuf_u4(qsrd, qo + (unsigned)(Q_) * 1024u, (unsigned)(tile0 + t) * (unsigned)nq * 1024u);      \
        S_[t] = vbuf_u2(ssrd, so + (unsigned)(Q_) * 128u, (unsigned)(tile0 + t) * (unsigned)nq * 128u); }
#else
    // staging role: wave handles pairs {wave + 4u}; lane (g, li) stages row 16*i + li, K slots of group g
    const float* xrow[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
        const int pair = wave + 4 * u, j = pair / MTB, i = pair % MTB;
        xrow[u] = p.x + (size_t)min(m0 + 16 * i + li, M - 1) * p.x_stride + 32 * j + 4 * g;
    }
    // MFMA role
    const uint4* wq[4]; const uint16_t* ws[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const size_t T = (size_t)min((n0 >> 4) + wn * 4 + t, n_tiles - 1);
        wq[t] = p.w.qt + T * nq * 64 + lane; ws[t] = p.w.st + (T * nq * 16 + li) * 4;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xa[NU], xb[NU]; uint4 bw[4], bwn[4]; uint2 bs[4], bsn[4];
#define VOX_ALOAD(Q_)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < NU; u++) {                                                           \
        xa[u] = *reinterpret_cast<const float4*>(xrow[u] + 128 * (Q_));                                        \
        xb[u] = *reinterpret_cast<const float4*>(xrow[u] + 128 * (Q_) + 16); }
#define VOX_BLOAD(W_, S_, Q_)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 4; t++) {                                                            \
        W_[t] = wq[t][(size_t)64 * (Q_)]; S_[t] = *reinterpret_cast<const uint2*>(ws[t] + 64 * (Q_)); }
#endif
    VOX_ALOAD(0)
    VOX_BLOAD(bw, bs, 0)
    // B operand of the correction MFMA: bf16(-136) in every slot (0xC308, exact), so  sum_k x~_k * (-136)  comes out of the matrix core directly
    const bf16x8 m136 = as_bf16x8(make_uint4(0xC308C308u, 0xC308C308u, 0xC308C308u, 0xC308C308u));
#if !defined(VOX_GEMM_BIG_SERIAL) && !defined(VOX_GEMM_BIG_SLOTS) && !defined(VOX_GEMM_BIG_SBUF) && !defined(VOX_ABL_BIG_NOCS) && !defined(VOX_ABL_BIG_NOFMA) && !defined(VOX_GEMM_BIG_OWNCS)
#define VOX_BIG_SHARED_CS 1
    // round 6: the -136 sum(x) correction of a (row, block) is the same for all four waves (they share the workgroup's 64 rows) -- it is computed ONCE per workgroup, by the
    // wave that stages the fragment (what it has just split IS the MFMA A operand of (block u, m-tile wave) for its lane), and published with the planes: 8 correction MFMAs
    // per wave and K step instead of 32 (160 -> 136 MFMAs per step; VOX_GEMM_BIG_OWNCS build: every wave its own, the round-5 form)
    __shared__ __attribute__((aligned(16))) float s_bcs[2][4][MTB][4][4];      // [buffer][block j][m-tile][lane group g][4 rows]
#define VOX_STAGE_CS(BUF_, U_, HI_, LO_) {                                                                     \
        f32x4 c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(HI_), m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
        c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(LO_), m136, c_, 0, 0, 0);                       \
        if (li == 0) *reinterpret_cast<f32x4*>(&s_bcs[(BUF_) == blds ? 0 : 1][U_][wave][g][0]) = c_; }
#else
#define VOX_STAGE_CS(BUF_, U_, HI_, LO_)
#endif
#define VOX_STAGE(BUF_)                                                                                        \
    _Pragma("unroll") for (int u = 0; u < NU; u++) {     /* K-slot order of the bit-trick B fragment: {4g, 4g+2, 16+4g, 16+4g+2, 4g+1, 4g+3, 16+4g+1, 16+4g+3} */ \
        uint4 hi, lo;                                                                                          \
        split_pair(xa[u].x, xa[u].z, hi.x, lo.x); split_pair(xb[u].x, xb[u].z, hi.y, lo.y);                    \
        split_pair(xa[u].y, xa[u].w, hi.z, lo.z); split_pair(xb[u].y, xb[u].w, hi.w, lo.w);                    \
        VOX_STAGE_CS(BUF_, u, hi, lo)                                                                          \
        (BUF_)[(wave + 4 * u) * 64 + lane] = hi; (BUF_)[PLANE + (wave + 4 * u) * 64 + lane] = lo; }
#if defined(VOX_GEMM_BIG_SLOTS)
    // HAND-ORDERED K step (round 5): a wave issues in order, so its VALU work only runs under its own MFMAs if the two alternate in the instruction stream -- PMC on the
    // loop-nest form (profiles/r05_pmc_gemm_big.txt): matrix pipe 44 % busy, VALU 47 % busy, next to no overlap (both workgroups of a CU drift into lock-step: all waves
    // convert, then all multiply).  The K step is 16 groups (block jj, m-tile i) of ten MFMA SLOTS; every slot = one MFMA + the VALU / LDS / VMEM work that is independent
    // of it, fenced by sched_barrier(0) so that hipcc keeps the order:
    //   slots 0-3  hi MFMA of n-tile t                 + the PREVIOUS group's four scale FMAs of n-tile t (its lo MFMA is >= 4 slots old); slot 0: the next group's A fragments (ds_read)
    //   slot  4    correction MFMA a of the NEXT group + one split_pair of the next K step's staging (double-buffered planes)
    //   slots 5, 6 lo MFMA of n-tiles 0, 1             + (last m-tile of a block) the next block's nibbles -> bf16 IN PLACE: the fragment's last reader has just issued
    //   slot  7    correction MFMA b of the next group + the staging's ds_writes and the A loads of the K step after next (their registers are free again)
    //   slots 8, 9 lo MFMA of n-tiles 2, 3             + next block's n-tiles 2, 3
    // Dependent MFMAs are >= 3 slots apart, every VALU reads MFMA results >= 4 slots old: no s_nop padding.
    VOX_STAGE(blds)
    __syncthreads();
    { const int q1 = min(1, nq - 1); VOX_ALOAD(q1) VOX_BLOAD(bwn, bsn, q1) }
    bf16x8 bf[4];
#pragma unroll
    for (int t = 0; t < 4; t++) bf[t] = as_bf16x8(q4_dword_to_bf16x8_biased(bw[t].x));
    f32x4 T[2][4]; bf16x8 AH[2], AL[2]; f32x4 cs[2];
#define VOX_SB() __builtin_amdgcn_sched_barrier(0)
#if defined(VOX_ABL_S_NOMFMA)      /* timing-only ablations of the slotted loop (wrong results) */
#define VOX_MF(A_, B_, C_) (C_)
#else
#define VOX_MF(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_, B_, C_, 0, 0, 0)
#endif
#if defined(VOX_ABL_S_NOFMA)
#define VOX_SF(D_, T_, ACC_) (T_)
#else
#define VOX_SF(D_, T_, ACC_) scale_fma(D_, T_, ACC_)
#endif
#if defined(VOX_ABL_S_NODS)
#define VOX_LDSR(EXPR_) as_bf16x8(make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u))
#else
#define VOX_LDSR(EXPR_) as_bf16x8(EXPR_)
#endif
#define VOX_DSC(JJ_, t_) f16_bits_to_f32((uint16_t)((((JJ_) & 2) ? BS[t_].y : BS[t_].x) >> (((JJ_) & 1) ? 16 : 0)))
    // (the K loop runs two steps per trip with the roles of the two weight-register sets swapped: a rotating copy at the end of a step made hipcc land the next loads in
    // temporaries and wait for ALL outstanding loads before the barrier)
    auto kstep = [&](const int q, uint4 (&BW)[4], uint2 (&BS)[4], uint4 (&BN)[4]) __attribute__((always_inline)) {
        uint4* const cur = blds + (q & 1) * (2 * PLANE);
        uint4* const nxt = blds + ((q + 1) & 1) * (2 * PLANE);
        const int q2 = min(q + 2, nq - 1);
        AH[0] = VOX_LDSR(cur[(wm * 4) * 64 + lane]); AL[0] = VOX_LDSR(cur[PLANE + (wm * 4) * 64 + lane]);
        AH[1] = VOX_LDSR(cur[(wm * 4 + 1) * 64 + lane]);
        cs[0] = VOX_MF(AH[0], m136, ((f32x4){0.f, 0.f, 0.f, 0.f}));
        cs[0] = VOX_MF(AL[0], m136, cs[0]);
        uint4 shi, slo;
        VOX_SB();
#pragma unroll
        for (int gI = 0; gI < 16; gI++) {
            const int jj = gI >> 2, i = gI & 3, P = gI & 1, Q = P ^ 1, jp = (gI - 1) >> 2, ip = (gI - 1) & 3, u = gI >> 2, part = gI & 3;
            // ---- slots 0-3
#pragma unroll
            for (int t = 0; t < 4; t++) {
                T[P][t] = VOX_MF(AH[P], bf[t], cs[P]);
                if (gI > 0) acc[t][ip] = VOX_SF(VOX_DSC(jp, t), T[Q][t], acc[t][ip]);
                if (t == 0 && gI < 15) {      // the next group's lo fragment: its register's last reader (the previous group's lo MFMA of n-tile 3) has just issued; first use 7 slots away
                    const int jn = (gI + 1) >> 2, in_ = (gI + 1) & 3;
                    AL[Q] = VOX_LDSR(cur[PLANE + (jn * MTB + wm * 4 + in_) * 64 + lane]);
                }
                VOX_SB();
            }
            // ---- slot 4
            if (gI < 15) cs[Q] = VOX_MF(AH[Q], m136, ((f32x4){0.f, 0.f, 0.f, 0.f}));
            if (gI < 14) {      // the hi fragment of the group after next, into the register this group's hi MFMAs have just finished with: ten slots ahead of its first use
                const int jn = (gI + 2) >> 2, in_ = (gI + 2) & 3;
                AH[P] = VOX_LDSR(cur[(jn * MTB + wm * 4 + in_) * 64 + lane]);
            }
            if (part == 0) split_pair_np(xa[u].x, xa[u].z, shi.x, slo.x);
            else if (part == 1) split_pair_np(xb[u].x, xb[u].z, shi.y, slo.y);
            else if (part == 2) split_pair_np(xa[u].y, xa[u].w, shi.z, slo.z);
            else split_pair_np(xb[u].y, xb[u].w, shi.w, slo.w);
            VOX_SB();
            // ---- slots 5, 6
#pragma unroll
            for (int t = 0; t < 2; t++) {
                T[P][t] = VOX_MF(AL[P], bf[t], T[P][t]);
                if (i == 3) bf[t] = as_bf16x8(q4_dword_to_bf16x8_biased(jj == 0 ? BW[t].y : jj == 1 ? BW[t].z : jj == 2 ? BW[t].w : BN[t].x));
                VOX_SB();
            }
            // ---- slot 7
            if (gI < 15) cs[Q] = VOX_MF(AL[Q], m136, cs[Q]);
            if (part == 3) {
                nxt[(wave + 4 * u) * 64 + lane] = shi; nxt[PLANE + (wave + 4 * u) * 64 + lane] = slo;
                xa[u] = vbuf_f4(xsrd, xo + (unsigned)q2 * 512u + (unsigned)u * 128u, 0);
                xb[u] = vbuf_f4(xsrd, xo + (unsigned)q2 * 512u + (unsigned)u * 128u + 64u, 0);
            }
            VOX_SB();
            // ---- slots 8, 9
#pragma unroll
            for (int t = 2; t < 4; t++) {
                T[P][t] = VOX_MF(AL[P], bf[t], T[P][t]);
                if (i == 3) bf[t] = as_bf16x8(q4_dword_to_bf16x8_biased(jj == 0 ? BW[t].y : jj == 1 ? BW[t].z : jj == 2 ? BW[t].w : BN[t].x));
                VOX_SB();
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t][3] = VOX_SF(VOX_DSC(3, t), T[1][t], acc[t][3]);      // the last group's scale FMAs
        VOX_BLOAD(BW, BS, q2)                                     // this step's weight registers are free: the step after next lands in them
        __syncthreads();                                          // every wave is done with this step's planes, the next step's are complete
    };
    for (int q = 0; q < nq; q += 2) {
        kstep(q, bw, bs, bwn);
        if (q + 1 < nq) kstep(q + 1, bwn, bsn, bw);
    }
#undef VOX_SB
#undef VOX_DSC
#undef VOX_MF
#undef VOX_SF
#undef VOX_LDSR
#elif !defined(VOX_GEMM_BIG_SBUF)
    // double-buffered A planes (2 x 32 KB per workgroup, two workgroups per CU): K step q + 1 is staged into the other buffer BEFORE step q's MFMAs, one barrier per K step
    VOX_STAGE(blds)
    __syncthreads();
    { const int q1 = min(1, nq - 1); VOX_ALOAD(q1) VOX_BLOAD(bwn, bsn, q1) }
    for (int q = 0; q < nq; q++) {
        uint4* const cur = blds + (q & 1) * (2 * PLANE);
        uint4* const nxt = blds + ((q + 1) & 1) * (2 * PLANE);
        if (q + 1 < nq) { VOX_STAGE(nxt) }                      // (last read in step q - 1, behind that step's barrier)
        if (q > 0) { const int q1 = min(q + 1, nq - 1); VOX_BLOAD(bwn, bsn, q1) }
        { const int q2 = min(q + 2, nq - 1); VOX_ALOAD(q2) }
#define blds cur
#else
    for (int q = 0; q < nq; q++) {
        __syncthreads();                                          // every wave is done reading the previous K step
        VOX_STAGE(blds)
        __syncthreads();
        { const int q1 = min(q + 1, nq - 1); VOX_ALOAD(q1) VOX_BLOAD(bwn, bsn, q1) }      // unconditional (clamped) prefetch
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !defined(VOX_GEMM_BIG_SLOTS)
#if !defined(VOX_GEMM_BIG_SERIAL) && !defined(VOX_ABL_BIG_NOCS) && !defined(VOX_ABL_BIG_NOFMA)
        // SOFTWARE-PIPELINED over the 16 (block j, m-tile i) groups of a K step: a group's ten MFMAs (correction pair, then the four n-tiles' hi MFMAs, then their lo MFMAs --
        // four independent chains, so a lo MFMA finds its hi result ready) are issued BEFORE the previous group's eight scale FMAs.  Written group by group -- MFMA, dependent
        // MFMA, dependent FMA per n-tile -- hipcc kept that order and padded every FMA with s_nop 7: one chain in flight per wave, the matrix pipe 38 % busy (round-1 PMC).
        f32x4 tp[4] = {}; float dp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            bf16x8 bf[4]; float d[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t w_ = jj == 0 ? bw[t].x : jj == 1 ? bw[t].y : jj == 2 ? bw[t].z : bw[t].w;
                bf[t] = as_bf16x8(q4_dword_to_bf16x8_biased(w_));
                const uint32_t pr = (jj & 2) ? bs[t].y : bs[t].x;
                d[t] = f16_bits_to_f32((uint16_t)((jj & 1) ? (pr >> 16) : (pr & 0xFFFFu)));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bf16x8 ah = as_bf16x8(blds[(jj * MTB + wm * 4 + i) * 64 + lane]);
                const bf16x8 al = as_bf16x8(blds[PLANE + (jj * MTB + wm * 4 + i) * 64 + lane]);
#ifdef VOX_BIG_SHARED_CS
                const f32x4 cs = *reinterpret_cast<const f32x4*>(&s_bcs[q & 1][jj][wm * 4 + i][g][0]);
#else
                f32x4 cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, m136, cs, 0, 0, 0);
#endif
                f32x4 tn[4];
#pragma unroll
                for (int t = 0; t < 4; t++) tn[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bf[t], cs, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; t++) tn[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bf[t], tn[t], 0, 0, 0);
                if (jj > 0 || i > 0) {      // the previous group's scale FMAs, under this group's MFMAs
                    const int ip = i > 0 ? i - 1 : 3;
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[t][ip] = scale_fma(dp[t], tp[t], acc[t][ip]);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) { tp[t] = tn[t]; dp[t] = d[t]; }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t][3] = scale_fma(dp[t], tp[t], acc[t][3]);
#else
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bf16x8 bf[4]; float d[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t w_ = j == 0 ? bw[t].x : j == 1 ? bw[t].y : j == 2 ? bw[t].z : bw[t].w;
                bf[t] = as_bf16x8(q4_dword_to_bf16x8_biased(w_));
                const uint32_t pr = (j & 2) ? bs[t].y : bs[t].x;
                d[t] = f16_bits_to_f32((uint16_t)((j & 1) ? (pr >> 16) : (pr & 0xFFFFu)));
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bf16x8 ah = as_bf16x8(blds[(j * MTB + wm * 4 + i) * 64 + lane]);
                const bf16x8 al = as_bf16x8(blds[PLANE + (j * MTB + wm * 4 + i) * 64 + lane]);
#ifdef VOX_ABL_BIG_NOCS      /* measurement build: results wrong, shows what the correction MFMAs cost */
                f32x4 cs = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
                f32x4 cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, m136, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, m136, cs, 0, 0, 0);
#endif
#pragma unroll
                for (int t = 0; t < 4; t++) {
#ifdef VOX_ABL_BIG_NOFMA
                    f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bf[t], acc[t][i], 0, 0, 0);
#else
                    f32x4 tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bf[t], cs, 0, 0, 0);
#endif
                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bf[t], tt, 0, 0, 0);
#ifdef VOX_ABL_BIG_NOFMA     /* measurement build: accumulate in the matrix core, no block scale */
                    acc[t][i] = tt;
#else
                    acc[t][i] = scale_fma(d[t], tt, acc[t][i]);
#endif
                }
            }
        }
#endif
#pragma unroll
        for (int t = 0; t < 4; t++) { bw[t] = bwn[t]; bs[t] = bsn[t]; }
#if !defined(VOX_GEMM_BIG_SBUF)
#undef blds
        __syncthreads();                                          // every wave is done with this step's buffer, the next step's is complete
#endif
    }
#endif      // !VOX_GEMM_BIG_SLOTS
#undef VOX_STAGE
#undef VOX_STAGE_CS
#undef VOX_BIG_SHARED_CS
#undef VOX_ALOAD
#undef VOX_BLOAD
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int n = n0 + (wn * 4 + t) * 16 + li; const bool nok = n < N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + (wm * 4 + i) * 16 + 4 * g + r;
                float v = acc[t][i][r] + bias;
                if (EPI == EPI_SWIGLU) {
                    const float other = dpp_mov<0xB1>(v);        // lane^1; rows interleaved: even n = gate, odd n = up
                    if (m < M && nok && !(n & 1)) p.out[(size_t)m * p.out_stride + (n >> 1)] = silu_f(v) * other;
                } else if (EPI == EPI_ROPE_ROWS) {      // rope_kernel's arithmetic on the pair (n & ~1, n | 1) held by this lane and its neighbour: p0 = fma(xr, c, -(xi s)), p1 = fma(xi, c, xr s)
                    const float other = dpp_mov<0xB1>(v);
                    if (m < M && nok) {
                        if (n < p.n_q) {
                            const int pos = p.pos ? p.pos[m] : (p.rope_seq_rows > 0 ? m % p.rope_seq_rows : m);
                            const size_t ti = (size_t)pos * (p.hd >> 1) + ((n & (p.hd - 1)) >> 1);      // (head_dim is a power of two: checked by the launcher)
                            const float c = p.rope_cos[ti], sn = p.rope_sin[ti];
                            const float os = __fmul_rn(other, sn);
                            v = (n & 1) ? __fmaf_rn(v, c, os) : __fmaf_rn(v, c, -os);
                        }
                        p.out[(size_t)m * p.out_stride + n] = v;
                    }
                } else if (m < M && nok) {
                    if (EPI == EPI_RESID) v = v + p.resid[(size_t)m * p.resid_stride + n];
                    if (EPI == EPI_GELU) v = gelu_f(v);
                    p.out[(size_t)m * p.out_stride + n] = v;
                }
            }
    }
}
template <int WGM, int WGN>
static hipError_t gemm_big_launch(const GemmParams& p, int epi, hipStream_t s) {
    dim3 grid((p.w.N + 64 * WGN - 1) / (64 * WGN), (p.M + 64 * WGM - 1) / (64 * WGM));
#if !defined(VOX_GEMM_BIG_SBUF) || defined(VOX_GEMM_BIG_SLOTS)
    const size_t lds = (size_t)2 * 2 * 4 * (4 * WGM) * 64 * sizeof(uint4);      // two buffers of WGM * 32 KB
#else
    const size_t lds = (size_t)2 * 4 * (4 * WGM) * 64 * sizeof(uint4);      // WGM * 32 KB
#endif
#define VOX_E(E_) case E_: { auto kern = q4_gemm_big_kernel<WGM, WGN, E_>; static DevOnce done;          \
        hipError_t e = ensure_dyn_lds(kern, lds, &done); if (e != hipSuccess) return e;                       \
        kern<<<grid, dim3(256), lds, s>>>(p); break; }
    switch (epi) { VOX_E(EPI_STORE) VOX_E(EPI_RESID) VOX_E(EPI_GELU) VOX_E(EPI_SWIGLU) VOX_E(EPI_ROPE_ROWS) default: return hipErrorInvalidValue; }
#undef VOX_E
    return hipGetLastError();
}

// ---- dense f32-class GEMM for the convolution stem (models/layers/conv.rs:78-83 as an im2col GEMM): weights are f32 in the
// checkpoint, held as two bf16 planes hi + lo ([N][K] row-major, w ~= hi + lo to 2^-17), activations split hi + lo on the way into LDS;
// three MFMAs per product (ah*bh + al*bh + ah*bl) and the accumulation over all of K stays inside the matrix core (no scales, so no
// per-block VALU at all).  Workgroup = 64 rows x (64*NTW) columns, wave = 64 x (16*NTW); K step 128; A rows may overlap (x_stride <
// K: row t of a stride-2 kernel-3 convolution is the contiguous window [2t-1, 2t+1] of the token-major, zero-padded input).
template <int NTW, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void dense2_gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) uint4 dlds[];    // [hi/lo][j 4][4 m-tiles][64]
    constexpr int PLANE = 4 * 4 * 64;
    const int N = p.w.N, K = p.w.K, M = p.M, nq = K >> 7;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * (64 * NTW);
    const float* xrow[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int pair = wave + 4 * u, j = pair >> 2, i = pair & 3;
        xrow[u] = p.x + (size_t)min(m0 + 16 * i + li, M - 1) * p.x_stride + 32 * j + 8 * g;
    }
    const uint4* wh[NTW]; const uint4* wl[NTW];
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const size_t row = (size_t)min(n0 + (wave * NTW + t) * 16 + li, N - 1);
        wh[t] = p.w.qs + row * (K >> 3) + g; wl[t] = reinterpret_cast<const uint4*>(p.w.sc) + row * (K >> 3) + g;
    }
    f32x4 acc[NTW][4];
#pragma unroll
    for (int t = 0; t < NTW; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[t][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 xa[4], xb[4]; uint4 bh[NTW][4], bl[NTW][4], bhn[NTW][4], bln[NTW][4];
#define VOX_ALOAD(Q_)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                            \
        xa[u] = *reinterpret_cast<const float4*>(xrow[u] + 128 * (Q_));                                        \
        xb[u] = *reinterpret_cast<const float4*>(xrow[u] + 128 * (Q_) + 4); }
#define VOX_BLOAD(H_, L_, Q_)                                                                                  \
    _Pragma("unroll") for (int t = 0; t < NTW; t++)                                                            \
        _Pragma("unroll") for (int j = 0; j < 4; j++) { H_[t][j] = wh[t][(Q_) * 16 + j * 4]; L_[t][j] = wl[t][(Q_) * 16 + j * 4]; }
    VOX_ALOAD(0)
    VOX_BLOAD(bh, bl, 0)
    for (int q = 0; q < nq; q++) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint4 hi, lo; split_bf16x8(xa[u], xb[u], hi, lo);
            dlds[(wave + 4 * u) * 64 + lane] = hi; dlds[PLANE + (wave + 4 * u) * 64 + lane] = lo;
        }
        __syncthreads();
        { const int q1 = min(q + 1, nq - 1); VOX_ALOAD(q1) VOX_BLOAD(bhn, bln, q1) }      // unconditional (clamped) prefetch
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bf16x8 ah = as_bf16x8(dlds[(j * 4 + i) * 64 + lane]);
                const bf16x8 al = as_bf16x8(dlds[PLANE + (j * 4 + i) * 64 + lane]);
#pragma unroll
                for (int t = 0; t < NTW; t++) {
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(bh[t][j]), acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, as_bf16x8(bh[t][j]), acc[t][i], 0, 0, 0);
                    acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(bl[t][j]), acc[t][i], 0, 0, 0);
                }
            }
#pragma unroll
        for (int t = 0; t < NTW; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) { bh[t][j] = bhn[t][j]; bl[t][j] = bln[t][j]; }
    }
#undef VOX_ALOAD
#undef VOX_BLOAD
#pragma unroll
    for (int t = 0; t < NTW; t++) {
        const int n = n0 + (wave * NTW + t) * 16 + li; const bool nok = n < N;
        const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + i * 16 + 4 * g + r;
                float v = acc[t][i][r] + bias;
                if (EPI == EPI_SWIGLU) {
                    const float other = dpp_mov<0xB1>(v);        // lane^1 = column n^1; rows interleaved: even n = gate, odd n = up
                    if (m < M && nok && !(n & 1)) p.out[(size_t)m * p.out_stride + (n >> 1)] = silu_f(v) * other;
                } else {
                    if (EPI == EPI_GELU) v = gelu_f(v);
                    if (EPI == EPI_RESID && m < M && nok) v = v + p.resid[(size_t)m * p.resid_stride + n];
                    if (m < M && nok) p.out[(size_t)m * p.out_stride + n] = v;
                }
            }
    }
}
hipError_t launch_dense2_gemm(const GemmParams& p, int epi, hipStream_t s) {
    if (p.w.fmt != WFMT_BF16X2 || p.w.K % 128 || (p.x_stride % 4) || p.M <= 0) return hipErrorInvalidValue;
    // wide form = 64 x 128 workgroups (two n-tiles per wave).  Four n-tiles per wave (64 x 256) kept the current AND the prefetched hi + lo B fragments of four tiles in
    // registers -- 256 VGPRs for B alone: 187-212 spilled VGPRs, ~700 B of scratch per lane (round-4 disassembly); two tiles fit (tools/kernel_resources.py: no scratch).
    const long wg2 = (long)((p.w.N + 127) / 128) * ((p.M + 63) / 64);
    const size_t lds = (size_t)2 * 4 * 4 * 64 * sizeof(uint4);      // 32 KB
    const bool wide = wg2 >= 400;
    dim3 grid(wide ? (p.w.N + 127) / 128 : (p.w.N + 63) / 64, (p.M + 63) / 64);
#define VOX_D2(E_) case E_: if (wide) dense2_gemm_kernel<2, E_><<<grid, dim3(256), lds, s>>>(p); else dense2_gemm_kernel<1, E_><<<grid, dim3(256), lds, s>>>(p); break;
    switch (epi) { VOX_D2(EPI_STORE) VOX_D2(EPI_GELU) VOX_D2(EPI_RESID) VOX_D2(EPI_SWIGLU) default: return hipErrorInvalidValue; }
#undef VOX_D2
    return hipGetLastError();
}
// [R][C] -> [C][R] (mel handed over as [128][T] by the reference's callers -> token-major rows for the im2col view)
__global__ void transpose_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: 32 x 8
    for (int k = ty; k < 32; k += 8) { const int r = r0 + k, c = c0 + tx; tile[k][tx] = (r < R && c < C) ? in[(size_t)r * C + c] : 0.f; }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) { const int c = c0 + k, r = r0 + tx; if (c < C && r < R) out[(size_t)c * R + r] = tile[tx][k]; }
}
hipError_t launch_transpose(const float* in, int R, int C, float* out, hipStream_t s) {
    transpose_kernel<<<dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s>>>(in, R, C, out);
    return hipGetLastError();
}

template <int NTW, int TILED>
static hipError_t skinny_launch_n(const GemmParams& p, int epi, int ks, hipStream_t s) {
    dim3 grid((p.w.N + 16 * NTW - 1) / (16 * NTW));
    const size_t lds = (size_t)ks * NTW * 64 * 4 * sizeof(float);
    if (p.xf) {      // fragment-ordered bf16 hi/lo input (batched decode step); tile-ordered weights only
        if (!TILED) return hipErrorInvalidValue;
        const bool pro = p.ssq_part != nullptr;
        if (pro && (p.n_part < 1 || p.n_part > 12 * (64 * ks / 16))) return hipErrorInvalidValue;     // partials per thread (q4_skinny_kernel PRO)
        // straight-line in-order pipeline when every wave owns exactly STEPS K-steps (the decode-step shapes of the real model); VOX_SKINNY_NO_STEPS=1:
        // the legacy loop (measurement knob)
        const int nq = p.w.nb / 4, per = nq % ks == 0 ? nq / ks : 0;
        if (per && !env_int("VOX_SKINNY_NO_STEPS")) {
#define VOX_ST(N_, E_, P_, S_) if (NTW == N_ && epi == E_ && (int)pro == P_ && per == S_) { \
            if (E_ == EPI_RESID_XF && (!p.xf_out || !p.xf_w || !p.ssq_out)) return hipErrorInvalidValue; \
            q4_skinny_kernel<N_, E_, 1, 1, P_, S_><<<grid, dim3(64 * ks), lds, s>>>(p); return hipGetLastError(); }
            VOX_ST(2, EPI_ROPE_KV, 1, 3) VOX_ST(2, EPI_SWIGLU_XF, 1, 6) VOX_ST(1, EPI_SWIGLU_XF, 1, 6) VOX_ST(1, EPI_RESID_XF, 0, 4) VOX_ST(4, EPI_SWIGLU_XF, 1, 6) VOX_ST(1, EPI_RESID_XF, 0, 9) VOX_ST(4, EPI_STORE, 1, 6)
#undef VOX_ST
        }
        if (epi == EPI_RESID_XF) {
            if (NTW != 1 || pro || !p.xf_out || !p.xf_w || !p.ssq_out) return hipErrorInvalidValue;
            q4_skinny_kernel<1, EPI_RESID_XF, 1, 1, 0><<<grid, dim3(64 * ks), lds, s>>>(p);
        } else if (pro) {
            switch (epi) {
            case EPI_STORE: q4_skinny_kernel<NTW, EPI_STORE, 1, 1, 1><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            case EPI_ROPE_KV: q4_skinny_kernel<NTW, EPI_ROPE_KV, 1, 1, 1><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            case EPI_SWIGLU_XF: q4_skinny_kernel<NTW, EPI_SWIGLU_XF, 1, 1, 1><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            default: return hipErrorInvalidValue;
            }
        } else {
            switch (epi) {
            case EPI_STORE: q4_skinny_kernel<NTW, EPI_STORE, 1, 1, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            case EPI_RESID: q4_skinny_kernel<NTW, EPI_RESID, 1, 1, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            case EPI_SWIGLU_XF: q4_skinny_kernel<NTW, EPI_SWIGLU_XF, 1, 1, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
            default: return hipErrorInvalidValue;
            }
        }
        return hipGetLastError();
    }
    switch (epi) {
    case EPI_STORE: q4_skinny_kernel<NTW, EPI_STORE, TILED, 0, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
    case EPI_RESID: q4_skinny_kernel<NTW, EPI_RESID, TILED, 0, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
    case EPI_GELU: q4_skinny_kernel<NTW, EPI_GELU, TILED, 0, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
    case EPI_SWIGLU: q4_skinny_kernel<NTW, EPI_SWIGLU, TILED, 0, 0><<<grid, dim3(64 * ks), lds, s>>>(p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
int q4_skinny_resid_xf_parts(int N) { return (N + 15) / 16; }     // partial sums of squares written by an EPI_RESID_XF launch
static hipError_t launch_q4_skinny(const GemmParams& p_in, int epi, hipStream_t s) {
    GemmParams p = p_in; p.tl_slot = tl_take_slot(2, epi, p.w.N, p.w.K);
    const int nq = p.w.nb / 4, tiles = (p.w.N + 15) / 16;
    // n-tiles per wave: as many as still leave >= 192 workgroups (N = 3072 has only 192 tiles); split-K over 4 waves, 8 when
    // the grid is small and K is long enough.  VOX_SKINNY_NTW / VOX_SKINNY_KS are measurement knobs.
    int ntw = tiles >= 4 * 192 ? 4 : (tiles >= 2 * 192 ? 2 : 1);      // the x fragments are converted once per wave: more tiles per wave = less VALU
    int ks = nq >= 4 ? 4 : (nq >= 2 ? 2 : 1);
    if (tiles / ntw < 256 && nq >= 16) ks = 8;                        // profiles/r01_skinny_sweep.txt
    { const int e = env_int("VOX_SKINNY_NTW"); if (e == 1 || e == 2 || e == 4) ntw = e; }
    { const char* f = knob_str("VOX_SKINNY_FORCE");      // measurement knob "N:ntw:ks": override for one weight shape only
      if (f) { int fn = 0, fw = 0, fk = 0; if (sscanf(f, "%d:%d:%d", &fn, &fw, &fk) == 3 && fn == p.w.N && (fw == 1 || fw == 2 || fw == 4) && (fk == 1 || fk == 2 || fk == 4 || fk == 8)) { ntw = fw; ks = fk; } } }
    if (epi == EPI_RESID_XF) ntw = 1;       // its partial sums of squares are per workgroup = per 16-column tile
    { const int e = env_int("VOX_SKINNY_KS"); if (e == 1 || e == 2 || e == 4 || e == 8) ks = e; }
    const bool tiled = p.w.qt && p.w.st && !env_int("VOX_SKINNY_NO_TILE");
    // (three n-tiles per wave for w1|w3 -- 768 = 3 x 256 workgroups instead of 576 -- was a round-2 knob: no faster, 15 spilled VGPRs; removed in round 5)
    if (ntw == 4) return tiled ? skinny_launch_n<4, 1>(p, epi, ks, s) : skinny_launch_n<4, 0>(p, epi, ks, s);
    if (ntw == 2) return tiled ? skinny_launch_n<2, 1>(p, epi, ks, s) : skinny_launch_n<2, 0>(p, epi, ks, s);
    return tiled ? skinny_launch_n<1, 1>(p, epi, ks, s) : skinny_launch_n<1, 0>(p, epi, ks, s);
}

template <int MT, int NT, int FMT, int TB = 0>
static hipError_t gemm_launch_mn(const GemmParams& p, int epi, hipStream_t s) {
    if (p.ksplit > 1 && (epi != EPI_STORE || p.ksplit > p.w.nb / 4)) return hipErrorInvalidValue;
    dim3 grid((p.w.N + 64 * NT - 1) / (64 * NT), (p.M + 16 * MT - 1) / (16 * MT), p.ksplit > 1 ? p.ksplit : 1);
    const size_t lds = (size_t)2 * 2 * 4 * MT * 64 * sizeof(uint4);     // MT * 16 KB
#define VOX_E(E_) case E_: { auto kern = q4_gemm_kernel<MT, NT, E_, FMT, TB>; static DevOnce done;      \
        hipError_t e = ensure_dyn_lds(kern, lds, &done); if (e != hipSuccess) return e;                       \
        kern<<<grid, dim3(256), lds, s>>>(p); break; }
    switch (epi) { VOX_E(EPI_STORE) VOX_E(EPI_RESID) VOX_E(EPI_GELU) VOX_E(EPI_SWIGLU) default: return hipErrorInvalidValue; }
#undef VOX_E
    return hipGetLastError();
}
template <int FMT>
static hipError_t gemm_launch_f(const GemmParams& p, int epi, hipStream_t s) {
    if (p.w.nb % 4) {   // K % 128 != 0: K-32 kernel
        dim3 grid((p.w.N + 63) / 64, (p.M + 63) / 64);
        switch (epi) {
        case EPI_STORE: q4_gemm_k32_kernel<EPI_STORE, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_RESID: q4_gemm_k32_kernel<EPI_RESID, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_GELU: q4_gemm_k32_kernel<EPI_GELU, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_SWIGLU: q4_gemm_k32_kernel<EPI_SWIGLU, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    // tile choice from the round-1 sweep (profiles/r01_gemm_sweep.txt): 16-row tiles up to M = 48, 32-row tiles above;
    // two n-tiles per wave only when that still leaves >= 256 workgroups.  VOX_GEMM_MT / VOX_GEMM_NT / VOX_GEMM_K32 are
    // measurement knobs.
    int mt = p.M <= 48 ? 1 : 2;
    int nt = 2;
    auto wgs = [&](int mt_, int nt_) { return (long)((p.w.N + 64 * nt_ - 1) / (64 * nt_)) * ((p.M + 16 * mt_ - 1) / (16 * mt_)); };
    if (wgs(mt, 2) < 256) nt = 1;
    { const int e = env_int("VOX_GEMM_MT"); if (e == 1 || e == 2) mt = e; }
    { const int e = env_int("VOX_GEMM_NT"); if (e == 1 || e == 2) nt = e; }
    if (env_int("VOX_GEMM_K32") || (!env_int("VOX_GEMM_MT") && p.M > 48 && env_int("VOX_GEMM_K32_BIG"))) {
        dim3 grid((p.w.N + 63) / 64, (p.M + 63) / 64);
        switch (epi) {
        case EPI_STORE: q4_gemm_k32_kernel<EPI_STORE, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_RESID: q4_gemm_k32_kernel<EPI_RESID, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_GELU: q4_gemm_k32_kernel<EPI_GELU, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        case EPI_SWIGLU: q4_gemm_k32_kernel<EPI_SWIGLU, FMT><<<grid, dim3(256), 0, s>>>(p); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    // 32-row tiles on the tile-ordered copy (no cross-lane transpose of the weight words): the single-clip encoder's wo / w2 (N = 1280, 800 rows)
    // 8.79 -> 8.33 ms per clip with two n-tiles per wave at 250 workgroups (48-row tiles: 9.2-9.4 ms).  VOX_GEMM_NO_TB=1: the row-plane form.
    if (FMT == WFMT_Q4_0 && p.w.qt && p.w.st && mt == 2 && !env_int("VOX_GEMM_NO_TB")) {
        if (!env_int("VOX_GEMM_NT") && wgs(2, 2) >= 128) nt = 2;      // (threshold 64 / 128 / 200: 8.32 / 8.30 / 8.49 ms)
        return nt == 2 ? gemm_launch_mn<2, 2, WFMT_Q4_0, 1>(p, epi, s) : gemm_launch_mn<2, 1, WFMT_Q4_0, 1>(p, epi, s);
    }
#define VOX_MN(M_, N_) if (mt == M_ && nt == N_) return gemm_launch_mn<M_, N_, FMT>(p, epi, s)
    VOX_MN(1, 1); VOX_MN(1, 2); VOX_MN(2, 1); VOX_MN(2, 2);
#undef VOX_MN
    return hipErrorInvalidValue;
}
hipError_t launch_rope(float* buf, int M, int stride, int n_rot, int hd, int pos_off, const float* cos_t, const float* sin_t, hipStream_t s, int seq_rows, const int* row_pos);
static hipError_t launch_q4_gemm_epi(const GemmParams& p, int epi, hipStream_t s, bool* fused_rope);
hipError_t launch_q4_gemm(const GemmParams& p, int epi, hipStream_t s) {
    if (epi != EPI_ROPE_ROWS) return launch_q4_gemm_epi(p, epi, s, nullptr);
    // the encoder's q|k|v: RoPE in the large-M kernel's epilogue (one launch and one pass over the rows less per layer); every other kernel stores and rope_kernel follows
    if (!p.rope_cos || !p.rope_sin || p.hd <= 0 || p.n_q <= 0 || p.n_q > p.w.N || p.n_q % 2) return hipErrorInvalidValue;
    bool fused = false;
    hipError_t e = launch_q4_gemm_epi(p, EPI_ROPE_ROWS, s, &fused);
    if (e != hipSuccess || fused) return e;
    return launch_rope(p.out, p.M, p.out_stride, p.n_q, p.hd, 0, p.rope_cos, p.rope_sin, s, p.rope_seq_rows, p.pos);
}
static hipError_t launch_q4_gemm_epi(const GemmParams& p, int epi_in, hipStream_t s, bool* fused_rope) {
    int epi = epi_in == EPI_ROPE_ROWS ? EPI_STORE : epi_in;      // (EPI_ROPE_ROWS: only the large-M kernel below takes it; the others store, the caller ropes)
    if (p.w.K % 32 || p.M <= 0) return hipErrorInvalidValue;
    if (p.ksplit > 1 && (p.M <= 48 || p.xf || p.w.nb % 4 || env_int("VOX_GEMM_K32") || p.w.fmt == WFMT_F32)) return hipErrorInvalidValue;      // split-K exists in q4_gemm_kernel only
    if (p.w.fmt == WFMT_F32) {      // true-f32 dense weights: bf16 hi + lo planes on the matrix cores (3 MFMAs per product, f32-class like the conv stem)
        if (p.xf) return hipErrorInvalidValue;
        GemmParams v = p; v.w.fmt = WFMT_BF16X2; v.w.qt = nullptr; v.w.st = nullptr;
        return launch_dense2_gemm(v, epi, s);
    }
    if (p.xf && p.M > 16) return (p.M <= 48 && p.w.fmt == WFMT_Q4_0 && p.w.nb % 4 == 0 && p.w.qt && p.w.st && (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_GELU || epi == EPI_SWIGLU))
                                     ? launch_q4_skinny_mt(p, epi, s) : hipErrorInvalidValue;      // ceil(M/16) XF tiles (launch_rms_norm_xf / xf_rows_kernel)
    if (p.xf) return (p.M <= 16 && p.w.fmt == WFMT_Q4_0 && p.w.nb % 4 == 0 && p.w.qt) ? launch_q4_skinny(p, epi, s) : hipErrorInvalidValue;
    if (p.M <= 16 && p.w.fmt == WFMT_Q4_0 && p.w.nb % 4 == 0 && !env_int("VOX_NO_SKINNY")) return launch_q4_skinny(p, epi, s);
    if (p.M > 16 && p.M <= 48 && p.w.fmt == WFMT_Q4_0 && p.w.qt && p.w.st && p.w.nb % 4 == 0 && (p.x_stride % 4) == 0 && (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_GELU || epi == EPI_SWIGLU) &&
        !env_int("VOX_NO_SKINNY_MT")) {
        // the 38-row decoder prefill: the weights are streamed ONCE for all m-tiles.  VOX_PREFILL_KERNEL: 0 auto, 1 = q4_skinny_mt_kernel (activation rows
        // from L2 per wave), 2 = q4_gemm_kernel<3, NT, ., ., tile-ordered B> (48 x 64NT workgroup tile, activations staged + converted once per workgroup)
        const int pk = env_int("VOX_PREFILL_KERNEL");
        if (pk != 2) return launch_q4_skinny_mt(p, epi, s);      // measured (profiles/r02_prefill_kernels.txt): 5.69 ms (round 1) / 4.84 / 7.06 ms (the 48-row tile leaves 48..288 workgroups)
        const long wg1 = (p.w.N + 63) / 64;
        int nt = wg1 >= 512 ? 2 : 1; { const int e = env_int("VOX_GEMM_NT"); if (e == 1 || e == 2) nt = e; }
        return nt == 2 ? gemm_launch_mn<3, 2, WFMT_Q4_0, 1>(p, epi, s) : gemm_launch_mn<3, 1, WFMT_Q4_0, 1>(p, epi, s);
    }
    if (p.w.fmt == WFMT_Q4_0 && p.w.qt && p.w.st && p.w.nb % 4 == 0 && (p.x_stride % 4) == 0) {
        // large M: 64 x 256 workgroup tiles (64 x 64 per wave) once they fill the chip -- 1.2-1.55x the 32 x 128 kernel
        // (profiles/r01_gemm_sweep.txt).  VOX_GEMM_BIG: 0 auto, 1 force, -1 off (measurement knob)
        const int big = env_int("VOX_GEMM_BIG");
        const long wg14 = (long)((p.w.N + 255) / 256) * ((p.M + 63) / 64);
        int big_min = 200; { const int e = env_int("VOX_GEMM_BIG_MIN_WG"); if (e > 0) big_min = e; }      // measurement knob: workgroups of 64 x 256 from which the big kernel takes over
        // (the big kernel addresses its operands through 32-bit buffer offsets: activations, tiles and scales each below 4 GB -- beyond that the 32 x 128 kernel's 64-bit pointers serve)
        const bool fits32 = ((size_t)(p.M - 1) * p.x_stride + (size_t)p.w.K) * 4 < 0xFFFFFFF0ull && (size_t)((p.w.N + 15) / 16) * (p.w.nb / 4) * 1024 < 0xFFFFFFF0ull;
        // N % 256 == 0: the kernel hands its n-tile index to the buffer loads as the SGPR offset, which the hardware leaves out of the bounds check -- a partial last
        // column tile would read past the tile planes (every Voxtral N is a multiple of 256, other shapes take the 32 x 128 kernel)
        if (p.ksplit <= 1 && fits32 && p.w.N % 256 == 0 && (big == 1 || (big == 0 && wg14 >= big_min))) {
            // RoPE in the epilogue only where the grid is several workgroups per CU deep (stacked encoders): at one clip (240 workgroups) the epilogue's table loads are the
            // tail of every CU's only workgroup -- +11 us per layer measured, against 3 us for rope_kernel
            if (epi_in == EPI_ROPE_ROWS && wg14 >= 1024 && (p.hd & (p.hd - 1)) == 0 && !knob_str("VOX_NO_ROPE_FUSE")) { if (fused_rope) *fused_rope = true; return gemm_big_launch<1, 4>(p, EPI_ROPE_ROWS, s); }
            return gemm_big_launch<1, 4>(p, epi, s);
        }
    }
    return p.w.fmt == WFMT_BF16 ? gemm_launch_f<WFMT_BF16>(p, epi, s) : gemm_launch_f<WFMT_Q4_0>(p, epi, s);
}

// ------------------------------------------------------------------------------------------------
// RMSNorm rows (models/layers/rms_norm.rs:44-46): one wave per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rms_norm_kernel(const float* __restrict__ x, int x_stride, int rows, int dim,
                                                       const float* __restrict__ gamma, const float* __restrict__ mul,
                                                       float eps, float* __restrict__ out, int out_stride) {
    // one wave per row; dim % 4 == 0; the row is held in registers (up to 16 float4 per lane = dim <= 4096) so x is
    // read once, with all loads of the row issued together.
    const int row = min(blockIdx.x * 4 + (threadIdx.x >> 6), rows - 1), lane = threadIdx.x & 63;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_stride);
    const int n4 = dim >> 2;
    float4 v[16];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = lane + 64 * i;
        v[i] = xr[min(c, n4 - 1)];
        if (c < n4) ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    for (int c = lane + 1024; c < n4; c += 64) { const float4 t = xr[c]; ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w); }   // dim > 4096
    ss = wave_sum(ss);
    const float rms = sqrtf(ss / (float)dim + eps);
    float4* o = reinterpret_cast<float4*>(out + (size_t)row * out_stride);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* m4 = reinterpret_cast<const float4*>(mul);
#define VOX_RMS_OUT(T_, C_)                                                                                     \
    { float4 t = (T_); const float4 gm = g4[C_];                                                                \
      t.x = (t.x / rms) * gm.x; t.y = (t.y / rms) * gm.y; t.z = (t.z / rms) * gm.z; t.w = (t.w / rms) * gm.w;   \
      if (mul) { const float4 mm = m4[C_]; t.x *= mm.x; t.y *= mm.y; t.z *= mm.z; t.w *= mm.w; }               \
      o[C_] = t; }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = lane + 64 * i;
        if (c < n4) VOX_RMS_OUT(v[i], c)
    }
    for (int c = lane + 1024; c < n4; c += 64) VOX_RMS_OUT(xr[c], c)
#undef VOX_RMS_OUT
}
// few rows (batched decode: one row per sequence): one 256-thread workgroup per row, every load issued at once
// planes != nullptr: the row first takes the sum of `kz` split-K partial planes ([kz][rows][dim], fixed order) -- the finishing step of q4_skinny_mt2_kernel's
// EPI_RESID GEMMs folded into the norm that follows them -- and is written back to x_rw.
__global__ __launch_bounds__(256) void rms_norm_row_kernel(const float* __restrict__ x, int x_stride, int dim, const float* __restrict__ gamma,
                                                           const float* __restrict__ mul, float eps, float* __restrict__ out, int out_stride,
                                                           uint16_t* __restrict__ xf, const float* __restrict__ planes = nullptr, size_t plane_stride = 0, int kz = 0,
                                                           float* __restrict__ x_rw = nullptr) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, n4 = dim >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * x_stride);
    float4 v[10]; float ss = 0.f;                       // dim <= 10240
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int c = tid + 256 * i;
        v[i] = xr[min(c, n4 - 1)];
        if (planes && c < n4) {
            plane_sum_add(v[i], reinterpret_cast<const float4*>(planes + (size_t)row * dim) + c, plane_stride >> 2, kz);
            reinterpret_cast<float4*>(x_rw + (size_t)row * x_stride)[c] = v[i];
        }
        if (c < n4) ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float rms = sqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)dim + eps);
    float4* o = reinterpret_cast<float4*>(out + (size_t)row * out_stride);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* m4 = reinterpret_cast<const float4*>(mul);
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int c = tid + 256 * i;
        if (c < n4) {
            float4 t = v[i]; const float4 gm = g4[c];
            t.x = (t.x / rms) * gm.x; t.y = (t.y / rms) * gm.y; t.z = (t.z / rms) * gm.z; t.w = (t.w / rms) * gm.w;
            if (mul) { const float4 mm = m4[c]; t.x *= mm.x; t.y *= mm.y; t.z *= mm.z; t.w *= mm.w; }
            if (xf) xf_store4(xf + (size_t)(row >> 4) * ((size_t)2 * (dim >> 7) * 256 * 8), dim, row & 15, 4 * c, t); else o[c] = t;      // XF tile row / 16
        }
    }
}
// x += sum of KS split-K partial planes (fixed order), written back; RMSNorm of the new row -> out.  One wave per row, dim <= 4096.
__global__ __launch_bounds__(256) VOX_NO_PK_F32 void rms_norm_sumk_kernel(float* __restrict__ x, int x_stride, int rows, int dim, const float* __restrict__ part, size_t part_stride, int ks,
                                                            const float* __restrict__ gamma, float eps, float* __restrict__ out, int out_stride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float4* xr = reinterpret_cast<float4*>(x + (size_t)row * x_stride);
    const int n4 = dim >> 2;
    float4 v[16];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = lane + 64 * i;
        if (c < n4) {
            float4 a = xr[c];
            plane_sum_add(a, reinterpret_cast<const float4*>(part + (size_t)row * dim) + c, part_stride >> 2, ks);
            v[i] = a; xr[c] = a;
            ss += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    ss = wave_sum(ss);
    const float rms = sqrtf(ss / (float)dim + eps);      // (x / rms) * gamma, like rms_norm_kernel
    float4* o = reinterpret_cast<float4*>(out + (size_t)row * out_stride);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int c = lane + 64 * i;
        if (c < n4) { const float4 g = g4[c]; o[c] = make_float4((v[i].x / rms) * g.x, (v[i].y / rms) * g.y, (v[i].z / rms) * g.z, (v[i].w / rms) * g.w); }
    }
}
hipError_t launch_rms_norm_sumk(float* x, int x_stride, int rows, int dim, const float* part, size_t part_stride, int ksplit, const float* gamma, float eps, float* out, int out_stride, hipStream_t s) {
    if (dim % 4 || dim > 4096 || rows <= 0 || ksplit < 1) return hipErrorInvalidValue;
    rms_norm_sumk_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, s>>>(x, x_stride, rows, dim, part, part_stride, ksplit, gamma, eps, out, out_stride);
    return hipGetLastError();
}
// same as launch_rms_norm, but the normalised rows (<= 16) go straight into the XF fragment planes of the following batched-decode GEMM
hipError_t launch_rms_norm_xf_sumk(float* x, int x_stride, int rows, int dim, const float* planes, int kz, const float* gamma, const float* mul, float eps,
                                   uint16_t* xf, hipStream_t s) {
    if (rows > 48 || dim > 10240 || dim % 128 || kz < 1 || !planes) return hipErrorInvalidValue;
    rms_norm_row_kernel<<<dim3(rows), dim3(256), 0, s>>>(x, x_stride, dim, gamma, mul, eps, nullptr, 0, xf, planes, (size_t)rows * dim, kz, x);
    return hipGetLastError();
}
hipError_t launch_rms_norm_xf(const float* x, int x_stride, int rows, int dim, const float* gamma, const float* mul, float eps,
                              uint16_t* xf, hipStream_t s) {
    if (rows > 48 || dim > 10240 || dim % 128) return hipErrorInvalidValue;      // up to three XF tiles (the 38-token prefill); rows of the last tile past `rows` keep their old contents
    rms_norm_row_kernel<<<dim3(rows), dim3(256), 0, s>>>(x, x_stride, dim, gamma, mul, eps, nullptr, 0, xf);
    return hipGetLastError();
}
hipError_t launch_rms_norm(const float* x, int x_stride, int rows, int dim, const float* gamma, const float* mul, float eps,
                           float* out, int out_stride, hipStream_t s) {
    if (rows <= 64 && dim <= 10240 && dim >= 1024)
        rms_norm_row_kernel<<<dim3(rows), dim3(256), 0, s>>>(x, x_stride, dim, gamma, mul, eps, out, out_stride, nullptr);
    else
        rms_norm_kernel<<<dim3((rows + 3) / 4), dim3(256), 0, s>>>(x, x_stride, rows, dim, gamma, mul, eps, out, out_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// RoPE (interleaved pairs) + KV store for the multi-row paths
// ------------------------------------------------------------------------------------------------
// (VOX_NO_PK_F32: hipcc's rotation was v_pk_mul_f32 op_sel:[0,1] -> v_pk_fma_f32; next to another stream's MFMA kernel ~2.6 % of the pairs lost their -xi * sin term: two
// contexts on one GPU produced different encoder outputs run to run.  Same arithmetic, scalar: p0 = fma(xr, c, -(xi * sn)), p1 = fma(xi, c, xr * sn).)
__global__ VOX_NO_PK_F32 void rope_kernel(float* __restrict__ buf, int M, int stride, int n_rot, int hd, int pos_off,
                            const float* __restrict__ cos_t, const float* __restrict__ sin_t, int seq_rows, const int* __restrict__ row_pos) {
    const int half_cols = n_rot >> 1;
    const long total = (long)M * half_cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / half_cols), pc = (int)(i % half_cols);
        const int col = pc * 2, j = (col % hd) >> 1;
        const size_t ti = (size_t)(pos_off + (row_pos ? row_pos[m] : seq_rows > 0 ? m % seq_rows : m)) * (hd >> 1) + j;   // stacked sequences restart at 0 (row_pos: packed ragged sequences, position per row)
        const float c = cos_t[ti], sn = sin_t[ti];
        float* p = buf + (size_t)m * stride + col;
        const float xr = p[0], xi = p[1];
        p[0] = xr * c - xi * sn; p[1] = xr * sn + xi * c;
    }
}
hipError_t launch_rope(float* buf, int M, int stride, int n_rot, int hd, int pos_off, const float* cos_t, const float* sin_t,
                       hipStream_t s, int seq_rows, const int* row_pos) {
    const long total = (long)M * (n_rot / 2);
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    rope_kernel<<<dim3(blocks), dim3(256), 0, s>>>(buf, M, stride, n_rot, hd, pos_off, cos_t, sin_t, seq_rows, row_pos);
    return hipGetLastError();
}

__global__ void kv_store_kernel(const float* __restrict__ buf, int M, int stride, int k_col, int n_kv, int hd, int pos_off,
                                float* __restrict__ kc, float* __restrict__ vc, int head_stride, int seq_rows, long kv_seq_stride) {
    const int w = n_kv * hd;
    const long total = (long)M * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / w), c = (int)(i % w), h = c / hd, d = c % hd;
        const int sq = seq_rows > 0 ? m / seq_rows : 0, mm = seq_rows > 0 ? m % seq_rows : m;     // stacked sequences: own cache slice each
        const size_t dst = (size_t)sq * kv_seq_stride + (size_t)h * head_stride + (size_t)(pos_off + mm) * hd + d;
        kc[dst] = buf[(size_t)m * stride + k_col + c];
        vc[dst] = buf[(size_t)m * stride + k_col + w + c];
    }
}
hipError_t launch_kv_store(const float* buf, int M, int stride, int k_col, int n_kv, int hd, int pos_off, float* kc, float* vc,
                           int head_stride, hipStream_t s, int seq_rows, long kv_seq_stride) {
    const long total = (long)M * n_kv * hd;
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    kv_store_kernel<<<dim3(blocks), dim3(256), 0, s>>>(buf, M, stride, k_col, n_kv, hd, pos_off, kc, vc, head_stride, seq_rows, kv_seq_stride);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// causal (+sliding window) attention for q_len > 1 (gguf/model.rs:100-120, masking.rs:9-107)
// flash-style online softmax; K/V tiles of 64 keys through LDS; wave = 16 queries x 4 head-dim quarters
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const AttnParams p) {
    constexpr int DP = HD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;            // [64][HD]
    float* Vs = smem + 64 * HD;  // [64][HD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y, kvh = h / (p.n_heads / p.n_kv_heads);
    const int m0 = blockIdx.x * 64;
    const int qi = lane & 15, part = lane >> 4;
    const int m = m0 + wave * 16 + qi;
    const bool q_ok = m < p.M;
    const int pos = p.offset + (q_ok ? m : (p.M - 1));
    const float scale = 1.0f / sqrtf((float)HD);      // head_dim^-0.5 (gguf/model.rs:65)

    float qv[DP], o[DP];
    {
        const float* qp = p.q + (size_t)(q_ok ? m : 0) * p.q_stride + h * HD + part * DP;
#pragma unroll
        for (int e = 0; e < DP; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + e);
            qv[e] = v.x; qv[e + 1] = v.y; qv[e + 2] = v.z; qv[e + 3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < DP; e++) o[e] = 0.f;
    }
    float mx = -INFINITY, l = 0.f;

    const int last_m = min(m0 + 63, p.M - 1);
    int j_lo = 0;
    if (p.window >= 0) j_lo = max(0, p.offset + m0 - p.window);
    const int j_hi = min(p.kv_len - 1, p.offset + last_m);   // inclusive
    const float* kbase = p.k + (size_t)kvh * p.kv_head_stride;
    const float* vbase = p.v + (size_t)kvh * p.kv_head_stride;

    for (int j0 = (j_lo / 64) * 64; j0 <= j_hi; j0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * (HD / 4); i += 256) {
            const int jj = i / (HD / 4), d4 = i % (HD / 4), j = j0 + jj;
            const int jc = min(j, p.kv_len - 1);     // unconditional (clamped) loads, masked afterwards
            float4 kk = *reinterpret_cast<const float4*>(kbase + (size_t)jc * p.kv_row_stride + d4 * 4);
            float4 vv = *reinterpret_cast<const float4*>(vbase + (size_t)jc * p.kv_row_stride + d4 * 4);
            // keys >= kv_len are masked to -inf below (p = 0), and the clamped rows hold finite data: no zeroing needed
            reinterpret_cast<float4*>(Ks)[i] = kk;
            reinterpret_cast<float4*>(Vs)[i] = vv;
        }
        __syncthreads();
        for (int jj = 0; jj < 64; jj += 4) {
            float sc[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float* kr = Ks + (jj + kk) * HD + part * DP;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < DP; e += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(kr + e);
                    s = fmaf(qv[e], kv.x, s); s = fmaf(qv[e + 1], kv.y, s); s = fmaf(qv[e + 2], kv.z, s); s = fmaf(qv[e + 3], kv.w, s);
                }
                s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
                const int j = j0 + jj + kk;
                bool vis = (j < p.kv_len) && (j <= pos);
                if (p.window >= 0) vis = vis && (pos - j <= p.window);
                sc[kk] = vis ? s * scale : -INFINITY;
            }
            const float mnew = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), mx);
            if (mnew > -INFINITY) {
                const float alpha = expf(mx - mnew);     // mx = -inf -> 0
                float pr[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) pr[kk] = expf(sc[kk] - mnew);
                l = l * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
                for (int e = 0; e < DP; e++) o[e] *= alpha;
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const float* vr = Vs + (jj + kk) * HD + part * DP;
#pragma unroll
                    for (int e = 0; e < DP; e += 4) {
                        const float4 vv = *reinterpret_cast<const float4*>(vr + e);
                        o[e] = fmaf(pr[kk], vv.x, o[e]); o[e + 1] = fmaf(pr[kk], vv.y, o[e + 1]);
                        o[e + 2] = fmaf(pr[kk], vv.z, o[e + 2]); o[e + 3] = fmaf(pr[kk], vv.w, o[e + 3]);
                    }
                }
                mx = mnew;
            }
        }
    }
    if (q_ok) {
        float* op = p.out + (size_t)m * p.out_stride + h * HD + part * DP;
        const float inv = 1.0f / l;
#pragma unroll
        for (int e = 0; e < DP; e += 4)
            *reinterpret_cast<float4*>(op + e) = make_float4(o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv);
    }
}
// ------------------------------------------------------------------------------------------------
// The same attention on the matrix cores.  Workgroup = 64 queries x one head, wave = 16 queries, K/V tiles of 64 keys staged
// through LDS as bf16 hi+lo planes (x ~= hi + lo keeps f32-class accuracy: 3 MFMAs per product, hi*hi + hi*lo + lo*hi).
// Both products are computed TRANSPOSED so that the softmax statistics stay lane-local and P never leaves registers:
//   S^T[key][query] = K . Q^T   (A = K rows from LDS, B = Q fragment held in registers for the whole kernel)
//   O^T[d][query]   = V^T . P^T (A = V^T rows from LDS -- V is transposed while it is staged --, B = P^T = the S^T
//                                accumulator registers themselves: C-layout row 4g+r of key sub-tile t is B-layout k-slot
//                                (g, r) of that sub-tile, so two sub-tiles form one 32-key MFMA step with no shuffle)
// In every fragment the lane's column index is the query (lane & 15): running max, running sum and the rescale factor of
// the online softmax are per-lane scalars; only the tile max needs a cross-lane step (xor 16, 32).
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_prefill_mfma_kernel(const AttnParams p) {
    constexpr int KS = HD / 32, DT = HD / 16, KROW = HD + 8, VROW = 72;
    extern __shared__ __attribute__((aligned(16))) uint16_t smh[];
    uint16_t* Kh = smh;                  // [64 keys][KROW]   bf16 hi
    uint16_t* Kl = Kh + 64 * KROW;       //                   bf16 lo
    uint16_t* Vh = Kl + 64 * KROW;       // [HD][VROW keys]   transposed, bf16 hi
    uint16_t* Vl = Vh + HD * VROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    // XCD-aware order: all query blocks of one (head, sequence) read the same K / V; the hardware deals workgroups round-robin to the
    // 8 XCDs by linear id, so (head, sequence) pairs are dealt to XCDs and a pair's query blocks take consecutive slots of its XCD.
    int bx = blockIdx.x, h = blockIdx.y, sq = blockIdx.z;       // stacked sequences (gridDim.z): sequence z owns rows [z*seq stride ..)
    {
        const int nblk = gridDim.x, pairs = gridDim.y * gridDim.z;
        if (nblk > 1 && pairs % 8 == 0) {
            const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * nblk + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
            const int pair = xcd * (pairs >> 3) + slot / nblk;
            bx = slot % nblk; h = pair % (int)gridDim.y; sq = pair / (int)gridDim.y;
        }
    }
    const int kvh = h / (p.n_heads / p.n_kv_heads);
    const int M = p.seq_len ? p.seq_len[sq] : p.M, kv_len = p.seq_len ? p.offset + M : p.kv_len;
    const int m0 = ((int)gridDim.x - 1 - bx) * 64;                 // causal: late query blocks have the most keys -- dispatch them first
    if (m0 >= M) return;                                           // whole workgroup (ragged batch)
    const int m = m0 + wave * 16 + c;
    const int mq = min(m, M - 1);
    const int pos = p.offset + mq;
    const int wave_pos_hi = p.offset + min(m0 + wave * 16 + 15, M - 1);   // last position any query of this wave has
    const float scale = 1.0f / sqrtf((float)HD);      // head_dim^-0.5 (gguf/model.rs:65)

    bf16x8 qh[KS], ql[KS];
    {
        const float* qp = p.q + (p.seq_row_off ? (size_t)p.seq_row_off[sq] * p.q_stride : (size_t)sq * p.q_seq_stride) + (size_t)mq * p.q_stride + h * HD + 8 * g;      // seq_row_off: ragged sequences packed back to back
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const float4 a = *reinterpret_cast<const float4*>(qp + ks * 32), b = *reinterpret_cast<const float4*>(qp + ks * 32 + 4);
            uint4 hi, lo; split_bf16x8(a, b, hi, lo); qh[ks] = as_bf16x8(hi); ql[ks] = as_bf16x8(lo);
        }
    }
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; dt++) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, lsum = 0.f;

    const int last_m = min(m0 + 63, M - 1);
    int j_lo = 0;
    if (p.window >= 0) j_lo = max(0, p.offset + m0 - p.window);
    const int j_hi = min(kv_len - 1, p.offset + last_m);   // inclusive
    const size_t kvso = p.seq_row_off ? (size_t)p.seq_row_off[sq] * p.kv_row_stride : (size_t)sq * p.kv_seq_stride;
    const float* kbase = p.k + kvso + (size_t)kvh * p.kv_head_stride;
    const float* vbase = p.v + kvso + (size_t)kvh * p.kv_head_stride;

    // register-staged software pipeline: the global loads of tile t+1 are in flight while tile t is being multiplied
    constexpr int NK = (64 * (HD / 4)) / 256, NV = (32 * (HD / 4)) / 256;
    float4 kreg[NK], vra[NV], vrb[NV];
#define VOX_ATT_LOAD(J0_)                                                                                           \
    _Pragma("unroll") for (int u = 0; u < NK; u++) {                                                                \
        const int i = tid + 256 * u, key = i / (HD / 4), d4 = i % (HD / 4);                                         \
        const int jc = min((J0_) + key, kv_len - 1);   /* unconditional clamped loads; masked keys get p = 0 */   \
        kreg[u] = *reinterpret_cast<const float4*>(kbase + (size_t)jc * p.kv_row_stride + d4 * 4);                  \
    }                                                                                                               \
    _Pragma("unroll") for (int u = 0; u < NV; u++) {                                                                \
        const int i = tid + 256 * u, kp = i / (HD / 4), d4 = i % (HD / 4);                                          \
        const int ja = min((J0_) + 2 * kp, kv_len - 1), jb = min((J0_) + 2 * kp + 1, kv_len - 1);               \
        vra[u] = *reinterpret_cast<const float4*>(vbase + (size_t)ja * p.kv_row_stride + d4 * 4);                   \
        vrb[u] = *reinterpret_cast<const float4*>(vbase + (size_t)jb * p.kv_row_stride + d4 * 4);                   \
    }
    const int j_first = (j_lo / 64) * 64, j_last = (j_hi / 64) * 64;
    VOX_ATT_LOAD(j_first)
    for (int j0 = j_first; j0 <= j_hi; j0 += 64) {
        __syncthreads();
        // ---- stage K (row-major) and V (transposed) as bf16 hi/lo planes
#pragma unroll
        for (int u = 0; u < NK; u++) {
            const int i = tid + 256 * u, key = i / (HD / 4), d4 = i % (HD / 4);
            uint2 hi, lo; split_pair(kreg[u].x, kreg[u].y, hi.x, lo.x); split_pair(kreg[u].z, kreg[u].w, hi.y, lo.y);
            *reinterpret_cast<uint2*>(Kh + key * KROW + d4 * 4) = hi;
            *reinterpret_cast<uint2*>(Kl + key * KROW + d4 * 4) = lo;
        }
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const int i = tid + 256 * u, kp = i / (HD / 4), d4 = i % (HD / 4);
            const float4 va = vra[u], vb = vrb[u];
            uint32_t hi, lo;
            split_pair(va.x, vb.x, hi, lo); *reinterpret_cast<uint32_t*>(Vh + (d4 * 4 + 0) * VROW + 2 * kp) = hi; *reinterpret_cast<uint32_t*>(Vl + (d4 * 4 + 0) * VROW + 2 * kp) = lo;
            split_pair(va.y, vb.y, hi, lo); *reinterpret_cast<uint32_t*>(Vh + (d4 * 4 + 1) * VROW + 2 * kp) = hi; *reinterpret_cast<uint32_t*>(Vl + (d4 * 4 + 1) * VROW + 2 * kp) = lo;
            split_pair(va.z, vb.z, hi, lo); *reinterpret_cast<uint32_t*>(Vh + (d4 * 4 + 2) * VROW + 2 * kp) = hi; *reinterpret_cast<uint32_t*>(Vl + (d4 * 4 + 2) * VROW + 2 * kp) = lo;
            split_pair(va.w, vb.w, hi, lo); *reinterpret_cast<uint32_t*>(Vh + (d4 * 4 + 3) * VROW + 2 * kp) = hi; *reinterpret_cast<uint32_t*>(Vl + (d4 * 4 + 3) * VROW + 2 * kp) = lo;
        }
        __syncthreads();
        { const int jn = min(j0 + 64, j_last); VOX_ATT_LOAD(jn) }        // next tile (the last one re-loads itself: unconditional)
        __builtin_amdgcn_sched_barrier(0);
        if (j0 > wave_pos_hi) continue;                       // wave-uniform: every key of this tile is in the future of all 16 queries

        // ---- S^T = K . Q^T for the four 16-key sub-tiles
        f32x4 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
            sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (j0 + kt * 16 <= wave_pos_hi) {                // wave-uniform
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    const bf16x8 ah = as_bf16x8(*reinterpret_cast<const uint4*>(Kh + (kt * 16 + c) * KROW + ks * 32 + 8 * g));
                    const bf16x8 al = as_bf16x8(*reinterpret_cast<const uint4*>(Kl + (kt * 16 + c) * KROW + ks * 32 + 8 * g));
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, qh[ks], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ql[ks], sc[kt], 0, 0, 0);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, qh[ks], sc[kt], 0, 0, 0);
                }
            }
        }
        // ---- mask + online softmax (lane: query c, keys j0 + 16 kt + 4 g + r)
        float mt = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int j = j0 + kt * 16 + 4 * g + r;
                bool vis = (j < kv_len) && (j <= pos);
                if (p.window >= 0) vis = vis && (pos - j <= p.window);
                const float v = vis ? sc[kt][r] * scale : -INFINITY;
                sc[kt][r] = v; mt = fmaxf(mt, v);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64)); mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mnew = fmaxf(mx, mt);
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
        const float alpha = __expf(mx - msafe);               // mx = -inf -> 0
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const float e = __expf(sc[kt][r] - msafe); sc[kt][r] = e; ps += e; }
        lsum = lsum * alpha + ps; mx = mnew;
#pragma unroll
        for (int dt = 0; dt < DT; dt++) { o[dt][0] *= alpha; o[dt][1] *= alpha; o[dt][2] *= alpha; o[dt][3] *= alpha; }
        // ---- O^T += V^T . P^T, 32 keys per MFMA step
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            if (j0 + kk * 32 <= wave_pos_hi) {                // wave-uniform
                uint4 ph, pl;
                split_pair(sc[2 * kk][0], sc[2 * kk][1], ph.x, pl.x); split_pair(sc[2 * kk][2], sc[2 * kk][3], ph.y, pl.y);
                split_pair(sc[2 * kk + 1][0], sc[2 * kk + 1][1], ph.z, pl.z); split_pair(sc[2 * kk + 1][2], sc[2 * kk + 1][3], ph.w, pl.w);
                const bf16x8 bh = as_bf16x8(ph), bl = as_bf16x8(pl);
#pragma unroll
                for (int dt = 0; dt < DT; dt++) {
                    const uint16_t* vh = Vh + (dt * 16 + c) * VROW + kk * 32 + 4 * g;
                    const uint16_t* vl = Vl + (dt * 16 + c) * VROW + kk * 32 + 4 * g;
                    const uint2 h0 = *reinterpret_cast<const uint2*>(vh), h1 = *reinterpret_cast<const uint2*>(vh + 16);
                    const uint2 l0 = *reinterpret_cast<const uint2*>(vl), l1 = *reinterpret_cast<const uint2*>(vl + 16);
                    const bf16x8 ah = as_bf16x8(make_uint4(h0.x, h0.y, h1.x, h1.y)), al = as_bf16x8(make_uint4(l0.x, l0.y, l1.x, l1.y));
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, o[dt], 0, 0, 0);
                }
            }
        }
    }
#undef VOX_ATT_LOAD
    lsum += __shfl_xor(lsum, 16, 64); lsum += __shfl_xor(lsum, 32, 64);
    if (m < M) {
        const float inv = 1.0f / lsum;
        float* op = p.out + (p.seq_row_off ? (size_t)p.seq_row_off[sq] * p.out_stride : (size_t)sq * p.out_seq_stride) + (size_t)m * p.out_stride + h * HD + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; dt++)
            *reinterpret_cast<float4*>(op + dt * 16) = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
    }
}
template <int HD>
static hipError_t attn_prefill_mfma_launch(const AttnParams& p, hipStream_t s, int n_seq) {
    constexpr size_t lds = ((size_t)2 * 64 * (HD + 8) + (size_t)2 * HD * 72) * sizeof(uint16_t);
    auto kern = attn_prefill_mfma_kernel<HD>;
    static DevOnce attr_done;
    hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
    if (e != hipSuccess) return e;
    kern<<<dim3((p.M + 63) / 64, p.n_heads, n_seq), dim3(256), lds, s>>>(p);
    return hipGetLastError();
}
// ---- causal attention of a SHORT sequence from position 0 (<= 48 rows: the 38-token decoder prefill, gguf/model.rs:908-919 -> Q4Attention::forward_with_cache with an
// empty cache, :125-174).  The 64-query MFMA tiles above cost 29 us per layer for 38 x 38 scores per head: one mostly masked tile, staged and converted per workgroup.  Here:
// one workgroup per (query head, sequence), K and V rows of its KV head in LDS as f32 (2 x 25 KB), a wave per query row -- lane j scores key j (128 sequential f32 FMAs over
// padded, conflict-free rows: the reference's own arithmetic, no bf16 anywhere), wave-shuffle softmax, then P . V with 32 lanes x 4 columns on each half of the keys.  The rows
// can leave as XF tiles (the wo GEMM's A fragments): no f32 round trip, no xf_rows launch.
template <int HD>
__global__ __launch_bounds__(256) void attn_prefill_small_kernel(const AttnParams p) {
    constexpr int KROW = HD + 4, MAXR = 48;
    extern __shared__ __attribute__((aligned(16))) float sms[];
    float* Ks = sms; float* Vs = Ks + MAXR * KROW; float* Qs = Vs + MAXR * HD; float* Ps = Qs + 4 * HD;      // [48][HD + 4] | [48][HD] | [4 waves][HD] | [4 waves][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, sq = blockIdx.y, kvh = h / (p.n_heads / p.n_kv_heads);
    const int M = p.seq_len ? p.seq_len[sq] : p.M;
    const int r = 4 * (int)blockIdx.z + wave;                  // workgroup = four consecutive query rows (grid z), wave = one row
    const int n_keys = min(M, 4 * (int)blockIdx.z + 4);        // causal: this workgroup's rows see keys 0 .. its last row
    if (4 * (int)blockIdx.z >= M) {                            // XF tiles only: the rows behind the sequence are written as zeros (the GEMM multiplies whole tiles)
        if (p.out_xf_tiles && lane < 32) xf_store4(p.out_xf_tiles + (size_t)(r >> 4) * p.out_xf_tile_stride, p.n_heads * HD, r & 15, h * HD + 4 * lane, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    const size_t kvso = (size_t)sq * p.kv_seq_stride;
    const float* kbase = p.k + kvso + (size_t)kvh * p.kv_head_stride; const float* vbase = p.v + kvso + (size_t)kvh * p.kv_head_stride;
    for (int i = tid; i < n_keys * (HD / 4); i += 256) {
        const int j = i / (HD / 4), d4 = i % (HD / 4);
        *reinterpret_cast<float4*>(Ks + j * KROW + 4 * d4) = *reinterpret_cast<const float4*>(kbase + (size_t)j * p.kv_row_stride + 4 * d4);
        *reinterpret_cast<float4*>(Vs + j * HD + 4 * d4) = *reinterpret_cast<const float4*>(vbase + (size_t)j * p.kv_row_stride + 4 * d4);
    }
    float* qs = Qs + wave * HD; float* ps = Ps + wave * 64;
    if (r < M) { const float* qp = p.q + (size_t)sq * p.q_seq_stride + (size_t)r * p.q_stride + h * HD; for (int d = lane; d < HD; d += 64) qs[d] = qp[d]; }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)HD);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < M) {
        const float* kr = Ks + min(lane, n_keys - 1) * KROW;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four chains (a single one is 128 dependent FMAs)
#pragma unroll 8
        for (int d = 0; d < HD; d += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + d), qv = *reinterpret_cast<const float4*>(qs + d);
            s0 = fmaf(qv.x, kv.x, s0); s1 = fmaf(qv.y, kv.y, s1); s2 = fmaf(qv.z, kv.z, s2); s3 = fmaf(qv.w, kv.w, s3);
        }
        const bool vis = lane <= r && (p.window < 0 || r - lane <= p.window);
        const float sc = vis ? ((s0 + s1) + (s2 + s3)) * scale : -INFINITY;
        const float mx = wave_max(sc);
        const float e = vis ? expf(sc - mx) : 0.f;
        const float sum = wave_sum(e);
        ps[lane] = e / sum;
        __builtin_amdgcn_wave_barrier();
        const int c = lane & 31, half = lane >> 5;
        for (int j = half; j <= r; j += 2) {
            const float pj = ps[j]; const float4 vv = *reinterpret_cast<const float4*>(Vs + j * HD + 4 * c);
            o.x = fmaf(pj, vv.x, o.x); o.y = fmaf(pj, vv.y, o.y); o.z = fmaf(pj, vv.z, o.z); o.w = fmaf(pj, vv.w, o.w);
        }
        o.x += __shfl_xor(o.x, 32, 64); o.y += __shfl_xor(o.y, 32, 64); o.z += __shfl_xor(o.z, 32, 64); o.w += __shfl_xor(o.w, 32, 64);
    }
    if (lane < 32) {
        const int col = h * HD + 4 * lane;
        if (p.out_xf_tiles) xf_store4(p.out_xf_tiles + (size_t)(r >> 4) * p.out_xf_tile_stride, p.n_heads * HD, r & 15, col, o);      // (rows M .. of a started block: zeros)
        else if (r < M) *reinterpret_cast<float4*>(p.out + (size_t)sq * p.out_seq_stride + (size_t)r * p.out_stride + col) = o;
    }
}
bool attn_prefill_small_ok(const AttnParams& p, int hd, int n_seq) {
    return hd == 128 && p.M >= 1 && p.M <= 48 && p.offset == 0 && p.kv_len == p.M && !p.seq_row_off && (p.window < 0 || p.window >= p.M) && (p.q_stride % 4) == 0 && (p.kv_row_stride % 4) == 0 &&
           n_seq == 1 /* stacked prefills (64 sequences x 10 row blocks x 32 heads of tiny workgroups) measured 1 % slower than the z-stacked MFMA kernel */ &&
           !env_int("VOX_ATTN_NO_SMALL") && !env_int("VOX_ATTN_F32");
}
static hipError_t attn_prefill_small_launch(const AttnParams& p, hipStream_t s, int n_seq) {
    constexpr int HD = 128;
    const size_t lds = (size_t)(48 * (HD + 4) + 48 * HD + 4 * HD + 4 * 64) * sizeof(float);
    auto kern = attn_prefill_small_kernel<HD>; static DevOnce done;
    hipError_t e = ensure_dyn_lds(kern, lds, &done); if (e != hipSuccess) return e;
    const int rows = p.out_xf_tiles ? ((p.M + 15) / 16) * 16 : p.M;
    kern<<<dim3(p.n_heads, n_seq, (rows + 3) / 4), dim3(256), lds, s>>>(p);
    return hipGetLastError();
}
hipError_t launch_attn_prefill(const AttnParams& p, int hd, hipStream_t s, int n_seq) {
    if (attn_prefill_small_ok(p, hd, n_seq)) return attn_prefill_small_launch(p, s, n_seq);
    if (p.out_xf_tiles) return hipErrorInvalidValue;      // (XF output exists in the short-sequence kernel only: callers ask attn_prefill_small_ok first)
    const int f32_only = env_int("VOX_ATTN_F32");               // ablation / cross-check knob: the f32 VALU kernel (read from the knob table per launch: tests toggle it)
    if (!f32_only && (p.q_stride % 4) == 0 && (p.kv_row_stride % 4) == 0) {
        if (hd == 64) return attn_prefill_mfma_launch<64>(p, s, n_seq);
        if (hd == 128) return attn_prefill_mfma_launch<128>(p, s, n_seq);
    }
    if (n_seq != 1 || p.seq_len) return hipErrorInvalidValue;      // stacked sequences: MFMA kernel only
    dim3 grid((p.M + 63) / 64, p.n_heads);
    const size_t lds = (size_t)2 * 64 * hd * sizeof(float);
    if (hd == 64) {
        attn_prefill_kernel<64><<<grid, dim3(256), lds, s>>>(p);
    } else if (hd == 128) {
        auto kern = attn_prefill_kernel<128>;
        static DevOnce attr_done;
        hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
        if (e != hipSuccess) return e;
        kern<<<grid, dim3(256), lds, s>>>(p);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// single-query GQA attention against the cache (gguf/model.rs:125-174 with q_len == 1): one workgroup per
// q-head, KV heads are NOT expanded (model.rs:177-197 materialises x4; here q-head h reads kv-head h/group).
// ------------------------------------------------------------------------------------------------
// One query head against its KV head's cache rows [j_lo, pos], 256 threads; shared by attn_decode_kernel and attn_wo_kernel.  NT: every q / K / V load is non-temporal (bypasses this CU's L1,
// served by L2 / memory): required when the rows were written by OTHER workgroups of the SAME launch (write-through stores).
// Returns the normalised output float4 for threads tid < HD/4 (column tid); sc: n floats, red: 8 floats, osum: 256 float4 of LDS.
template <bool NT>
__device__ __forceinline__ float4 ldf4(const float* p) {
    if (NT) { const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p)); return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
    return *reinterpret_cast<const float4*>(p);
}
// LATE_V: the V rows of the first NPRE*32 keys are requested only after their K rows have been consumed (the K registers are reused:
// ~half the VGPRs, one more round trip that overlaps the softmax) -- for callers that need 3 waves per SIMD (none today).
// SPEC: the K / V rows of the first NPRE*32 keys are requested at rows 0 .. NPRE*32-1 of the cache (clamped to its spec_rows rows) WITHOUT
// knowing pos -- the position word was written by the previous step's argmax kernel (another XCD: a memory round trip), and with the row
// indices depending on it the kernel was pos -> K -> softmax -> V; now pos, q, K and V are four independent requests in flight together.
// Rows past pos hold stale or uninitialised data: their scores are never stored and their V values are replaced by 0.  A sliding window
// that has started to move (j_lo > 0) re-requests the rows (uniform branch).
// after_issue(): called once the q / K (/ V) requests of the first NPRE*32 keys are out -- a caller's own loads placed there are YOUNGER in the
// in-order vmcnt queue, so the waits for K and V do not wait for them (attn_wo_kernel: its block of wo).
template <int HD, bool NT, bool LATE_V, bool SPEC, class Hook>
__device__ __forceinline__ float4 attn_decode_core(const float* __restrict__ qh, const float* __restrict__ kb, const float* __restrict__ vb, int kv_row_stride,
                                                   int pos_in, int window, float* __restrict__ sc, float* __restrict__ red, float4* __restrict__ osum, int tl_slot, int tlw,
                                                   int spec_rows, Hook after_issue) {
    // Latency-bound (a few hundred KB of K/V per layer): the structure maximises independent loads in flight.
    // scores: 8 lanes per key (each lane HD/8 contiguous floats, float4 loads), 32 keys per pass, 2 passes unrolled;
    // P.V   : 8 key groups x HD/4 float4 columns, 4 keys unrolled.  All loads are unconditional (clamped).
    static_assert(HD == 128 || HD == 64, "head_dim");
    constexpr int PER = HD / 8;       // floats per lane in the score phase
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float scale = 1.0f / sqrtf((float)HD);
    const int ks = tid >> 3, part = tid & 7;
    (void)tl_slot; (void)tlw;
    float qv[PER];
#pragma unroll
    for (int e = 0; e < PER; e += 4) {
        const float4 v = ldf4<NT>(qh + part * PER + e);
        qv[e] = v.x; qv[e + 1] = v.y; qv[e + 2] = v.z; qv[e + 3] = v.w;
    }
    // The first NPRE*32 keys (every 16 s clip: <= 146 positions) are handled with ALL their K and V loads issued up front, before any
    // arithmetic: the kernel is a chain of dependent round trips (q -> K -> softmax -> V) and this collapses the K and V trips into one.
    constexpr int NPRE = LATE_V ? 4 : 5, COLS = HD / 4, GROUPS = 256 / COLS;      // LATE_V: 128 keys up front
    static_assert(4 * GROUPS == 32 || HD != 128, "prefetch tiling assumes 32 keys per P.V iteration for HD = 128");
    const int grp = tid / COLS, col = tid % COLS;
    float4 kpre[NPRE][PER / 4], vpre[NPRE][4];
#define VOX_KPRE(ROW_)                                                                                        \
    _Pragma("unroll") for (int u = 0; u < NPRE; u++) {                                                        \
        const int jc = (ROW_);                                                                                \
        const float* kr = kb + (size_t)jc * kv_row_stride + part * PER;                                       \
        _Pragma("unroll") for (int e = 0; e < PER / 4; e++) kpre[u][e] = ldf4<NT>(kr + 4 * e);                \
    }
#define VOX_VPRE_AT(ROW_)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < NPRE; u++)                                                          \
        _Pragma("unroll") for (int w = 0; w < 4; w++) {                                                       \
            const int iv = u * 4 * GROUPS + grp + w * GROUPS;                                                 \
            vpre[u][w] = ldf4<NT>(vb + (size_t)(ROW_) * kv_row_stride + col * 4);                             \
        }
#define VOX_VPRE VOX_VPRE_AT(j_lo + min(iv, n - 1))
    static_assert(!SPEC || !LATE_V, "speculative rows: K and V are both requested up front");
    if (SPEC) {                              // pos_in is still in flight (a per-lane vector load issued by the caller): nothing here may use it
        VOX_KPRE(min(32 * u + ks, spec_rows - 1))
        VOX_VPRE_AT(min(iv, spec_rows - 1))
        __builtin_amdgcn_sched_barrier(0);   // keep the wait for the position word BELOW the requests
    }
    const int pos = SPEC ? __builtin_amdgcn_readfirstlane(pos_in) : pos_in;      // SPEC: the first wait on the position word, all requests are out
    const int len = pos + 1;
    const int j_lo = window >= 0 ? max(0, pos - window) : 0;
    const int n = len - j_lo;
    if (SPEC) {
        if (j_lo != 0) {                     // the window moved: rows 0.. are not the window's rows (rare; uniform)
            VOX_KPRE(j_lo + min(32 * u + ks, n - 1))
            VOX_VPRE
        }
    } else {
        VOX_KPRE(j_lo + min(32 * u + ks, n - 1))
        if (!LATE_V) { VOX_VPRE }
    }
    __builtin_amdgcn_sched_barrier(0);
    after_issue();
    __builtin_amdgcn_sched_barrier(0);
#undef VOX_KPRE
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < PER / 4; e++) {
            const float4 kv = kpre[u][e];
            s = fmaf(qv[4 * e], kv.x, s); s = fmaf(qv[4 * e + 1], kv.y, s); s = fmaf(qv[4 * e + 2], kv.z, s); s = fmaf(qv[4 * e + 3], kv.w, s);
        }
        s = group8_sum(s);
        const int i = 32 * u + ks;
        if (part == 0 && i < n) sc[i] = s * scale;
    }
    if (LATE_V) { VOX_VPRE }
#undef VOX_VPRE
#undef VOX_VPRE_AT
    for (int i0 = 32 * NPRE; i0 < n; i0 += 64) {
        float s2[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int i = i0 + 32 * u + ks, jc = j_lo + min(i, n - 1);
            const float* kr = kb + (size_t)jc * kv_row_stride + part * PER;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < PER; e += 4) {
                const float4 kv = ldf4<NT>(kr + e);
                s = fmaf(qv[e], kv.x, s); s = fmaf(qv[e + 1], kv.y, s); s = fmaf(qv[e + 2], kv.z, s); s = fmaf(qv[e + 3], kv.w, s);
            }
            s2[u] = s;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float s = group8_sum(s2[u]);
            const int i = i0 + 32 * u + ks;
            if (part == 0 && i < n) sc[i] = s * scale;
        }
    }
    __syncthreads();
    VOX_TL(tl_slot, tlw, 1);
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) { const float e = expf(sc[i] - mx); sc[i] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    VOX_TL(tl_slot, tlw, 2);
    sum = (red[4] + red[5]) + (red[6] + red[7]);
    // P.V : thread -> (key group, float4 column)
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * GROUPS == 32) {
#pragma unroll
        for (int u = 0; u < NPRE; u++)
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int i = u * 32 + grp + w * GROUPS;
                const float pr = i < n ? sc[min(i, n - 1)] : 0.f;
                float4 vv = vpre[u][w];
                if (SPEC && i >= n) vv = make_float4(0.f, 0.f, 0.f, 0.f);      // a row past pos: 0 * (stale bits, possibly NaN) must stay 0
                o.x = fmaf(pr, vv.x, o.x); o.y = fmaf(pr, vv.y, o.y); o.z = fmaf(pr, vv.z, o.z); o.w = fmaf(pr, vv.w, o.w);
            }
    }
    for (int i0 = (4 * GROUPS == 32 ? 32 * NPRE : 0) + grp; i0 < n; i0 += 4 * GROUPS) {
        float4 vv[4]; float pr[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * GROUPS, ic = min(i, n - 1);
            vv[u] = ldf4<NT>(vb + (size_t)(j_lo + ic) * kv_row_stride + col * 4);
            pr[u] = i < n ? sc[ic] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            o.x = fmaf(pr[u], vv[u].x, o.x); o.y = fmaf(pr[u], vv[u].y, o.y); o.z = fmaf(pr[u], vv[u].z, o.z); o.w = fmaf(pr[u], vv[u].w, o.w);
        }
    }
    osum[tid] = o;
    __syncthreads();
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < COLS) {
        float4 t = osum[tid];
#pragma unroll
        for (int gq = 1; gq < GROUPS; gq++) { const float4 u = osum[gq * COLS + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float inv = 1.0f / sum;
        r4 = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    return r4;
}

template <int HD, bool SPEC>
__global__ __launch_bounds__(256) void attn_decode_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sc = smem;                 // scores / probabilities, up to max_seq
    __shared__ float red[8];
    __shared__ float4 osum[256];
    const int tid = threadIdx.x, wave = tid >> 6;
    // Many sequences: hardware deals workgroups round-robin to the 8 XCDs (each with its own L2) by linear id, which would scatter the
    // G query heads that share one KV head over G different L2s.  Re-map so that they occupy consecutive slots of ONE XCD: the K / V rows
    // are then fetched from HBM / MALL once and hit in L2 for the other G-1 heads.
    int h = blockIdx.x, seq = blockIdx.y;
    {
        const int G_ = p.n_heads / p.n_kv_heads, total = gridDim.x * gridDim.y;
        if (G_ > 1 && total % (8 * G_) == 0 && !p.no_xcd_remap) {
            const int lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, slot = lin >> 3;
            const int pair = xcd * (total / (8 * G_)) + slot / G_;                 // (kv head, sequence) pair index
            h = (pair % p.n_kv_heads) * G_ + slot % G_; seq = pair / p.n_kv_heads;
        }
    }
    const int kvh = h / (p.n_heads / p.n_kv_heads);
    // SPEC: the position word is fetched with a per-lane VECTOR load (the address carries a term the compiler cannot fold: p.spec_zero == 0):
    // a scalar load would be waited for with lgkmcnt(0) before the first K request; a vector load is simply the oldest entry of the in-order
    // vmcnt queue and the core waits for it only after q, K and V have been requested.
    int pos;
    if (SPEC) pos = (p.pos_ptr ? __builtin_nontemporal_load(p.pos_ptr + (p.pos_per_seq ? seq : 0) + (tid & p.spec_zero)) : 0) + p.offset;
    else pos = (p.pos_ptr ? p.pos_ptr[p.pos_per_seq ? seq : 0] : 0) + p.offset;
    const int crow = p.kv_row ? p.kv_row[seq] : seq;      // cache slice of this sequence (continuous batching: the utterance the slot holds)
    const float* kb = p.k + (size_t)crow * p.kv_seq_stride + (size_t)kvh * p.kv_head_stride;
    const float* vb = p.v + (size_t)crow * p.kv_seq_stride + (size_t)kvh * p.kv_head_stride;
    const float* qrow = p.q + (size_t)seq * p.q_seq_stride;
    float* orow = p.out + (size_t)seq * p.out_seq_stride;
    const int tlw = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave; (void)tlw;
    VOX_TL(p.tl_slot, tlw, 0);
    const float4 r4 = attn_decode_core<HD, false, false, SPEC>(qrow + h * HD, kb, vb, p.kv_row_stride, pos, p.window, sc, red, osum, p.tl_slot, tlw, p.spec_rows);
    if (tid < HD / 4) {
        if (p.out_xf) xf_store4(p.out_xf + (size_t)(seq >> 4) * p.out_xf_gstride, p.n_heads * HD, seq & 15, h * HD + tid * 4, r4);       // batched decode: A-fragments of the wo GEMM
        else *reinterpret_cast<float4*>(orow + h * HD + tid * 4) = r4;
    }
    VOX_TL(p.tl_slot, tlw, 3);
}
// Batched decode (many sequences): one workgroup per (KV head, sequence) serves all G = n_heads / n_kv_heads query heads of the
// group, so every K / V row is fetched from L2 once instead of G times (512 -> 128 workgroups at 16 sequences, a quarter of the
// traffic).  Measured: neutral for one group of 16 sequences, +3 % at 64 sequences (four concurrent groups) -- used for wide batches only.  Same structure: all K loads of the
// first 160 keys up front, softmax through LDS, V loads issued before the softmax, P.V reduced across 8 key groups.
template <int G>
__global__ __launch_bounds__(256) void attn_decode_gqa_kernel(const AttnParams p) {
    constexpr int HD = 128, PER = HD / 8, NPRE = 5, COLS = HD / 4, GROUPS = 256 / COLS;   // GROUPS = 8
    extern __shared__ __attribute__((aligned(16))) float smem[];      // scores [G][max_seq]
    __shared__ float red[2 * G * 4];
    __shared__ float4 osum[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kvh = blockIdx.x, seq = blockIdx.y, max_seq = p.kv_head_stride / HD;
    const int pos = (p.pos_ptr ? p.pos_ptr[p.pos_per_seq ? seq : 0] : 0) + p.offset;
    const int len = pos + 1;
    const int j_lo = p.window >= 0 ? max(0, pos - p.window) : 0;
    const int n = len - j_lo;
    const float scale = 1.0f / sqrtf((float)HD);
    const int crow = p.kv_row ? p.kv_row[seq] : seq;      // cache slice of this sequence (continuous batching: the utterance the slot holds)
    const float* kb = p.k + (size_t)crow * p.kv_seq_stride + (size_t)kvh * p.kv_head_stride;
    const float* vb = p.v + (size_t)crow * p.kv_seq_stride + (size_t)kvh * p.kv_head_stride;
    const float* qrow = p.q + (size_t)seq * p.q_seq_stride + (size_t)kvh * G * HD;
    const int ks = tid >> 3, part = tid & 7;
    const int grp = tid / COLS, col = tid % COLS;
    float4 kpre[NPRE][PER / 4];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int jc = j_lo + min(32 * u + ks, n - 1);
        const float* kr = kb + (size_t)jc * p.kv_row_stride + part * PER;
#pragma unroll
        for (int e = 0; e < PER / 4; e++) kpre[u][e] = *reinterpret_cast<const float4*>(kr + 4 * e);
    }
    float4 qv[G][PER / 4];
#pragma unroll
    for (int hh = 0; hh < G; hh++)
#pragma unroll
        for (int e = 0; e < PER / 4; e++) qv[hh][e] = *reinterpret_cast<const float4*>(qrow + hh * HD + part * PER + 4 * e);
#define VOX_SCORE(KR_, I_)                                                                                      \
    _Pragma("unroll") for (int hh = 0; hh < G; hh++) {                                                          \
        float s_ = 0.f;                                                                                         \
        _Pragma("unroll") for (int e = 0; e < PER / 4; e++) {                                                   \
            const float4 kv = KR_[e], qq = qv[hh][e];                                                           \
            s_ = fmaf(qq.x, kv.x, s_); s_ = fmaf(qq.y, kv.y, s_); s_ = fmaf(qq.z, kv.z, s_); s_ = fmaf(qq.w, kv.w, s_); \
        }                                                                                                       \
        s_ = group8_sum(s_);                                                                                    \
        if (part == 0 && (I_) < n) smem[hh * max_seq + (I_)] = s_ * scale;                                      \
    }
#pragma unroll
    for (int u = 0; u < NPRE; u++) { const int i = 32 * u + ks; VOX_SCORE(kpre[u], i) }
    for (int i0 = 32 * NPRE; i0 < n; i0 += 32) {
        const int i = i0 + ks, jc = j_lo + min(i, n - 1);
        float4 kr[PER / 4];
#pragma unroll
        for (int e = 0; e < PER / 4; e++) kr[e] = *reinterpret_cast<const float4*>(kb + (size_t)jc * p.kv_row_stride + part * PER + 4 * e);
        VOX_SCORE(kr, i)
    }
#undef VOX_SCORE
    // V rows of the first NPRE*32 keys: in flight while the softmax runs
    float4 vpre[NPRE][4];
#pragma unroll
    for (int u = 0; u < NPRE; u++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int ic = min(u * 32 + grp + w * GROUPS, n - 1);
            vpre[u][w] = *reinterpret_cast<const float4*>(vb + (size_t)(j_lo + ic) * p.kv_row_stride + col * 4);
        }
    __syncthreads();
    float mx[G], sum[G];
#pragma unroll
    for (int hh = 0; hh < G; hh++) {
        float m_ = -INFINITY;
        for (int i = tid; i < n; i += 256) m_ = fmaxf(m_, smem[hh * max_seq + i]);
        m_ = wave_max(m_);
        if (lane == 0) red[hh * 4 + wave] = m_;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < G; hh++) {
        mx[hh] = fmaxf(fmaxf(red[hh * 4], red[hh * 4 + 1]), fmaxf(red[hh * 4 + 2], red[hh * 4 + 3]));
        float s_ = 0.f;
        for (int i = tid; i < n; i += 256) { const float e = expf(smem[hh * max_seq + i] - mx[hh]); smem[hh * max_seq + i] = e; s_ += e; }
        s_ = wave_sum(s_);
        if (lane == 0) red[G * 4 + hh * 4 + wave] = s_;
    }
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < G; hh++) sum[hh] = (red[G * 4 + hh * 4] + red[G * 4 + hh * 4 + 1]) + (red[G * 4 + hh * 4 + 2] + red[G * 4 + hh * 4 + 3]);
    float4 o[G];
#pragma unroll
    for (int hh = 0; hh < G; hh++) o[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < NPRE; u++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int i = u * 32 + grp + w * GROUPS, ic = min(i, n - 1);
            const float4 vv = vpre[u][w];
#pragma unroll
            for (int hh = 0; hh < G; hh++) {
                const float pr = i < n ? smem[hh * max_seq + ic] : 0.f;
                o[hh].x = fmaf(pr, vv.x, o[hh].x); o[hh].y = fmaf(pr, vv.y, o[hh].y); o[hh].z = fmaf(pr, vv.z, o[hh].z); o[hh].w = fmaf(pr, vv.w, o[hh].w);
            }
        }
    for (int i0 = 32 * NPRE + grp; i0 < n; i0 += GROUPS) {
        const float4 vv = *reinterpret_cast<const float4*>(vb + (size_t)(j_lo + i0) * p.kv_row_stride + col * 4);
#pragma unroll
        for (int hh = 0; hh < G; hh++) {
            const float pr = smem[hh * max_seq + i0];
            o[hh].x = fmaf(pr, vv.x, o[hh].x); o[hh].y = fmaf(pr, vv.y, o[hh].y); o[hh].z = fmaf(pr, vv.z, o[hh].z); o[hh].w = fmaf(pr, vv.w, o[hh].w);
        }
    }
#pragma unroll
    for (int hh = 0; hh < G; hh++) {
        __syncthreads();
        osum[tid] = o[hh];
        __syncthreads();
        if (tid < COLS) {
            float4 t = osum[tid];
#pragma unroll
            for (int gq = 1; gq < GROUPS; gq++) { const float4 u = osum[gq * COLS + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            const float inv = 1.0f / sum[hh];
            const float4 r4 = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
            const int h = kvh * G + hh;
            if (p.out_xf) xf_store4(p.out_xf + (size_t)(seq >> 4) * p.out_xf_gstride, p.n_heads * HD, seq & 15, h * HD + tid * 4, r4);
            else *reinterpret_cast<float4*>(p.out + (size_t)seq * p.out_seq_stride + h * HD + tid * 4) = r4;
        }
    }
}
// ---- single-stream decode: attention + wo in ONE launch (gguf/model.rs:125-174 + the wo linear of model.rs:230-238), no cross-workgroup wait.
// Workgroup (h, slice): runs the attention of query head h (redundantly in each of the 8 slices: K / V rows are L2 hits after the first; the
// 4 heads of a KV group and their slices share blockIdx % 8, i.e. one XCD) and multiplies the head's 128 outputs with ITS 384 x 128 block of
// wo -- requested at entry, so the weights stream from HBM while the attention's dependent chain (q, K -> softmax -> V) runs.  The 32-way K
// split of wo is combined with ORDER-INDEPENDENT fixed-point atomics: every partial product is rounded to a multiple of 2^-32 and added as an
// int64 (agent scope) into acc[row] -- integer addition commutes, so the result is bit-identical however the 32 workgroups interleave
// (tools/micro/atomic_reduce.hip: +1.1 us for the 98 k atomics of a launch).  The consumer (q4_gemv_kernel PRO_RMS_MUL_SUM) converts back,
// adds the residual and writes the new residual stream.  Replaces two launches (attention: 32 workgroups with HBM idle; wo GEMV: bound by the
// cross-XCD read of the attention output) -- profiles/r02_attn_wo.txt.
template <int NI>      // NI passes of 64 rows per workgroup (6: 8 row slices per head, 256 workgroups)
__global__ __launch_bounds__(256) void attn_wo_kernel(const AttnParams p, const Q4W wo, long long* __restrict__ acc) {
    constexpr int HD = 128, CH = HD / 32, RPI = 256 / CH;             // 4 chunks per head; 64 rows per pass
    extern __shared__ __attribute__((aligned(16))) float smem[];      // scores [max_seq]
    __shared__ float red[8];
    __shared__ float4 osum[256];
    __shared__ __attribute__((aligned(16))) float4 a_s[HD / 4];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int G = p.n_heads / p.n_kv_heads, n_slices = gridDim.x / p.n_heads;
    const int kvh = blockIdx.x % p.n_kv_heads, j_ = blockIdx.x / p.n_kv_heads, h = kvh * G + j_ / n_slices, slice = j_ % n_slices;
    const int tlw = blockIdx.x * 4 + wave; (void)tlw;
    VOX_TL(p.tl_slot, tlw, 0);
    // (1) this workgroup's block of wo: thread -> chunk c of rows r0 + 64 i
    const int c = tid & (CH - 1), r0 = slice * (RPI * NI) + (tid >> 2);
    uint4 wq[NI]; uint16_t wd[NI];
    auto load_w = [&]() {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const size_t idx = (size_t)(r0 + RPI * i) * wo.nb + h * CH + c;
            wq[i] = ld_nt_u4(wo.qs + idx); wd[i] = __builtin_nontemporal_load(wo.sc + idx);
        }
    };
    // (2) attention of head h; the weight requests go out right behind its q / K / V requests
    const int pos = (p.pos_ptr ? p.pos_ptr[0] : 0) + p.offset;
    const float4 r4 = attn_decode_core<HD, false, false, false>(p.q + h * HD, p.k + (size_t)kvh * p.kv_head_stride, p.v + (size_t)kvh * p.kv_head_stride,
                                                                 p.kv_row_stride, pos, p.window, smem, red, osum, p.tl_slot, tlw, 0, load_w);
    if (tid < HD / 4) a_s[tid] = r4;
    __syncthreads();
    VOX_TL(p.tl_slot, tlw, 2);
    // (3) GEMV block: the lane's chunk column is the same for its NI rows
    float xv[32]; float sx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float4 v = a_s[c * 8 + j];
        xv[4 * j] = v.x; xv[4 * j + 1] = v.y; xv[4 * j + 2] = v.z; xv[4 * j + 3] = v.w;
        sx += (v.x + v.y) + (v.z + v.w);
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        float val = f16_bits_to_f32(wd[i]) * (q4_chunk_dot(wq[i], xv) - 8.0f * sx);
        val += __shfl_xor(val, 1, 64); val += __shfl_xor(val, 2, 64);
        if (c == 0) __hip_atomic_fetch_add(acc + r0 + RPI * i, __double2ll_rn((double)val * 4294967296.0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    VOX_TL(p.tl_slot, tlw, 3);
}
bool attn_wo_supported(const AttnParams& p, const Q4W& wo, int hd, int max_seq) {
    const int G = p.n_kv_heads > 0 ? p.n_heads / p.n_kv_heads : 0;
    if (wo.fmt != WFMT_Q4_0 || hd != 128 || G < 1 || p.n_heads != G * p.n_kv_heads || wo.K != p.n_heads * hd || p.kv_row_stride != hd || p.kv_head_stride != max_seq * hd) return false;
    if (256 % p.n_heads) return false;
    const int n_slices = 256 / p.n_heads;                     // 256 workgroups: one per CU
    return wo.N % (n_slices * 64) == 0 && wo.N / (n_slices * 64) == 6 && !env_int("VOX_NO_ATTN_WO");
}
hipError_t launch_attn_wo(const AttnParams& p_in, const Q4W& wo, long long* acc, int max_seq, hipStream_t s) {
    if (!attn_wo_supported(p_in, wo, 128, max_seq) || !acc) return hipErrorInvalidValue;
    AttnParams p = p_in; p.tl_slot = tl_take_slot(1, 0, p.n_heads, 128);
    const size_t lds = (size_t)max_seq * sizeof(float);
    static DevOnce attr_done;
    auto kern = attn_wo_kernel<6>;       // 8 row slices per head = 256 workgroups (4 slices: 1.05 ms per step, 16: 0.975, 8: 0.94 -- profiles/r02_attn_wo.txt)
    hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
    if (e != hipSuccess) return e;
    kern<<<dim3(256), dim3(256), lds, s>>>(p, wo, acc);
    return hipGetLastError();
}

hipError_t launch_attn_decode(const AttnParams& p_in, int hd, int max_seq, hipStream_t s, int n_seq) {
    AttnParams p = p_in; p.tl_slot = tl_take_slot(1, 0, p.n_heads, hd);
    const size_t lds = (size_t)max_seq * sizeof(float);
    if (hd == 128 && p.prefer_gqa && p.n_heads == 4 * p.n_kv_heads && p.kv_head_stride == max_seq * 128 && !env_int("VOX_ATTN_NO_GQA")) {
        auto kern = attn_decode_gqa_kernel<4>;      // many sequences: one workgroup per (KV head, sequence), K/V fetched once for its 4 query heads
        static DevOnce attr_done;
        hipError_t e = ensure_dyn_lds(kern, 4 * lds, &attr_done);
        if (e != hipSuccess) return e;
        kern<<<dim3(p.n_kv_heads, n_seq), dim3(256), 4 * lds, s>>>(p);
        return hipGetLastError();
    }
    // Speculative K / V rows (attn_decode_core SPEC) are opt-in (VOX_ATTN_SPEC=1): measured neutral for one sequence (the position word is not the
    // long pole: the K rows themselves take ~2.5 us to arrive) and 14 % slower at 16 sequences (every workgroup then requests 160 rows whatever
    // its length) -- profiles/r02_decode_knobs.txt.
    if (!env_int("VOX_ATTN_SPEC")) p.spec_rows = 0;
    if (hd == 128 && p.spec_rows > 0) {
        auto kern = attn_decode_kernel<128, true>;
        static DevOnce attr_done;
        hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
        if (e != hipSuccess) return e;
        kern<<<dim3(p.n_heads, n_seq), dim3(256), lds, s>>>(p);
    } else if (hd == 128) {
        auto kern = attn_decode_kernel<128, false>;
        static DevOnce attr_done;
        hipError_t e = ensure_dyn_lds(kern, lds, &attr_done);
        if (e != hipSuccess) return e;
        kern<<<dim3(p.n_heads, n_seq), dim3(256), lds, s>>>(p);
    } else if (hd == 64) {
        auto kern = attn_decode_kernel<64, false>;
        kern<<<dim3(p.n_heads, n_seq), dim3(256), lds, s>>>(p);
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Conv1d(k3, s2, p1) + exact-erf GELU (models/layers/conv.rs:78-83). f32, LDS-tiled:
// workgroup = 64 output channels x 64 output positions, 16 input channels per step, 4x4 outputs / thread.
// ------------------------------------------------------------------------------------------------
template <int TOKEN_MAJOR>
__global__ __launch_bounds__(256) void conv1d_gelu_kernel(const float* __restrict__ in, int Cin, int L,
                                                          const float* __restrict__ w, const float* __restrict__ b, int Cout,
                                                          int Lo, float* __restrict__ out) {
    constexpr int CI = 16, TW = 2 * 64 + 1;    // input span for 64 outputs: positions 2*t0-1 .. 2*t0+127
    __shared__ float ins[CI][TW + 3];
    __shared__ float ws[CI][3][64 + 1];
    const int tid = threadIdx.x;
    const int co0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    const int tc = tid & 15, tt = tid >> 4;    // thread: channels co0 + tc + 16*a, positions t0 + tt + 16*bq
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[a][q] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CI) {
        __syncthreads();
        for (int i = tid; i < CI * TW; i += 256) {
            const int ci = i / TW, x = i % TW, pos = 2 * t0 - 1 + x;
            const float v = in[(size_t)min(c0 + ci, Cin - 1) * L + min(max(pos, 0), L - 1)];     // clamped, then masked
            ins[ci][x] = (c0 + ci < Cin && pos >= 0 && pos < L) ? v : 0.f;
        }
        for (int i = tid; i < 64 * CI * 3; i += 256) {
            const int co = i / (CI * 3), rem = i % (CI * 3), ci = rem / 3, kk = rem % 3;
            const float v = w[((size_t)min(co0 + co, Cout - 1) * Cin + min(c0 + ci, Cin - 1)) * 3 + kk];
            ws[ci][kk][co] = (co0 + co < Cout && c0 + ci < Cin) ? v : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int ci = 0; ci < CI; ci++) {
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                float wv[4], xv[4];
#pragma unroll
                for (int a = 0; a < 4; a++) wv[a] = ws[ci][kk][tc + 16 * a];
#pragma unroll
                for (int q = 0; q < 4; q++) xv[q] = ins[ci][2 * (tt + 16 * q) + kk];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[a][q] = fmaf(xv[q], wv[a], acc[a][q]);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int co = co0 + tc + 16 * a, t = t0 + tt + 16 * q;
            if (co < Cout && t < Lo) {
                const float v = gelu_f(acc[a][q] + (b ? b[co] : 0.f));
                if (TOKEN_MAJOR) out[(size_t)t * Cout + co] = v; else out[(size_t)co * Lo + t] = v;
            }
        }
}
hipError_t launch_conv1d_gelu(const float* in, int Cin, int L, const float* w, const float* b, int Cout, float* out,
                              int token_major, hipStream_t s) {
    const int Lo = (L + 2 - 3) / 2 + 1;
    dim3 grid((Lo + 63) / 64, (Cout + 63) / 64);
    if (token_major) conv1d_gelu_kernel<1><<<grid, dim3(256), 0, s>>>(in, Cin, L, w, b, Cout, Lo, out);
    else conv1d_gelu_kernel<0><<<grid, dim3(256), 0, s>>>(in, Cin, L, w, b, Cout, Lo, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// log-mel (audio/mel.rs:128-244): one workgroup per frame. Reflect-padded STFT frame * periodic Hann,
// 400-point real DFT (201 bins) by table lookup, power, Slaney filterbank, log10 / floor / scale.
// The padded signal zeros(left) + scale*audio + zeros(right) is never materialised.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ audio, long n, long left, long right,
                                                  const float* __restrict__ scale_ptr, MelTables t, float* __restrict__ out,
                                                  int T, int transposed) {
    __shared__ float fr[400];
    __shared__ float ct[400], st[400];
    __shared__ float pw[201 + 3];
    const int f = blockIdx.x, tid = threadIdx.x;
    const long plen = left + n + right;             // length of the padded signal handed to the STFT
    const float scale = scale_ptr ? *scale_ptr : 1.0f;
    for (int j = tid; j < 400; j += 256) {
        long idx = (long)f * 160 + j - 200;          // index into the padded signal, before reflection (mel.rs:190-205)
        if (idx < 0) { long r = -idx; long lim = plen > 0 ? plen - 1 : 0; idx = r < lim ? r : lim; }
        else if (idx >= plen) { long i2 = idx - plen; long a = plen >= 2 ? plen - 2 : 0; idx = a >= i2 ? a - i2 : 0; }
        const long ai = idx - left;
        float v = 0.f;
        if (plen > 0 && ai >= 0 && ai < n) { v = audio[ai]; if (scale_ptr) v *= scale; }
        fr[j] = v * t.window[j];
        ct[j] = t.cos_t[j]; st[j] = t.sin_t[j];
    }
    __syncthreads();
    if (tid < 201) {
        float re = 0.f, im = 0.f; int ph = 0;
        for (int j = 0; j < 400; j++) {
            const float v = fr[j];
            re = fmaf(v, ct[ph], re); im = fmaf(-v, st[ph], im);
            ph += tid; if (ph >= 400) ph -= 400;
        }
        pw[tid] = re * re + im * im;
    }
    __syncthreads();
    if (tid < 128) {
        const int lo = t.fb_lo[tid], hi = t.fb_hi[tid];
        const float* row = t.fb + tid * 201;
        float acc = 0.f;
        for (int j = lo; j < hi; j++) acc += row[j] * pw[j];
        float v = log10f(fmaxf(acc, 1e-10f));
        v = fmaxf(v, 1.5f - 8.0f);
        v = (v + 4.0f) / 4.0f;
        if (transposed) out[(size_t)tid * T + f] = v; else out[(size_t)f * 128 + tid] = v;
    }
}
hipError_t launch_mel(const float* audio, long n, long left, long right, const float* scale_ptr, MelTables t, float* out, int T,
                      int transposed, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    mel_kernel<<<dim3(T), dim3(256), 0, s>>>(audio, n, left, right, scale_ptr, t, out, T, transposed);
    return hipGetLastError();
}

// peak_normalize scale (audio/io.rs:59-68): scale = target / max|x| (1 if max < 1e-10). Single workgroup.
__global__ __launch_bounds__(1024) void absmax_kernel(const float* __restrict__ x, long n, float target, float* __restrict__ scale_out) {
    __shared__ float red[16];
    float m = 0.f;
    // max is exact and order-independent: 16-byte loads, four per thread in flight (the scalar loop was a chain of 250 dependent round trips per thread: 65 us for a 16 s clip)
    const long head = min(n, (long)(((16 - ((uintptr_t)x & 15)) & 15) >> 2)), n4 = (n - head) >> 2;
    for (long i = threadIdx.x; i < head; i += 1024) m = fmaxf(m, fabsf(x[i]));
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
    for (long i = threadIdx.x; i < n4; i += 4096) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = x4[min(i + 1024 * u, n4 - 1)];      // clamped: a repeated element does not change a maximum
#pragma unroll
        for (int u = 0; u < 4; u++) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (long i = head + 4 * n4 + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; i++) m = fmaxf(m, red[i]);
        *scale_out = m < 1e-10f ? 1.0f : target / m;
    }
}
hipError_t launch_absmax(const float* x, long n, float target, float* scale_out, hipStream_t s) {
    absmax_kernel<<<dim3(1), dim3(1024), 0, s>>>(x, n, target, scale_out);
    return hipGetLastError();
}

// Peak of a GROUP of units (the 1200-frame chunks of one file: the reference normalises the FILE once, bin/transcribe.rs:207, then chunks it, :210-226): every unit
// folds its max|x| into its group's cell (non-negative floats order like their bit patterns; a maximum is exact and order-independent, so the atomics are deterministic),
// then unit_scale[i] = target / max of its group (1 if the group is silent, audio/io.rs:61-63); group < 0 = the unit is used as it is (scale 1).
__global__ __launch_bounds__(1024) void absmax_group_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ group_max) {
    __shared__ float red[16];
    float m = 0.f;
    const long head = min(n, (long)(((16 - ((uintptr_t)x & 15)) & 15) >> 2)), n4 = (n - head) >> 2;
    for (long i = threadIdx.x; i < head; i += 1024) m = fmaxf(m, fabsf(x[i]));
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
    for (long i = threadIdx.x; i < n4; i += 4096) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = x4[min(i + 1024 * u, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    for (long i = head + 4 * n4 + threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; i++) m = fmaxf(m, red[i]);
        atomicMax(group_max, __float_as_uint(m));
    }
}
__global__ void group_scale_kernel(const unsigned* __restrict__ group_max, const int* __restrict__ unit_group, int n, float target, float* __restrict__ unit_scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int g = unit_group[i];
    float sc = 1.0f;
    if (g >= 0) { const float m = __uint_as_float(group_max[g]); sc = m < 1e-10f ? 1.0f : target / m; }
    unit_scale[i] = sc;
}
hipError_t launch_absmax_group(const float* x, long n, unsigned* group_max_cell, hipStream_t s) {
    absmax_group_kernel<<<dim3(1), dim3(1024), 0, s>>>(x, n, group_max_cell);
    return hipGetLastError();
}
hipError_t launch_group_scale(const unsigned* group_max, const int* unit_group, int n, float target, float* unit_scale, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    group_scale_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(group_max, unit_group, n, target, unit_scale);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// token embedding row dequant (+ audio embedding add) (gguf/model.rs:584-618, :898-902, :942-948)
// ------------------------------------------------------------------------------------------------
// one token-embedding row (Q4_0 row dequant, gguf/model.rs:584-618, or dense bf16 row, models/decoder.rs:250-262)
// plus the audio embedding of the same position (gguf/model.rs:898-902, :942-948); 256 threads cooperate.
__device__ __forceinline__ void embed_row(const Q4W& tok, int id, const float* __restrict__ arow, float* __restrict__ o, int D) {
    id = (unsigned)id < (unsigned)tok.N ? id : 0;      // an argmax over NaN logits (a decode engine that timed out half-way leaves garbage behind) yields no index: never read out of bounds
    if (tok.fmt == WFMT_F32) {
        const float4* w = reinterpret_cast<const float4*>(tok.qt) + (size_t)id * (D >> 2);
        for (int c = threadIdx.x; c < (D >> 2); c += blockDim.x) {
            float4 a = w[c];
            if (arow) { const float4 u = reinterpret_cast<const float4*>(arow)[c]; a.x = u.x + a.x; a.y = u.y + a.y; a.z = u.z + a.z; a.w = u.w + a.w; }
            reinterpret_cast<float4*>(o)[c] = a;
        }
        return;
    }
    if (tok.fmt == WFMT_BF16) {
        const uint4* w = tok.qs + (size_t)id * (D >> 3);
        for (int c = threadIdx.x; c < (D >> 3); c += blockDim.x) {
            const uint4 q = w[c];
            float4 a = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xFFFF0000u));
            float4 b = make_float4(__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xFFFF0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xFFFF0000u));
            if (arow) {
                const float4 u = reinterpret_cast<const float4*>(arow)[2 * c], v = reinterpret_cast<const float4*>(arow)[2 * c + 1];
                a.x = u.x + a.x; a.y = u.y + a.y; a.z = u.z + a.z; a.w = u.w + a.w; b.x = v.x + b.x; b.y = v.y + b.y; b.z = v.z + b.z; b.w = v.w + b.w;
            }
            reinterpret_cast<float4*>(o)[2 * c] = a; reinterpret_cast<float4*>(o)[2 * c + 1] = b;
        }
        return;
    }
    for (int c = threadIdx.x; c < tok.nb; c += blockDim.x) {
        const uint4 q = tok.qs[(size_t)id * tok.nb + c];
        const float d = f16_bits_to_f32(tok.sc[(size_t)id * tok.nb + c]);
        const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t lo = ww[i] & 0x0F0F0F0Fu, hi = (ww[i] >> 4) & 0x0F0F0F0Fu;
            float4 a = make_float4((ub0(lo) - 8.0f) * d, (ub1(lo) - 8.0f) * d, (ub2(lo) - 8.0f) * d, (ub3(lo) - 8.0f) * d);
            float4 b = make_float4((ub0(hi) - 8.0f) * d, (ub1(hi) - 8.0f) * d, (ub2(hi) - 8.0f) * d, (ub3(hi) - 8.0f) * d);
            if (arow) {
                const float4 u = reinterpret_cast<const float4*>(arow)[c * 8 + i], v = reinterpret_cast<const float4*>(arow)[c * 8 + 4 + i];
                a.x = u.x + a.x; a.y = u.y + a.y; a.z = u.z + a.z; a.w = u.w + a.w; b.x = v.x + b.x; b.y = v.y + b.y; b.z = v.z + b.z; b.w = v.w + b.w;
            }
            reinterpret_cast<float4*>(o)[c * 8 + i] = a; reinterpret_cast<float4*>(o)[c * 8 + 4 + i] = b;
        }
    }
}

__global__ __launch_bounds__(256) void embed_kernel(Q4W tok, const int* __restrict__ ids, const float* __restrict__ audio, int D,
                                                    const int* __restrict__ pos_ptr, int id_off, int a_off, float* __restrict__ out) {
    const int i = blockIdx.x;
    const int base = pos_ptr ? *pos_ptr : 0;
    const int id = ids[base + id_off + i];
    embed_row(tok, id, audio ? audio + (size_t)(base + a_off + i) * D : nullptr, out + (size_t)i * D, D);
}
hipError_t launch_embed(Q4W tok, const int* ids, int n, const float* audio, int D, const int* pos_ptr, int id_off, int a_off,
                        float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    embed_kernel<<<dim3(n), dim3(256), 0, s>>>(tok, ids, audio, D, pos_ptr, id_off, a_off, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int n_parts,
                                                           int* __restrict__ tokens, int* __restrict__ pos_ptr, int tok_off, int inc) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    float v = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n_parts; i += 256) {
        const float x = pv[i]; const int ii = pi[i];
        if (x > v || (x == v && ii < idx)) { v = x; idx = ii; }
    }
    bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float x = bv[threadIdx.x + s]; const int ii = bi[threadIdx.x + s];
            if (x > bv[threadIdx.x] || (x == bv[threadIdx.x] && ii < bi[threadIdx.x])) { bv[threadIdx.x] = x; bi[threadIdx.x] = ii; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int base = pos_ptr ? *pos_ptr : 0;
        tokens[base + tok_off] = bi[0];
        if (pos_ptr && inc) *pos_ptr = base + inc;
    }
}
hipError_t launch_argmax_final(const float* pv, const int* pi, int n_parts, int* tokens, int* pos_ptr, int tok_off, int inc,
                               hipStream_t s) {
    argmax_final_kernel<<<dim3(1), dim3(256), 0, s>>>(pv, pi, n_parts, tokens, pos_ptr, tok_off, inc);
    return hipGetLastError();
}

// decode-step tail: final argmax over the lm_head partials -> tokens[cur+1]; cur += 1; then the NEXT step's input
// h = audio[cur] + dequant(tok_emb[tokens[cur]]) (gguf/model.rs:938-948) -- one launch instead of two.
__global__ __launch_bounds__(256) void argmax_embed_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int n_parts,
                                                           int* __restrict__ tokens, int* __restrict__ pos_ptr, Q4W tok,
                                                           const float* __restrict__ audio, int D, float* __restrict__ h) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ int s_tok, s_cur;
    float v = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n_parts; i += 256) {
        const float x = pv[i]; const int ii = pi[i];
        if (x > v || (x == v && ii < idx)) { v = x; idx = ii; }
    }
    bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const float x = bv[threadIdx.x + st]; const int ii = bi[threadIdx.x + st];
            if (x > bv[threadIdx.x] || (x == bv[threadIdx.x] && ii < bi[threadIdx.x])) { bv[threadIdx.x] = x; bi[threadIdx.x] = ii; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int cur = *pos_ptr + 1;
        tokens[cur] = bi[0]; *pos_ptr = cur; s_tok = bi[0]; s_cur = cur;
    }
    __syncthreads();
    embed_row(tok, s_tok, audio + (size_t)s_cur * D, h, D);
}
hipError_t launch_argmax_embed(const float* pv, const int* pi, int n_parts, int* tokens, int* pos_ptr, Q4W tok, const float* audio, int D,
                               float* h, hipStream_t s) {
    argmax_embed_kernel<<<dim3(1), dim3(256), 0, s>>>(pv, pi, n_parts, tokens, pos_ptr, tok, audio, D, h);
    return hipGetLastError();
}

// ---- batched decode step helpers -------------------------------------------------------------------------------
// RoPE on q,k of every sequence's fused [q|k|v] row at that sequence's own position, k/v scattered into its cache slice.
__global__ void rope_kv_batch_kernel(float* __restrict__ qkv, int n, int stride, int n_q, int n_kv, int hd, const int* __restrict__ pos,
                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t, float* __restrict__ kc,
                                     float* __restrict__ vc, long seq_stride, int head_stride) {
    const int kd = n_kv * hd, pairs = (n_q + 2 * kd) >> 1, half = hd >> 1;
    const long total = (long)n * pairs;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / pairs), col = (int)(i % pairs) * 2, ps = pos[r];
        float* row = qkv + (size_t)r * stride;
        const float a = row[col], b = row[col + 1];
        if (col < n_q + kd) {
            const int dd = col % hd;
            const float c = cos_t[(size_t)ps * half + (dd >> 1)], sn = sin_t[(size_t)ps * half + (dd >> 1)];
            const float ra = a * c - b * sn, rb = a * sn + b * c;
            if (col < n_q) { row[col] = ra; row[col + 1] = rb; }
            else {
                const int kn = col - n_q;
                float* dst = kc + (size_t)r * seq_stride + (size_t)(kn / hd) * head_stride + (size_t)ps * hd + dd;
                dst[0] = ra; dst[1] = rb;
            }
        } else {
            const int vn = col - n_q - kd;
            float* dst = vc + (size_t)r * seq_stride + (size_t)(vn / hd) * head_stride + (size_t)ps * hd + (vn % hd);
            dst[0] = a; dst[1] = b;
        }
    }
}
hipError_t launch_rope_kv_batch(float* qkv, int n, int stride, int n_q, int n_kv, int hd, const int* pos, const float* cos_t, const float* sin_t,
                                float* kc, float* vc, long seq_stride, int head_stride, hipStream_t s) {
    const long total = (long)n * ((n_q + 2 * n_kv * hd) / 2);
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    rope_kv_batch_kernel<<<dim3(blocks), dim3(256), 0, s>>>(qkv, n, stride, n_q, n_kv, hd, pos, cos_t, sin_t, kc, vc, seq_stride, head_stride);
    return hipGetLastError();
}

// one workgroup per sequence: argmax of its logits row (lowest index wins ties) -> tokens[s][pos+1], pos[s]++ (until the
// sequence's last position), then the next step's input h[s] = audio[s][pos] + embed(tokens[s][pos]).
__global__ __launch_bounds__(1024) void argmax_embed_batch_kernel(const float* __restrict__ logits, int vocab, int* __restrict__ tokens,
                                                                 int tok_stride, int* __restrict__ pos, const int* __restrict__ seq_len, Q4W tok,
                                                                 const float* __restrict__ audio, long audio_seq_stride, int D, float* __restrict__ h,
                                                                 uint16_t* __restrict__ xf, const float* __restrict__ xf_w, float* __restrict__ ssq_out,
                                                                 long xf_group_stride, int ssq_group_stride, const long* __restrict__ audio_off) {
    __shared__ float bv[1024];
    __shared__ int bi[1024];
    __shared__ int s_tok, s_cur;
    const int sq = blockIdx.x, nt = blockDim.x;
    const float* lg = logits + (size_t)sq * vocab;
    float v = -INFINITY; int idx = 0x7fffffff;
    const int v4 = vocab >> 2;
    for (int i = threadIdx.x; i < v4; i += nt) {            // ascending i per thread: first max wins
        const float4 x = reinterpret_cast<const float4*>(lg)[i];
        if (x.x > v) { v = x.x; idx = 4 * i; } if (x.y > v) { v = x.y; idx = 4 * i + 1; }
        if (x.z > v) { v = x.z; idx = 4 * i + 2; } if (x.w > v) { v = x.w; idx = 4 * i + 3; }
    }
    for (int i = 4 * v4 + threadIdx.x; i < vocab; i += nt) { const float x = lg[i]; if (x > v) { v = x; idx = i; } }
    bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int st = nt >> 1; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const float x = bv[threadIdx.x + st]; const int ii = bi[threadIdx.x + st];
            if (x > bv[threadIdx.x] || (x == bv[threadIdx.x] && ii < bi[threadIdx.x])) { bv[threadIdx.x] = x; bi[threadIdx.x] = ii; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int cur = pos[sq];
        if (cur + 1 < seq_len[sq]) { cur += 1; tokens[(size_t)sq * tok_stride + cur] = bi[0]; pos[sq] = cur; }   // finished sequences idle in place
        s_cur = cur; s_tok = tokens[(size_t)sq * tok_stride + cur];
    }
    __syncthreads();
    embed_row(tok, s_tok, audio + (audio_off ? (size_t)audio_off[sq] : (size_t)sq * audio_seq_stride) + (size_t)s_cur * D, h + (size_t)sq * D, D);      // audio_off: packed audio rows (no common stride)
    if (xf) {     // the first layer's RMSNorm folded in: XF planes of h * gamma and the row's sum of squares (one partial);
        // sequences are processed in groups of 16 rows, each group with its own XF planes / partial-sum block
        xf += (size_t)(sq >> 4) * xf_group_stride; ssq_out += (size_t)(sq >> 4) * ssq_group_stride;
        const int row = sq & 15;
        __syncthreads();
        float ss = 0.f;
        for (int c = threadIdx.x; c < (D >> 2); c += nt) {
            float4 v = reinterpret_cast<const float4*>(h + (size_t)sq * D)[c]; const float4 gm = reinterpret_cast<const float4*>(xf_w)[c];
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
            xf_store4(xf, D, row, 4 * c, v);
        }
        ss = wave_sum(ss);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) bv[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) { float a = 0.f; for (int w = 0; w < (nt >> 6); w++) a += bv[w]; ssq_out[row] = a; }
    }
}
hipError_t launch_argmax_embed_batch(const float* logits, int n, int vocab, int* tokens, int tok_stride, int* pos, const int* seq_len, Q4W tok,
                                     const float* audio, long audio_seq_stride, int D, float* h, hipStream_t s,
                                     uint16_t* xf, const float* xf_w, float* ssq_out, long xf_group_stride, int ssq_group_stride, const long* audio_off) {
    argmax_embed_batch_kernel<<<dim3(n), dim3(1024), 0, s>>>(logits, vocab, tokens, tok_stride, pos, seq_len, tok, audio, audio_seq_stride, D, h, xf, xf_w, ssq_out,
                                                             xf_group_stride, ssq_group_stride, audio_off);
    return hipGetLastError();
}

// ---- continuous batching (vox_transcribe_batch, wide batches): the decode step runs over SLOTS, not over utterances.  Slot s decodes the utterance
// slot_clip[s]; when that utterance gets its last token the slot takes the next one of its (host-planned, static: token counts are a pure function of the audio
// length, there is no EOS -- gguf/model.rs:936-960) queue in the SAME launch: position = the prefix length, cache slice = the new utterance's (kv_row[s]), input row
// = the utterance's first decode input h0 (audio[38] + embed(first token), computed behind its prefill).  No host round trip, no idle step: the groups of a ragged
// batch stay full until the queues run dry.  One workgroup per slot; `init` = 1: no argmax, every slot takes the head of its queue.
__global__ __launch_bounds__(1024) void argmax_embed_slots_kernel(const SlotStepParams p) {
    __shared__ float bv[1024];
    __shared__ int bi[1024];
    __shared__ int s_tok, s_cur, s_c, s_mode;      // mode 0: next position of the same utterance; 1: the slot switched to utterance s_c; 2: the slot's queue is empty
    const int sl = blockIdx.x, nt = blockDim.x, D = p.D;
    const int c = p.init ? -1 : p.slot_clip[sl];
    if (!p.init && c < 0) return;                   // idle slot (uniform per workgroup): its input row and XF rows stay zero
    if (!p.init) {
        const float* lg = p.logits + (size_t)sl * p.vocab;
        float v = -INFINITY; int idx = 0x7fffffff;
        const int v4 = p.vocab >> 2;
        for (int i = threadIdx.x; i < v4; i += nt) {            // ascending i per thread: first max wins (lowest index on ties, like argmax_embed_batch_kernel)
            const float4 x = reinterpret_cast<const float4*>(lg)[i];
            if (x.x > v) { v = x.x; idx = 4 * i; } if (x.y > v) { v = x.y; idx = 4 * i + 1; }
            if (x.z > v) { v = x.z; idx = 4 * i + 2; } if (x.w > v) { v = x.w; idx = 4 * i + 3; }
        }
        for (int i = 4 * v4 + threadIdx.x; i < p.vocab; i += nt) { const float x = lg[i]; if (x > v) { v = x; idx = i; } }
        bv[threadIdx.x] = v; bi[threadIdx.x] = idx;
        __syncthreads();
        for (int st = nt >> 1; st > 0; st >>= 1) {
            if (threadIdx.x < st) {
                const float x = bv[threadIdx.x + st]; const int ii = bi[threadIdx.x + st];
                if (x > bv[threadIdx.x] || (x == bv[threadIdx.x] && ii < bi[threadIdx.x])) { bv[threadIdx.x] = x; bi[threadIdx.x] = ii; }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        bool take_next = p.init != 0; int q = -1;
        if (!p.init) {
            const int cur = p.pos[sl] + 1;            // the position the step just produced (pos = index of the last token written)
            p.tokens[(size_t)c * p.tok_stride + cur] = bi[0];
            if (cur + 1 < p.clip_len[c]) { p.pos[sl] = cur; s_cur = cur; s_tok = bi[0]; s_c = c; s_mode = 0; }
            else { take_next = true; q = p.slot_qpos[sl]; }
        }
        if (take_next) {
            q += 1;
            const int c2 = q < p.q_stride ? p.queue[(size_t)sl * p.q_stride + q] : -1;
            p.slot_qpos[sl] = q; p.slot_clip[sl] = c2; p.kv_row[sl] = c2 >= 0 ? c2 : p.n_clips; p.pos[sl] = p.first_pos;
            s_c = c2; s_mode = c2 >= 0 ? 1 : 2; s_cur = p.first_pos; s_tok = 0;
        }
    }
    __syncthreads();
    float* hrow = p.h + (size_t)sl * D;
    if (s_mode == 0) embed_row(p.tok, s_tok, p.audio + p.audio_off[s_c] + (size_t)s_cur * D, hrow, D);
    else {
        const float4* src = s_mode == 1 ? reinterpret_cast<const float4*>(p.h0 + (size_t)s_c * D) : nullptr;
        for (int i = threadIdx.x; i < (D >> 2); i += nt) reinterpret_cast<float4*>(hrow)[i] = src ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.xf) {     // the first layer's RMSNorm folded in, exactly as argmax_embed_batch_kernel does for its rows (group = slot / 16, row = slot % 16)
        uint16_t* xf = p.xf + (size_t)(sl >> 4) * p.xf_group_stride; float* ssq_out = p.ssq_out + (size_t)(sl >> 4) * p.ssq_group_stride;
        const int row = sl & 15;
        __syncthreads();
        float ss = 0.f;
        for (int k = threadIdx.x; k < (D >> 2); k += nt) {
            float4 v = reinterpret_cast<const float4*>(hrow)[k]; const float4 gm = reinterpret_cast<const float4*>(p.xf_w)[k];
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            v.x *= gm.x; v.y *= gm.y; v.z *= gm.z; v.w *= gm.w;
            xf_store4(xf, D, row, 4 * k, v);
        }
        ss = wave_sum(ss);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) bv[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) { float a = 0.f; for (int w = 0; w < (nt >> 6); w++) a += bv[w]; ssq_out[row] = a; }
    }
}
hipError_t launch_argmax_embed_slots(const SlotStepParams& p, int n_slots, hipStream_t s) {
    if (n_slots <= 0 || !p.slot_clip || !p.slot_qpos || !p.queue || !p.pos || !p.kv_row || !p.h || !p.h0 || !p.tokens || !p.clip_len || !p.audio || !p.audio_off || (!p.init && !p.logits) || (p.D & 3)) return hipErrorInvalidValue;
    argmax_embed_slots_kernel<<<dim3(n_slots), dim3(1024), 0, s>>>(p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Sample-rate conversion (audio/resample.rs:16-52 `resample`).  The reference calls rubato 1.0's synchronous FFT resampler (`Fft`, FixedSync::Input,
// chunk 1024, 2 sub-chunks): blocks of fft_in samples, zero-padded to 2 fft_in, real FFT, times the FFT of a Blackman-Harris^2 windowed sinc, the
// low new_len bins re-synthesised by an unnormalised inverse real FFT of length 2 fft_out, overlap-added, minus fft_out / 2 samples of delay (plan
// and filter taps: vox_api.cpp resample_plan_make).  Every step is linear and identical for every block, so one block is a fixed
// (2 fft_out) x fft_in matrix applied to the block's samples:
//     out_buf[m] = sum_n x[n] * A[m][n],   A[m][n] = sum_{k < new_len} w_k Re(H[k] e^{2 pi i k (m fft_in - n fft_out) / (2 fft_in fft_out)}),  w_0 = 1, w_k = 2
// (H = the filter's spectrum; the inverse transform ignores the imaginary part of bin 0; new_len <= fft_out so the Nyquist bin never carries data).
// resample_matrix_kernel builds A once per rate pair in f64 (rotation recurrence over k from one sincospi per element, error ~ k * 2^-53), stored
// transposed At[n][m] in f32; resample_apply_kernel is the block matrix product + overlap-add + delay trim: one thread per output sample,
//     out[i] = sum_n At[n][m] x[c fft_in + n]  +  sum_n At[n][fft_out + m] x[(c - 1) fft_in + n],   (c, m) = divmod(i + delay, fft_out),
// the first sum being the block's own first half and the second the previous block's tail (rubato: output_buf[n] + overlap[n]); samples past the
// end of the input are zeros.  At stays in L2 (<= a few MB for every common rate); a 30 s clip is ~1 GFLOP of f32 FMAs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_matrix_kernel(const double* __restrict__ H, int new_len, int fft_in, int fft_out, float* __restrict__ At) {
    const long total = (long)fft_in * 2 * fft_out;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int n = (int)(e / (2 * fft_out)), m = (int)(e % (2 * fft_out));
    const long P = 2L * fft_in * fft_out;
    long d = ((long)m * fft_in - (long)n * fft_out) % P; if (d < 0) d += P;
    double s1, c1; sincospi(2.0 * (double)d / (double)P, &s1, &c1);
    double c = 1.0, s = 0.0, acc = H[0];
    for (int k = 1; k < new_len; k++) {
        const double cn = c * c1 - s * s1, sn = s * c1 + c * s1; c = cn; s = sn;
        acc += 2.0 * (H[2 * k] * c - H[2 * k + 1] * s);
    }
    At[e] = (float)acc;
}
__global__ __launch_bounds__(256) void resample_apply_kernel(const float* __restrict__ x, long n_in, const float* __restrict__ At, int fft_in, int fft_out, int delay,
                                                             float* __restrict__ out, long n_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const long j = i + delay, c = j / fft_out; const int m = (int)(j - c * fft_out);
    const int ld = 2 * fft_out;
    float a0 = 0.f, a1 = 0.f;
    {
        const long base = c * fft_in; const int cnt = (int)max(0L, min((long)fft_in, n_in - base));
        const float* xp = x + base; const float* ap = At + m;
        for (int n = 0; n < cnt; n++) a0 = fmaf(xp[n], ap[(size_t)n * ld], a0);
    }
    if (c > 0) {
        const long base = (c - 1) * fft_in; const int cnt = (int)max(0L, min((long)fft_in, n_in - base));
        const float* xp = x + base; const float* ap = At + fft_out + m;
        for (int n = 0; n < cnt; n++) a1 = fmaf(xp[n], ap[(size_t)n * ld], a1);
    }
    out[i] = a0 + a1;
}
hipError_t launch_resample_matrix(const double* H, int new_len, int fft_in, int fft_out, float* At, hipStream_t s) {
    const long total = (long)fft_in * 2 * fft_out;
    resample_matrix_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(H, new_len, fft_in, fft_out, At);
    return hipGetLastError();
}
hipError_t launch_resample(const float* x, long n_in, const float* At, int fft_in, int fft_out, int delay, float* out, long n_out, hipStream_t s) {
    if (n_out <= 0) return hipSuccess;
    resample_apply_kernel<<<dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s>>>(x, n_in, At, fft_in, fft_out, delay, out, n_out);
    return hipGetLastError();
}

__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
// test hook (vox_debug_occupy): workgroups that do nothing but hold their CUs for `ticks` s_memrealtime ticks (100 MHz)
__global__ __launch_bounds__(1024) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); n++; }
    if (sink && n == 0xFFFFFFFFu) *sink = n;
}
hipError_t launch_occupy(int workgroups, int micros, hipStream_t s) {
    occupy_kernel<<<dim3(workgroups), dim3(1024), 0, s>>>((unsigned long long)micros * 100ull, nullptr);
    return hipGetLastError();
}
hipError_t launch_add_rows(const float* a, const float* b, float* out, long n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    add_rows_kernel<<<dim3(blocks), dim3(256), 0, s>>>(a, b, out, n);
    return hipGetLastError();
}
__global__ void gelu_kernel(float* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = gelu_f(x[i]);
}
// argmax of every row of x [rows][V] (lowest index wins ties, NaN never wins): `logits.argmax(2)` of the piecewise decode loop (bin/e2e_bench.rs:219).  One
// 1024-thread workgroup per row, float4 loads; a 131 072-column row is 512 KB = a few microseconds.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ x, int V, int* __restrict__ out) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const float* row = x + (size_t)blockIdx.x * V;
    float v = -INFINITY; int idx = 0x7fffffff;
    const int V4 = ((reinterpret_cast<uintptr_t>(row) & 15) == 0) ? V >> 2 : 0;
    for (int i = threadIdx.x; i < V4; i += 1024) {
        const float4 q = reinterpret_cast<const float4*>(row)[i];
        const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 4; u++) if (e[u] > v) { v = e[u]; idx = 4 * i + u; }      // ascending index per thread: strict > keeps the lowest
    }
    for (int i = 4 * V4 + threadIdx.x; i < V; i += 1024) { const float e = row[i]; if (e > v || (e == v && i < idx)) { v = e; idx = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o); const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = v; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) if (bv[w] > v || (bv[w] == v && bi[w] < idx)) { v = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
    }
}
hipError_t launch_argmax_rows(const float* x, int rows, int V, int* out, hipStream_t s) {
    if (rows <= 0 || V <= 0) return hipErrorInvalidValue;
    argmax_rows_kernel<<<dim3(rows), dim3(1024), 0, s>>>(x, V, out);
    return hipGetLastError();
}
hipError_t launch_gelu(float* x, long n, hipStream_t s) {
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    gelu_kernel<<<dim3(blocks), dim3(256), 0, s>>>(x, n);
    return hipGetLastError();
}

}  // namespace vox
