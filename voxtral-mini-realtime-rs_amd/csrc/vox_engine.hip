// vox_engine.hip -- persistent decode-step engine for gfx950 (MI355X): one launch per decoded token.
//
// What it replaces: the 26 x 4 + 1 dependent launches of one single-stream decode step (reference: gguf/model.rs:938-960 -> forward_hidden_with_cache
// + lm_head, model.rs:566-691).  Why: a decode layer streams 65.5 MB of Q4 weights -- 10 us at HBM speed -- but every launch pays a kernel
// boundary, a ramp, and the round trip of its activation vector before the first weight byte is consumed.  Here the
// weight stream never stops: it does not depend on anything, so a loader wave per CU runs AHEAD of every dependency edge, and the edges
// themselves are 8-byte {value, tag} write-through granules swept by one wave per CU (MI355X_MICROARCH.md "Persistent kernels" price list).
//
// Geometry (fixed: the real Voxtral decoder -- D 3072, 32 query heads / 8 KV heads x 128, FFN 9216; other shapes keep the per-operator path):
//   grid = 256 workgroups (one per CU, all resident) x 512 threads = 8 waves:
//     wave 0      LOADER   global_load_lds_dwordx4 ... nt: this CU's slice of q|k|v, wo, w1|w3, w2 of every layer, then lm_head, as 21 KiB packets
//                          into a ring of LDS slots; 2-3 packets in flight; waits only for free slots.
//     wave 1      COMM     sweeps granules written by other CUs into LDS staging (activation vectors, partial sums), applies RMSNorm weights,
//                          reduces partial sums in a FIXED order (deterministic), publishes this CU's 12 rows of the residual stream.
//     waves 2..7  CONSUMERS  one "pass" (3456 B: 192 Q4 blocks) per wave per packet: v_cvt_pk_f32_fp8 turns two nibble bytes into two floats
//                          (an e4m3 byte 0x0q is exactly q * 2^-9), v_pk_fma_f32 against the activation slice held in REGISTERS; attention.
//   CU b = (g = b % 8: KV head / XCD, j = b / 8): query head h = 4 g + j / 8, slice s = j % 8.  Per layer:
//     q|k|v   CU computes q rows [128 h + 16 s, +16), k rows [4 j, +4) and v rows [4 j, +4) of KV head g (K = 3072, RMSNorm folded)   -> granules G
//     attn    CU (h, s) gathers q_h, k_g, v_g (new row) and runs head h's single-query attention over the cache (redundantly per slice)
//     wo      CU (h, s): rows [384 s, +384) x columns of head h (K = 128)  -> 32 partial planes PW; owner of rows [12 b, +12) sums them + residual -> H1
//     w1|w3   CU (g, j): SwiGLU outputs [1152 g + 36 j, +36) (K = 3072)     -> granules A (read inside the XCD group only)
//     w2      CU (g, j): rows [96 j, +96) x K slice [1152 g, +1152), split in 3 sub-slices over the consumer waves -> 24 partial planes P2;
//             owner sums + residual -> H0 (next layer's input)
//   Every all-to-all edge (H0, H1) is one 24 KB granule sweep per CU; the other edges are <= 1152 granules.
// Tags = launch serial * 64 + layer + 1: unique per (launch, layer), so no buffer is ever re-initialised and a stale granule can never match.
// Every spin is bounded (20 ms): on timeout the workgroup sets *err, marks itself dead and runs to completion without waiting.
#include "vox_kernels.h"

#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

namespace vox {
namespace {

typedef unsigned long long u64;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ED = 3072, ENH = 32, ENKV = 8, EHD = 128, EQD = ENH * EHD, EKD = ENKV * EHD, EF = 9216;
constexpr int NCU = 256, NCONS = 6, NWAVES = NCONS + 2, NTHR = 64 * NWAVES;
constexpr int PASS_A = 3456, PASS_WO = 2304;          // bytes per pass: 3 (2) planes of 64 x 16 B nibbles + 64 x 2 B scales
constexpr int PK_A = NCONS * PASS_A, PK_WO = NCONS * PASS_WO;   // packet = one pass per consumer wave: 20736 / 13824 bytes, stored back to back (no padding)
constexpr int LINES_A = (PK_A + 1023) / 1024, LINES_WO = (PK_WO + 1023) / 1024;      // LDS-DMA instructions per packet (the last one partial: 16 / 32 lanes)
static_assert(LINES_A == 21 && LINES_WO == 14, "wait_vmcnt() enumerates the in-flight line counts 14 / 21 / 28 / 35 / 42");
constexpr int SLOT_BYTES = PK_A, NSLOT = 6;            // w1|w3's six packets fit: with five slots its last pass waited for a refill (4 us tail per layer)
constexpr int QKV_PK = 2, WO_PK = 2, W13_PK = 6, W2_PK = 3;
constexpr int LAYER_BYTES = (QKV_PK + W13_PK + W2_PK) * PK_A + WO_PK * PK_WO;   // 255744 bytes per CU per layer = exactly the Q4 bytes
constexpr int OFF_QKV = 0, OFF_WO = QKV_PK * PK_A, OFF_W13 = OFF_WO + WO_PK * PK_WO, OFF_W2 = OFF_W13 + W13_PK * PK_A;   // byte offsets inside a layer
constexpr int SC_MAX = 1024;                          // attention scores in LDS: cache rows per KV head (max_seq) <= 1024
constexpr int OWN = ED / NCU;                         // 12 rows of the residual stream per CU
constexpr int NPW = ENH, NP2 = 24;                    // partial planes of wo / w2
constexpr u64 TIMEOUT_TICKS = 2000000;                // s_memrealtime ticks (100 MHz): 20 ms

enum { EOP_QKV = 0, EOP_WO = 1, EOP_W13 = 2, EOP_W2 = 3, EOP_LM = 4 };
enum { ERR_RING = 1, ERR_STAGE = 2, ERR_SWEEP = 3, ERR_CBAR = 4, ERR_SLOT = 5 };

__host__ __device__ inline int lm_rows_per_cu(int vocab) { return vocab / NCU; }
__host__ __device__ inline int lm_passes(int vocab) { return lm_rows_per_cu(vocab) / 2; }
__host__ __device__ inline int lm_packets(int vocab) { return (lm_passes(vocab) + NCONS - 1) / NCONS; }
__host__ __device__ inline size_t cu_stream_bytes(int n_layers, int vocab) { return (size_t)n_layers * LAYER_BYTES + (size_t)lm_packets(vocab) * PK_A + 1024; }      // + 1 KiB: the stream is read in whole 16-byte lanes only, the pad keeps the allocation comfortable

// (row, block) of weight matrix `op` that lands in 16-byte chunk [plane p][lane] of pass q on CU b.  Lanes that split one row hold whole Q4 blocks.
__host__ __device__ inline void eng_src(int op, int b, int q, int p, int lane, int vocab, int* row, int* blk) {
    const int g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7;
    if (op == EOP_QKV || op == EOP_W13 || op == EOP_LM) {              // 2 rows per pass, 32 lanes x 3 blocks per row
        const int hr = lane >> 5, li = lane & 31;
        *blk = li + 32 * p;
        if (op == EOP_W13) *row = 2 * (1152 * g + 36 * j + q) + hr;
        else if (op == EOP_LM) *row = lm_rows_per_cu(vocab) * b + 2 * q + hr;
        else if (q < 8) *row = 128 * h + 16 * s + 2 * q + hr;
        else if (q < 10) *row = EQD + 128 * g + 4 * j + 2 * (q - 8) + hr;
        else *row = EQD + EKD + 128 * g + 4 * j + 2 * (q - 10) + hr;
    } else if (op == EOP_WO) {                                         // 32 rows per pass, 2 lanes x 2 blocks per row (K = the head's 128 columns)
        *row = 384 * s + 32 * q + (lane >> 1);
        *blk = 4 * h + (lane & 1) + 2 * p;
    } else {                                                           // w2: 16 rows per pass, 4 lanes x 3 blocks; wave w = q % 6: sub-slice w % 3, row half w / 3
        const int t3 = q / NCONS, w = q % NCONS, ts = w % 3, rh = w / 3;
        *row = 96 * j + 48 * rh + 16 * t3 + (lane >> 2);
        *blk = 36 * g + 12 * ts + (lane & 3) + 4 * p;
    }
}

__global__ __launch_bounds__(192) void eng_pack_kernel(Q4W w, int op, unsigned char* __restrict__ stream, size_t cu_stride, size_t op_off, int vocab) {
    const int q = blockIdx.x, b = blockIdx.y, t = threadIdx.x, NB = op == EOP_WO ? 2 : 3;
    if (t >= NB * 64) return;
    const int p = t >> 6, lane = t & 63;
    int row, blk; eng_src(op, b, q, p, lane, vocab, &row, &blk);
    const size_t pk_bytes = op == EOP_WO ? PK_WO : PK_A, pass_bytes = op == EOP_WO ? PASS_WO : PASS_A;
    // packet-major: packet k of all 256 CUs is contiguous ([k][cu][bytes]) -- at any moment the 256 loaders read one contiguous ~5 MB window, spread over
    // every HBM channel (CU-major streams 7.3 MB apart put all loaders on the same channels at the same time)
    (void)cu_stride;
    unsigned char* dst = stream + (size_t)NCU * (op_off + (size_t)(q / NCONS) * pk_bytes) + (size_t)b * pk_bytes + (size_t)(q % NCONS) * pass_bytes;
    const size_t src = (size_t)row * w.nb + blk;
    reinterpret_cast<uint4*>(dst)[p * 64 + lane] = w.qs[src];
    reinterpret_cast<uint16_t*>(dst + NB * 1024)[p * 64 + lane] = w.sc[src];
}

// ------------------------------------------------------------------------------------------------
// LDS map
// ------------------------------------------------------------------------------------------------
struct EngCtl {
    unsigned ring_ready[8], ring_done[8];         // monotonic per slot: fills landed / passes consumed
    unsigned xs0_flag, xs1_flag, xa_flag, qkv_flag;   // layer + 1 of the staged content (monotonic)
    unsigned cbar, dead, gathering, gw_flag;
    unsigned xcd_ok, xcc_id, pad3[2];
    float rstd0, rstd1, pad1, pad2;
    float best_val[8]; int best_idx[8];
    float h_own[16], h1_own[16];
};
constexpr int L_RING = 0;
constexpr int L_XS = L_RING + NSLOT * SLOT_BYTES;       // [3072] f32, swizzled chunks: the all-gathered input of q|k|v / w1|w3 / lm_head.  ONE buffer: it is re-staged only
constexpr int L_XS0 = L_XS, L_XS1 = L_XS;               //   after every consumer wave has published results computed from the previous content (registers hold x during an operator)
constexpr int L_U = L_XS + ED * 4;                      // time-shared: the XCD group's 1152 SwiGLU outputs (w2 input) | attention scratch of the NEXT layer
constexpr int L_XA = L_U;                               //   [1152] staged after w1|w3, loaded to registers at the start of w2
constexpr int L_SC = L_U;                               //   [SC_MAX] scores
constexpr int L_PO = L_SC + SC_MAX * 4;                 //   [12][128] partial attention outputs
constexpr int L_PL = L_PO + 12 * 128 * 4;               //   [16] partial softmax sums
constexpr int L_XO = L_PL + 64;                         // [128] attention output of head h (wo input)
constexpr int L_QKVN = L_XO + 128 * 4;                  // q_h[128] k_g[128] v_g[128] of this step (plain order)
constexpr int L_TMP = L_QKVN + 384 * 4;                 // [384] partial sums swept by the comm wave
constexpr int L_TAB = L_TMP + 384 * 4;                  // [MAX_LAYERS] copy of the layer table: pointer reads never touch VMEM (a vector load behind a publish waits for the store)
constexpr int MAX_LAYERS = 32;
constexpr int L_GW = L_TAB + MAX_LAYERS * (int)sizeof(EngLayerTab);      // [MAX_LAYERS + 1][2][16] norm weight * 512 of this CU's 12 rows: [l][0] attn_norm (l = L: final norm), [l][1] ffn_norm * Ada
constexpr int L_CTL = L_GW + (MAX_LAYERS + 1) * 32 * 4;
constexpr int L_TOTAL = L_CTL + (int)sizeof(EngCtl);
static_assert(L_TOTAL <= 160 * 1024, "LDS budget");
static_assert(L_TAB % 16 == 0 && sizeof(EngLayerTab) == 40, "layer table");
static_assert(L_XS % 16 == 0 && L_XA % 16 == 0 && L_XO % 16 == 0 && L_QKVN % 16 == 0 && L_PO % 16 == 0 && L_CTL % 16 == 0 && 1152 * 4 <= SC_MAX * 4 + 12 * 128 * 4, "16-byte aligned carve");

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
#define ENG_CFENCE() asm volatile("" ::: "memory")
typedef const __attribute__((address_space(1))) float* gcf_p;      // pointers that come out of the device-resident layer table: the compiler cannot infer
typedef __attribute__((address_space(1))) float* gf_p;             // their address space, and a FLAT load also counts on lgkmcnt (an LDS wait would wait for it)
typedef float fv4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gcf_p as_g(const float* p) { return (gcf_p)(uintptr_t)p; }
__device__ __forceinline__ gf_p as_g(float* p) { return (gf_p)(uintptr_t)p; }
__device__ __forceinline__ float4 ldg4(gcf_p p) { const fv4 v = *(const __attribute__((address_space(1))) fv4*)p; return make_float4(v.x, v.y, v.z, v.w); }
#define RLX __ATOMIC_RELAXED
// granule / table loads are addressed as (wave-uniform base in SGPRs) + (32-bit byte offset in ONE VGPR): with 64-bit per-lane pointers a 48-load
// sweep carries 80 address VGPRs around its retry loop
#define WG __HIP_MEMORY_SCOPE_WORKGROUP
#define AG __HIP_MEMORY_SCOPE_AGENT
// control words are wave-uniform: readfirstlane keeps every branch on them a scalar branch (all 64 lanes stay active for the DPP reductions)
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
// granule load: buffer_load_dwordx2 ... offen sc1 (aux 16 = sc1: served by L2 / memory, never by this CU's L1)
__device__ __forceinline__ u64 ld_gran(srd_t srd, unsigned idx) { const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(srd, (int)(idx * 8u), 0, 16); return ((u64)v.y << 32) | (u64)v.x; }
__device__ __forceinline__ float ld_gf(srd_t srd, unsigned idx) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, (int)(idx * 4u), 0, 0)); }
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, RLX, WG)); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, RLX, WG); }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float rlf(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float row16_sum_e(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); v += dppf<0x140>(v); return v; }
__device__ __forceinline__ float row16_max_e(float v) { v = fmaxf(v, dppf<0xB1>(v)); v = fmaxf(v, dppf<0x4E>(v)); v = fmaxf(v, dppf<0x141>(v)); v = fmaxf(v, dppf<0x140>(v)); return v; }
__device__ __forceinline__ float wave_sum_e(float v) { v = row16_sum_e(v); return (rlf(v, 0) + rlf(v, 16)) + (rlf(v, 32) + rlf(v, 48)); }
__device__ __forceinline__ float wave_max_e(float v) { v = row16_max_e(v); return fmaxf(fmaxf(rlf(v, 0), rlf(v, 16)), fmaxf(rlf(v, 32), rlf(v, 48))); }
__device__ __forceinline__ float group8_sum_e(float v) { v += dppf<0xB1>(v); v += dppf<0x4E>(v); v += dppf<0x141>(v); return v; }
__device__ __forceinline__ float silu_e(float x) { return x / (1.0f + expf(-x)); }

// staged activation vectors: chunk c (32 floats) keeps its eight 16-byte pieces at piece index j ^ ((c >> 1) & 7), so the ds_read_b128 of lanes that
// hold consecutive chunks is bank-conflict free (the same swizzle as q4_gemv_kernel's)
__device__ __forceinline__ int sw_piece(int c, int j) { return c * 8 + (j ^ ((c >> 1) & 7)); }
__device__ __forceinline__ int sw_dword(int k) { const int c = k >> 5, e = k & 31; return sw_piece(c, e >> 2) * 4 + (e & 3); }

struct Tl {      // timeline stamps (measurement runs: p.tl != nullptr), lane 0 of the stamping wave
    u64* buf; bool on;
    __device__ __forceinline__ void operator()(int evt) const { if (on) buf[evt] = wall_clock64(); }
};

// wait until *word >= target (LDS word, monotonic).  Bounded; a dead workgroup never waits.
__device__ __forceinline__ bool wait_ge(unsigned* word, unsigned target, EngCtl* c, unsigned* err, unsigned code) {
    if (lds_ld(word) >= target) { ENG_CFENCE(); return true; }
    if (lds_ld(&c->dead)) return false;
    const u64 t0 = wall_clock64();
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_ld(word) >= target) break;
        if (lds_ld(&c->dead)) return false;
        if (wall_clock64() - t0 > TIMEOUT_TICKS) {
            lds_st(&c->dead, 1u);
            __hip_atomic_store(err, code | ((unsigned)blockIdx.x << 8), RLX, AG);
            return false;
        }
    }
    ENG_CFENCE();
    return true;
}

// ------------------------------------------------------------------------------------------------
// LOADER wave
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma_line(unsigned voff, unsigned lds_dst_, u64 gsrc_) {
    unsigned keep;
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_);      // wave-uniform by construction; make the compiler see it
    const u64 gsrc = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(gsrc_ >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gsrc_);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gsrc) : "memory");
}
__device__ __forceinline__ void wait_vmcnt(int n) {      // n = DMA lines allowed to stay in flight (younger packets)
    switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
    case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
    case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

struct Loader {
    EngCtl* c; unsigned* err; unsigned ring_lds; unsigned voff;
    int nfl = 0, s0 = 0, l0 = 0, s1 = 0, l1 = 0;    // packets issued, not yet published: (s0, l0) oldest, (s1, l1) newer -- plain scalars (an indexed array would live in scratch = VMEM)
    unsigned P = 0;                                   // next packet index
    bool thin; u64 pace = 0, t_last = 0, pause_ticks = 0;              // pace: minimum s_memrealtime ticks between two packet issues (0: none)
    __device__ __forceinline__ void publish_slot(int slot) { lds_st(&c->ring_ready[slot], lds_ld(&c->ring_ready[slot]) + 1u); }   // only this wave writes ring_ready
    __device__ __forceinline__ void flush() {
        if (nfl == 2) { wait_vmcnt(l1); publish_slot(s0); s0 = s1; l0 = l1; nfl = 1; }
        if (nfl == 1) { wait_vmcnt(0); publish_slot(s0); nfl = 0; }
    }
    __device__ __forceinline__ void issue(u64 gsrc, int bytes, int lane, bool nodma) {
        const int full = bytes >> 10, tail = (bytes & 1023) >> 4, lines = full + (tail ? 1 : 0);      // tail: lanes of the last, partial LDS-DMA instruction
        const int slot = (int)(P % NSLOT); const unsigned k = P / NSLOT;
        if (k > 0 && lds_ld(&c->ring_done[slot]) < NCONS * k) {
            flush();                                   // publish what has landed before blocking: the consumers may be waiting for exactly that
            wait_ge(&c->ring_done[slot], NCONS * k, c, err, ERR_SLOT);
        }
        if (thin && lds_ld(&c->gathering)) flush();    // one fill outstanding while this CU's comm wave sweeps (MI355X_MICROARCH.md gather-pass)
        if (pause_ticks && lds_ld(&c->gathering)) {    // nothing new in flight while this CU's comm wave waits on an edge (bounded: never a deadlock)
            flush();
            const u64 tp = wall_clock64();
            while (lds_ld(&c->gathering) && wall_clock64() - tp < pause_ticks) __builtin_amdgcn_s_sleep(2);
        }
        if (pace) { while (wall_clock64() - t_last < pace) __builtin_amdgcn_s_sleep(1); t_last = wall_clock64(); }
        const unsigned dst = ring_lds + (unsigned)slot * SLOT_BYTES;
        if (!nodma) {
#pragma unroll 1
            for (int i = 0; i < full; i++) dma_line(voff, dst + (unsigned)i * 1024u, gsrc + (u64)i * 1024u);
            if (lane < tail) dma_line(voff, dst + (unsigned)full * 1024u, gsrc + (u64)full * 1024u);      // EXEC-masked: only `tail` lanes write
        }
        P++;
        if (nfl == 2) { wait_vmcnt(l1 + lines); publish_slot(s0); s0 = s1; l0 = l1; s1 = slot; l1 = lines; }      // three in flight: retire the oldest
        else if (nfl == 1) { s1 = slot; l1 = lines; nfl = 2; }
        else { s0 = slot; l0 = lines; nfl = 1; }
    }
};

// ONE rolled loop over the step's packets (the kernel's code must stay small: the instruction cache is shared by two CUs and every layer walks
// through all three roles' code -- a 100 KB kernel ran 13 % slower than a 68 KB one with the same structure)
__device__ __forceinline__ void eng_loader(const EngParams& p, EngCtl* c, unsigned ring_lds, int lane, const Tl& tl) {
    Loader ld; ld.c = c; ld.err = p.err; ld.ring_lds = ring_lds; ld.voff = (unsigned)lane * 16u; ld.thin = (p.flags & 1) != 0; ld.pace = (u64)p.pace_ticks;
    const bool fake = (p.flags & 2) != 0;      // diagnostic: every packet re-reads one packet (L2 hits, no HBM traffic; results wrong)
    const bool nodma = (p.flags & 32) != 0;    // diagnostic: no LDS-DMA at all inside the layers (results wrong)
    if (p.flags & 64) ld.pause_ticks = 300;
    const u64 base = (u64)p.stream;
    constexpr int PK_LAYER = QKV_PK + WO_PK + W13_PK + W2_PK;
    const unsigned n_layer_pk = (unsigned)p.n_layers * PK_LAYER, n_pk = n_layer_pk + (unsigned)lm_packets(p.vocab);
    unsigned l = 0, r = 0;                       // layer, packet within the layer
    u64 off = 0;                                 // byte offset of the next packet in this CU's stream (packets are stored in consumption order)
#pragma unroll 1
    for (unsigned pk = 0; pk < n_pk; pk++) {
        int bytes = PK_A;
        if (pk < n_layer_pk) {
            if (r >= QKV_PK && r < QKV_PK + WO_PK) bytes = PK_WO;
            if (r == 0 && (int)l == p.tl_layer) tl(16);
        } else { ld.pace = 0; ld.pause_ticks = 0; }      // no edge left to protect: the lm_head streams at full depth
        ld.issue(fake ? base + (u64)blockIdx.x * PK_A : base + (u64)NCU * off + (u64)blockIdx.x * (u64)bytes, bytes, lane, nodma);
        off += (u64)bytes;
        if (++r == PK_LAYER) { if ((int)l == p.tl_layer) tl(17); r = 0; l++; }
    }
    ld.flush();
    tl(18);
}

// ------------------------------------------------------------------------------------------------
// COMM wave
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool sweep_bail(u64& t0, unsigned tag, EngCtl* c, unsigned* err) {
    if (lds_ld(&c->dead)) return true;
    if (t0 == 0) t0 = wall_clock64();
    else if (wall_clock64() - t0 > TIMEOUT_TICKS) {
        lds_st(&c->dead, 1u);
        __hip_atomic_store(err, (unsigned)ERR_SWEEP | ((unsigned)blockIdx.x << 8) | (tag << 16), RLX, AG);
        return true;
    }
    __builtin_amdgcn_s_sleep(2);
    return false;
}
// Sweep N granules per lane until every tag matches (values in v).  Bounded.  With `do_probe` the wave first polls ONE granule per lane (`probe()`:
// one granule of every producer, or of every n-th) instead of the whole set: 256 CUs polling 8 KB each would put TB/s of coherent reads next to the
// weight stream (MI355X_MICROARCH.md polling-cost).
template <int N, class IdxF, class ProbeF>
__device__ __forceinline__ bool sweep(const u64* base_, unsigned bytes, unsigned tag, IdxF idx, ProbeF probe, bool do_probe, float (&v)[N], EngCtl* c, unsigned* err) {
    const srd_t base = make_srd(base_, bytes);
    u64 t0 = 0;
    if (do_probe) {
        const unsigned pi = (unsigned)probe();
        for (;;) {
            const u64 gq = ld_gran(base, pi);
            if (__all((unsigned)(gq >> 32) == tag)) break;
            if (sweep_bail(t0, tag, c, err)) return false;
        }
    }
    for (;;) {      // (two polls in flight, half a round trip apart, were measured SLOWER on every edge: the extra coherent reads cost more than the earlier detection buys)
        bool ok = true;
#pragma unroll
        for (int u = 0; u < N; u++) {
            const u64 gq = ld_gran(base, (unsigned)idx(u));
            v[u] = __uint_as_float((unsigned)gq);
            ok &= (unsigned)(gq >> 32) == tag;
        }
        if (__all(ok)) return true;
        if (sweep_bail(t0, tag, c, err)) return false;
    }
}
__device__ __forceinline__ void publish(u64* g, unsigned tag, float v) { __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), RLX, AG); }
// XCD-local edge (producer and every reader on the same XCD -- checked at kernel start, see xcd_local): a PLAIN 8-byte store stays in the XCD's L2, where the
// readers' sc1 loads (which bypass only their own L1) find it at L2-hit latency; the write-through (sc1) form is served at the cross-XCD rate
// consumer-side publishes: buffer stores (uniform base in SGPRs + one 32-bit VGPR index -- a 64-bit per-lane pointer costs two VGPRs the consumer does not have)
__device__ __forceinline__ void publish_b(const u64* base, unsigned bytes, unsigned idx, unsigned tag, float v, bool local) {
    const srd_t srd = make_srd(base, bytes);
    v2u_t x; x.x = __float_as_uint(v); x.y = tag;
    if (local) __builtin_amdgcn_raw_buffer_store_b64(x, srd, (int)(idx * 8u), 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(x, srd, (int)(idx * 8u), 0, 16);      // aux 16 = sc1: write-through
}

// All-gather of a staged activation vector: the owners publish their 12 rows ALREADY multiplied by the consumer's norm weight (* Ada scale) * 512,
// plus one partial sum of squares per CU, so the sweep is granules -> LDS with no other memory operand (the per-layer norm vectors take microseconds to
// arrive and would sit in front of the granule loads: VMEM returns in order).  48 + 4 granules per lane; the first chunk starts with a probe of one
// row of every 4th producer.
__device__ __forceinline__ void comm_stage_x(const EngParams& p, EngCtl* c, int lane, const u64* src, const u64* ssq, unsigned tag, float* xs, float* rstd_out, const Tl& tl, bool T) {
    asm volatile("" : "+v"(lane));      // opaque per call: swizzled staging addresses are computed where they are used, not carried in VGPRs
    constexpr int NU = ED / 64;
    const srd_t sd = make_srd(src, ED * 8u), qd = make_srd(ssq, NCU * 8u);
    if (T) tl(20);
    u64 t0 = 0;
    if (p.flags & 512) {      // no probe: the owners publish within a fraction of a microsecond of each other, so wait out the store-to-visibility latency once
        const u64 tw = wall_clock64();      // and go straight for the full sweep (a miss costs one more round trip)
        while (wall_clock64() - tw < (u64)p.ag_delay_ticks) __builtin_amdgcn_s_sleep(1);
    } else
    for (;;) {      // probe: one row of every 4th producer
        const u64 gq = ld_gran(sd, 48u * (unsigned)lane);
        if (__all((unsigned)(gq >> 32) == tag)) break;
        if (sweep_bail(t0, tag, c, p.err)) break;
    }
    if (T) tl(21);
    u64 raw[NU], rq[4];
    for (;;) {      // all 48 + 4 granules of this lane in ONE round trip (three dependent 16-load round trips cost ~2 us more per all-gather)
        bool ok = true;
#pragma unroll
        for (int u = 0; u < NU; u++) raw[u] = ld_gran(sd, (unsigned)lane + 64u * u);
#pragma unroll
        for (int u = 0; u < 4; u++) rq[u] = ld_gran(qd, (unsigned)lane + 64u * u);
#pragma unroll
        for (int u = 0; u < NU; u++) ok &= (unsigned)(raw[u] >> 32) == tag;
#pragma unroll
        for (int u = 0; u < 4; u++) ok &= (unsigned)(rq[u] >> 32) == tag;
        if (__all(ok)) break;
        if (sweep_bail(t0, tag, c, p.err)) break;
    }
    if (T) tl(22);
    // element k = lane + 64 u lives in chunk c = 2 u + (lane >> 5), piece (lane & 31) >> 2, and the chunk's swizzle is (c >> 1) & 7 = u & 7: eight lane-dependent
    // byte offsets (one per value of u & 7) + 256 u cover all 48 stores -- ds_write_b32 with immediate offsets instead of 48 address computations
    {
        unsigned a8[8];
        const unsigned jj = (unsigned)(lane & 31) >> 2, rr = (unsigned)lane & 3u, hi = (unsigned)lane >> 5;
#pragma unroll
        for (int m8 = 0; m8 < 8; m8++) a8[m8] = ((hi * 8u + (jj ^ (unsigned)m8)) * 4u + rr) * 4u;
        unsigned char* xb = reinterpret_cast<unsigned char*>(xs);
#pragma unroll
        for (int u = 0; u < NU; u++) *reinterpret_cast<float*>(xb + a8[u & 7] + 256u * (unsigned)u) = __uint_as_float((unsigned)raw[u]);
    }
    const float ss = wave_sum_e((__uint_as_float((unsigned)rq[0]) + __uint_as_float((unsigned)rq[1])) + (__uint_as_float((unsigned)rq[2]) + __uint_as_float((unsigned)rq[3])));      // fixed order: bit-identical on every CU
    if (lane == 0) *rstd_out = 1.0f / sqrtf(ss / (float)ED + p.eps);
    if (T) tl(23);
}
// the owner's side: rows [12 b, +12) of a residual stream -> granules of row * (next norm weight) * 512, the CU's partial sum of squares, raw rows kept in LDS
__device__ __forceinline__ void comm_publish_rows(const EngParams& p, int lane, float hraw, float gw, u64* dst, u64* ssq, unsigned tag, float* own) {
    const int b = blockIdx.x;
    float sq = lane < OWN ? hraw * hraw : 0.f;
    sq = row16_sum_e(sq);                                   // lanes 0..11 live in the first DPP row
    if (lane < OWN) { publish(dst + OWN * b + lane, tag, hraw * gw); own[lane] = hraw; }
    if (lane == 0) publish(ssq + b, tag, sq);
}

// ONE rolled loop over the 2 L + 1 all-gathers of the step: stage 2 l = layer l's input (-> q|k|v), stage 2 l + 1 = its post-attention stream (-> w1|w3),
// stage 2 L = the final norm's input (-> lm_head); the small edges that follow each all-gather hang off the loop body.
__device__ __forceinline__ void eng_comm(const EngParams& p, EngCtl* c, unsigned char* lds, const int lane0, const Tl& tl) {
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3);
    float* xs0 = reinterpret_cast<float*>(lds + L_XS0); float* xs1 = reinterpret_cast<float*>(lds + L_XS1);
    float* xa = reinterpret_cast<float*>(lds + L_XA); float* qkvn = reinterpret_cast<float*>(lds + L_QKVN);
    float* tmp = reinterpret_cast<float*>(lds + L_TMP);
    const EngLayerTab* tab = reinterpret_cast<const EngLayerTab*>(lds + L_TAB);
    const unsigned tag_base = (*p.serial + 1u) * 64u;     // tags of this launch: tag_base (the step's input) .. tag_base + n_layers; never 0, never reused
    const bool PROBE_SMALL = (p.flags & 4) != 0;          // small edges (<= 18 granules per lane) are polled with the sweep itself unless this is set
    const int own_r = min(lane0, OWN - 1);
    const unsigned own_k = (unsigned)(OWN * b + own_r);
    const float* gwt = reinterpret_cast<const float*>(lds + L_GW);      // filled by consumer wave 5 while everybody waits for the first all-gather
    // the step's input joins the granule protocol: every CU publishes its 12 rows of h_in, so layer 0 takes the same all-gather as every other layer
    {
        const float g0 = ld_gf(make_srd(p.n_layers > 0 ? tab[0].attn_norm : p.final_norm, ED * 4u), own_k) * 512.0f;
        comm_publish_rows(p, lane0, ld_gf(make_srd(p.h_in, ED * 4u), own_k), g0, p.H0, p.SS0, tag_base, c->h_own);
    }
#pragma unroll 1
    for (int st = 0; st <= 2 * p.n_layers; st++) {
        int lane = lane0; asm volatile("" : "+v"(lane));      // opaque per stage: lane-derived addresses are recomputed, not carried around the loop in VGPRs
        const int l = st >> 1; const bool odd = st & 1, last = st == 2 * p.n_layers, T = l == p.tl_layer;
        const unsigned tag = tag_base + (unsigned)l + 1u;       // written during layer l
        lds_st(&c->gathering, 1u);
        if (odd) comm_stage_x(p, c, lane, p.H1, p.SS1, tag, xs1, &c->rstd1, tl, T);
        else comm_stage_x(p, c, lane, p.H0, p.SS0, tag - 1u, xs0, &c->rstd0, tl, false);
        ENG_CFENCE(); lds_st(odd ? &c->xs1_flag : &c->xs0_flag, (unsigned)l + 1u);
        if (T) tl(odd ? 11 : 8);
        if (last) { lds_st(&c->gathering, 0u); break; }
        if (!odd) {
            {   // this step's q_h, k_g, v_g rows
                float v[6];
                sweep<6>(p.G, (EQD + 2 * EKD) * 8u, tag, [&](int u) { const int i = lane + 64 * u, seg = i >> 7, e = i & 127; return seg == 0 ? 128 * h + e : (seg == 1 ? EQD : EQD + EKD) + 128 * g + e; },
                         [&]() { return lane < 32 ? EQD + 128 * g + 4 * lane : EQD + EKD + 128 * g + 4 * (lane - 32); }, PROBE_SMALL, v, c, p.err);      // probe: a k / v row of each of the group's 32 CUs
#pragma unroll
                for (int u = 0; u < 6; u++) qkvn[lane + 64 * u] = v[u];
                ENG_CFENCE(); lds_st(&c->qkv_flag, (unsigned)l + 1u);
            }
            if (T) tl(9);
            {   // wo: 32 partial planes of this CU's 12 rows -> residual stream after attention, published as the w1|w3 input (gw = ffn_norm * Ada * 512)
                float v[6];
                sweep<6>(p.PW, NPW * ED * 8u, tag, [&](int u) { const int i = lane + 64 * u, hh = i / OWN, r = i - hh * OWN; return hh * ED + OWN * b + r; }, [&]() { return (lane & 31) * ED + OWN * b; }, PROBE_SMALL, v, c, p.err);
#pragma unroll
                for (int u = 0; u < 6; u++) tmp[lane + 64 * u] = v[u];
                ENG_CFENCE();
                float a = 0.f;
                const int r = min(lane, OWN - 1);
#pragma unroll 4
                for (int hh = 0; hh < NPW; hh++) a += tmp[hh * OWN + r];      // fixed order
                wait_ge(&c->gw_flag, 1u, c, p.err, ERR_STAGE);
                comm_publish_rows(p, lane, c->h_own[r] + a, gwt[(l * 2 + 1) * 16 + r], p.H1, p.SS1, tag, c->h1_own);
            }
            if (T) tl(10);
        } else {
            {   // the XCD group's 1152 SwiGLU outputs -> w2 input
                float v[18];
                sweep<18>(p.A, EF * 8u, tag, [&](int u) { return 1152 * g + lane + 64 * u; }, [&]() { return 1152 * g + 36 * (lane & 31) + 35; }, PROBE_SMALL, v, c, p.err);      // probe: the last output of each CU of the group
#pragma unroll
                for (int u = 0; u < 18; u++) xa[sw_dword(lane + 64 * u)] = v[u] * 512.0f;
                ENG_CFENCE(); lds_st(&c->xa_flag, (unsigned)l + 1u);
            }
            if (T) tl(12);
            {   // w2: 24 partial planes of this CU's 12 rows -> the layer's output, published as the next layer's q|k|v input (gw = next attn_norm * 512)
                float v[5];
                sweep<5>(p.P2, NP2 * ED * 8u, tag, [&](int u) { const int i = min(lane + 64 * u, NP2 * OWN - 1), pp = i / OWN, r = i - pp * OWN; return pp * ED + OWN * b + r; }, [&]() { return min(lane, NP2 - 1) * ED + OWN * b; }, PROBE_SMALL, v, c, p.err);
#pragma unroll
                for (int u = 0; u < 5; u++) if (lane + 64 * u < NP2 * OWN) tmp[lane + 64 * u] = v[u];
                ENG_CFENCE();
                float a = 0.f;
                const int r = min(lane, OWN - 1);
#pragma unroll 4
                for (int pp = 0; pp < NP2; pp++) a += tmp[pp * OWN + r];      // fixed order
                comm_publish_rows(p, lane, c->h1_own[r] + a, gwt[((l + 1) * 2) * 16 + r], p.H0, p.SS0, tag, c->h_own);
            }
            if (T) tl(13);
        }
        lds_st(&c->gathering, 0u);
    }
}

// ------------------------------------------------------------------------------------------------
// CONSUMER waves
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f2 cvt2(unsigned w, bool hi) { return hi ? __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false); }
// sum_k x'[k] * (q_k / 512) over one Q4_0 block (x' = 512 x: the staged vector is pre-scaled, so this is sum x q exactly as in f32); `init` = -8 sum x
__device__ __forceinline__ float block_dot(const uint4 q, const f2* __restrict__ x, float init) {
    f2 a0 = {init, 0.f}, a1 = {0.f, 0.f};
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const unsigned lo = w[d] & 0x0F0F0F0Fu, hi = (w[d] >> 4) & 0x0F0F0F0Fu;
        a0 = __builtin_elementwise_fma(cvt2(lo, false), x[2 * d], a0);
        a1 = __builtin_elementwise_fma(cvt2(lo, true), x[2 * d + 1], a1);
        a0 = __builtin_elementwise_fma(cvt2(hi, false), x[8 + 2 * d], a0);
        a1 = __builtin_elementwise_fma(cvt2(hi, true), x[8 + 2 * d + 1], a1);
    }
    return (a0.x + a0.y) + (a1.x + a1.y);
}

template <int NB>
struct XRegs {
    f2 x[NB][16]; float m8[NB];
    // chunk(p) = first + step * p of the staged vector xs
    __device__ __forceinline__ void load(const float* xs, int first, int step) {
        asm volatile("" : "+v"(first));      // opaque per call: keeps the 8 swizzled piece addresses out of the loop invariants (they cost ~100 VGPRs hoisted)
        const float4* x4 = reinterpret_cast<const float4*>(xs);
#pragma unroll
        for (int p = 0; p < NB; p++) {
            const int cidx = first + step * p;
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const float4 v = x4[sw_piece(cidx, jj)];
                x[p][2 * jj] = f2{v.x, v.y}; x[p][2 * jj + 1] = f2{v.z, v.w};
                s += (v.x + v.y) + (v.z + v.w);
            }
            m8[p] = s * (-1.0f / 64.0f);        // -8 * sum x = -(8 / 512) * sum x'
        }
    }
};

struct Cons {
    const EngParams& p; EngCtl* c; unsigned char* lds; int cw, lane; unsigned P; unsigned cbar_n;
    __device__ __forceinline__ Cons(const EngParams& p_, EngCtl* c_, unsigned char* lds_, int cw_, int lane_) : p(p_), c(c_), lds(lds_), cw(cw_), lane(lane_), P(0), cbar_n(0) {}
    __device__ __forceinline__ void cbarrier() {
        cbar_n += NCONS;
        ENG_CFENCE();
        if (lane == 0) __hip_atomic_fetch_add(&c->cbar, 1u, RLX, WG);
        wait_ge(&c->cbar, cbar_n, c, p.err, ERR_CBAR);
    }
    // fetch this wave's pass of packet pk into registers and release the slot; `real` false: only release (a packet with fewer passes)
    template <int NB>
    __device__ __forceinline__ void fetch(unsigned pk, uint4 (&Q)[NB], float (&S)[NB], bool real) {
        const int slot = (int)(pk % NSLOT); const unsigned k = pk / NSLOT;
        wait_ge(&c->ring_ready[slot], k + 1u, c, p.err, ERR_RING);
        if (real) {
            const unsigned char* base = lds + L_RING + slot * SLOT_BYTES + cw * (NB == 2 ? PASS_WO : PASS_A);
#pragma unroll
            for (int i = 0; i < NB; i++) Q[i] = reinterpret_cast<const uint4*>(base)[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < NB; i++) S[i] = __half2float(__ushort_as_half(reinterpret_cast<const unsigned short*>(base + NB * 1024)[i * 64 + lane]));
        }
        ENG_CFENCE();      // the LDS pipeline executes a wave's instructions in order: the reads above have been served when this add is
        if (lane == 0) __hip_atomic_fetch_add(&c->ring_done[slot], 1u, RLX, WG);
    }
    template <int NB>
    __device__ __forceinline__ float pass_dot(const uint4 (&Q)[NB], const float (&S)[NB], const XRegs<NB>& xr) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < NB; i++) acc = fmaf(S[i], block_dot(Q[i], xr.x[i], xr.m8[i]), acc);
        return acc;
    }
};

// ONE rolled loop over the step's 4 L + 1 operators (one copy of the pass body in the instruction cache): item 4 l + op, op in EOP_* order
// (q|k|v, attention + wo, w1|w3, w2), then the lm_head.  Passes are software-pipelined: the next pass's weights are requested from the ring (LDS)
// before the current pass is multiplied.
__device__ __forceinline__ void eng_consumer(const EngParams& p, EngCtl* c, unsigned char* lds, int cw, const int lane0, const Tl& tl) {
    Cons cs(p, c, lds, cw, lane0);
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7;
    const float* xs0 = reinterpret_cast<const float*>(lds + L_XS0); const float* xs1 = reinterpret_cast<const float*>(lds + L_XS1);
    const float* xa = reinterpret_cast<const float*>(lds + L_XA); float* xo = reinterpret_cast<float*>(lds + L_XO);
    const float* qkvn = reinterpret_cast<const float*>(lds + L_QKVN);
    float* sc = reinterpret_cast<float*>(lds + L_SC); float4* po = reinterpret_cast<float4*>(lds + L_PO); float* pl = reinterpret_cast<float*>(lds + L_PL);
    const unsigned tag_base = (*p.serial + 1u) * 64u;
    const int pos = *p.pos_ptr + p.pos_off;
    const int j_lo = p.window >= 0 ? max(0, pos - p.window) : 0, n_old = pos - j_lo, last_old = max(n_old - 1, 0);
    // RoPE factors of this wave's two q|k|v passes (the same rows in every layer): pass cw is a q pair; pass 6 + cw is q (cw < 2), k (cw 2, 3) or v
    const int half = EHD / 2;
    const int pr0 = 8 * s + cw, pr1 = cw < 2 ? 8 * s + 6 + cw : 2 * j + (cw - 2);
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };      // wave-uniform values live in SGPRs (the VGPR file is full)
    const float rc0 = uni(p.rope_cos[(size_t)pos * half + pr0]), rs0 = uni(p.rope_sin[(size_t)pos * half + pr0]);
    const float rc1 = cw < 4 ? uni(p.rope_cos[(size_t)pos * half + pr1]) : 1.0f, rs1 = cw < 4 ? uni(p.rope_sin[(size_t)pos * half + pr1]) : 0.0f;
    const float scale = 1.0f / sqrtf((float)EHD);
    const int n_items = 3 * p.n_layers, npass_lm = lm_passes(p.vocab), row0_lm = lm_rows_per_cu(p.vocab) * b;
    if (cw == NCONS - 1) {      // norm weights (* Ada scale) * 512 of this CU's 12 rows for every layer -> LDS, once per launch, while everybody waits for the first all-gather
        const EngLayerTab* tab = reinterpret_cast<const EngLayerTab*>(lds + L_TAB);
        float* gwt = reinterpret_cast<float*>(lds + L_GW);
        for (int i = lane0; i <= p.n_layers * OWN + OWN - 1; i += 64) {
            const int l = i / OWN, r = i - l * OWN; const unsigned k = (unsigned)(OWN * b + r);
            if (l < p.n_layers) {
                gwt[(l * 2) * 16 + r] = as_g(tab[l].attn_norm)[k] * 512.0f;
                gwt[(l * 2 + 1) * 16 + r] = as_g(tab[l].ffn_norm)[k] * as_g(tab[l].ada_mul)[k] * 512.0f;
            } else gwt[(l * 2) * 16 + r] = as_g(p.final_norm)[k] * 512.0f;
        }
        // XCD-local edges (q|k|v -> attention, SwiGLU -> w2) go through the shared L2 only if the 32 workgroups of group g really sit on ONE XCD: workgroup b is
        // observed on XCD (b + rotation) % 8 (the dispatcher's round-robin carries over from the previous launch), so the ids are EXCHANGED and compared
        unsigned ok = 0;
        if (p.flags & 128) {
            const unsigned my = c->xcc_id;
            if (lane0 == 0) publish(p.XC + b, tag_base, __uint_as_float(my));
            float v[1];
            const bool got = sweep<1>(p.XC, NCU * 8u, tag_base, [&](int) { return 8 * (lane0 & 31) + g; }, [&]() { return 0; }, false, v, c, p.err);
            ok = got && __all(__float_as_uint(v[0]) == my) ? 1u : 0u;
            if (!ok && lane0 == 0) __hip_atomic_store(p.err + 1, 9u | ((unsigned)b << 8) | (my << 16), RLX, AG);      // informational (err[1]): the fast edges are off for this launch
        }
        lds_st(&c->xcd_ok, ok);
        ENG_CFENCE(); lds_st(&c->gw_flag, 1u);
    }
    wait_ge(&c->gw_flag, 1u, c, p.err, ERR_STAGE);
    float best = -INFINITY; int best_i = 0x7fffffff;
    const bool xloc = (p.flags & 128) != 0 && lds_ld(&c->xcd_ok) != 0;      // this workgroup runs on XCD blockIdx % 8, like (by the same check) the group's other 31

    // one pass of a 3-plane operator on its two rows: (a, b) = the pass's two row sums (before the RMSNorm scale)
    auto two_rows = [&](float acc, float& a, float& bq) { acc = row16_sum_e(acc); a = rlf(acc, 0) + rlf(acc, 16); bq = rlf(acc, 32) + rlf(acc, 48); };
#pragma unroll 1
    for (int it = 0; it <= n_items; it++) {
        int lane = lane0; asm volatile("" : "+v"(lane));      // opaque per item: lane-derived addresses are recomputed, not carried around the loop in VGPRs
        const int l = it / 3, op = it == n_items ? (int)EOP_LM : (it - 3 * l == 0 ? (int)EOP_QKV : it - 3 * l == 1 ? (int)EOP_W13 : (int)EOP_W2);
        const bool T = l == p.tl_layer && cw == 0 && op != EOP_LM;
        const unsigned tag = tag_base + (unsigned)l + 1u;
        const EngLayerTab* L = reinterpret_cast<const EngLayerTab*>(lds + L_TAB) + (op == EOP_LM ? 0 : l);
        if (op == EOP_QKV) {
            // ================= q|k|v  ->  attention of head h  ->  wo =================
            const gf_p kc = as_g(L->kc) + (size_t)g * p.max_seq * EHD, vc = as_g(L->vc) + (size_t)g * p.max_seq * EHD;
            const int t6 = cw * 64 + lane;
            const int part = t6 & 7, ks = t6 >> 3;              // scores: 8 lanes per key, 48 keys per pass
            const int kg = t6 >> 5, col = t6 & 31;              // P.V: 12 key groups x 32 float4 columns
            wait_ge(&c->xs0_flag, (unsigned)l + 1u, c, p.err, ERR_STAGE);
            if (T) tl(0);
            // the old K rows do not depend on this step: requested BEFORE the q|k|v passes, so they are home before the q|k|v edge is polled (a prefetch burst
            // right behind the publish sat in front of this CU's own granule sweep: +3 us on the edge); the V rows are requested once the edge has resolved
            constexpr int NKP = 3;      // 144 keys up front (48 VGPRs); later keys take the loop below
            float4 kpre[NKP][4];
#pragma unroll
            for (int u = 0; u < NKP; u++) {
                const unsigned ko = (unsigned)(j_lo + min(ks + 48 * u, last_old)) * EHD + part * 16;      // 32-bit lane offset + uniform base: one VGPR per address
#pragma unroll
                for (int e = 0; e < 4; e++) kpre[u][e] = ldg4(kc + (ko + 4 * e));
            }
            {
                XRegs<3> xr; xr.load(xs0, lane & 31, 32);
                const float rstd = c->rstd0;
#pragma unroll
                for (int t = 0; t < QKV_PK; t++) {
                    uint4 Qa[3]; float Sa[3];      // (no register double-buffering here: the 64 K-prefetch registers are live)
                    cs.fetch<3>(cs.P + t, Qa, Sa, true);
                    float a, bq; two_rows(cs.pass_dot<3>(Qa, Sa, xr), a, bq); a *= rstd; bq *= rstd;
                    const int q = cw + 6 * t;
                    int n; float c_, s_;
                    if (t == 0) { n = 128 * h + 16 * s + 2 * q; c_ = rc0; s_ = rs0; }
                    else if (cw < 2) { n = 128 * h + 16 * s + 2 * q; c_ = rc1; s_ = rs1; }
                    else if (cw < 4) { n = EQD + 128 * g + 4 * j + 2 * (q - 8); c_ = rc1; s_ = rs1; }
                    else { n = EQD + EKD + 128 * g + 4 * j + 2 * (q - 10); c_ = 1.0f; s_ = 0.0f; }
                    const float ra = a * c_ - bq * s_, rb = a * s_ + bq * c_;        // interleaved-pair RoPE (rope.rs:99-141); identity for v
                    if (lane < 2) {
                        const float v = lane ? rb : ra;
                        publish_b(p.G, (EQD + 2 * EKD) * 8u, (unsigned)(n + lane), tag, v, xloc);
                        if (t == 1 && cw >= 2) (cw < 4 ? kc : vc)[(size_t)pos * EHD + (n & 127) + lane] = v;      // k / v rows also go to the cache (read by later steps)
                    }
                }
                cs.P += QKV_PK;
            }
            if (T) tl(1);
            float4 vpre[16];
#pragma unroll
            for (int u = 0; u < 16; u++) vpre[u] = ldg4(vc + ((unsigned)(j_lo + min(kg + 12 * u, last_old)) * EHD + col * 4));
            wait_ge(&c->qkv_flag, (unsigned)l + 1u, c, p.err, ERR_STAGE);
            if (T) tl(2);
            {
                float qv[16];
#pragma unroll
                for (int e = 0; e < 4; e++) { const float4 v = *reinterpret_cast<const float4*>(qkvn + part * 16 + 4 * e); qv[4 * e] = v.x; qv[4 * e + 1] = v.y; qv[4 * e + 2] = v.z; qv[4 * e + 3] = v.w; }
                auto dot16 = [&](const float4 (&kk)[4]) {
                    float sacc = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; e++) { sacc = fmaf(qv[4 * e], kk[e].x, sacc); sacc = fmaf(qv[4 * e + 1], kk[e].y, sacc); sacc = fmaf(qv[4 * e + 2], kk[e].z, sacc); sacc = fmaf(qv[4 * e + 3], kk[e].w, sacc); }
                    return group8_sum_e(sacc);
                };
#pragma unroll
                for (int u = 0; u < NKP; u++) {
                    const float sv = dot16(kpre[u]);
                    const int i = ks + 48 * u;
                    if (part == 0 && i < n_old) sc[i] = sv * scale;
                }
                for (int i0 = 48 * NKP; i0 < n_old; i0 += 48) {      // later keys
                    const int i = i0 + ks;
                    float4 kk[4];
                    const unsigned ko = (unsigned)(j_lo + min(i, last_old)) * EHD + part * 16;
#pragma unroll
                    for (int e = 0; e < 4; e++) kk[e] = ldg4(kc + (ko + 4 * e));
                    const float sv = dot16(kk);
                    if (part == 0 && i < n_old) sc[i] = sv * scale;
                }
                if (cw == 0) {                                       // the new key (this step's k row)
                    float4 kk[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) kk[e] = *reinterpret_cast<const float4*>(qkvn + 128 + part * 16 + 4 * e);
                    const float sv = dot16(kk);
                    if (t6 == 0) sc[n_old] = sv * scale;
                }
            }
            cs.cbarrier();
            {
                const int n = n_old + 1;
                float mx = -INFINITY;
                for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sc[i]);
                mx = wave_max_e(mx);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f); float lsum = 0.f;
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const int i = kg + 12 * u;
                    if (i < n_old) {
                        const float pr = expf(sc[i] - mx); const float4 vv = vpre[u];
                        o.x = fmaf(pr, vv.x, o.x); o.y = fmaf(pr, vv.y, o.y); o.z = fmaf(pr, vv.z, o.z); o.w = fmaf(pr, vv.w, o.w); lsum += pr;
                    }
                }
                for (int i = 192 + kg; i < n_old; i += 12) {
                    const float pr = expf(sc[i] - mx); const float4 vv = ldg4(vc + ((unsigned)(j_lo + i) * EHD + col * 4));
                    o.x = fmaf(pr, vv.x, o.x); o.y = fmaf(pr, vv.y, o.y); o.z = fmaf(pr, vv.z, o.z); o.w = fmaf(pr, vv.w, o.w); lsum += pr;
                }
                if (kg == 0) {
                    const float pr = expf(sc[n_old] - mx); const float4 vv = *reinterpret_cast<const float4*>(qkvn + 256 + col * 4);
                    o.x = fmaf(pr, vv.x, o.x); o.y = fmaf(pr, vv.y, o.y); o.z = fmaf(pr, vv.z, o.z); o.w = fmaf(pr, vv.w, o.w); lsum += pr;
                }
                po[kg * 32 + col] = o;
                if (col == 0) pl[kg] = lsum;
            }
            cs.cbarrier();
            if (t6 < EHD) {
                const float* pof = reinterpret_cast<const float*>(po);
                float so = 0.f, sl = 0.f;
#pragma unroll
                for (int q = 0; q < 12; q++) { so += pof[q * 128 + t6]; sl += pl[q]; }      // fixed order
                xo[sw_dword(t6)] = so * (1.0f / sl) * 512.0f;
            }
            cs.cbarrier();
            if (T) tl(3);
            {   // ---------------- wo: rows [384 s, +384) x head h's 128 columns ----------------
                XRegs<2> xr; xr.load(xo, lane & 1, 2);
                uint4 Qa[2], Qb[2]; float Sa[2], Sb[2];
                cs.fetch<2>(cs.P, Qa, Sa, true);
#pragma unroll 1
                for (int t = 0; t < WO_PK; t++) {
                    if (t + 1 < WO_PK) cs.fetch<2>(cs.P + t + 1, Qb, Sb, true);
                    float acc = cs.pass_dot<2>(Qa, Sa, xr);
                    acc += dppf<0xB1>(acc);
                    if ((lane & 1) == 0) publish_b(p.PW, NPW * ED * 8u, (unsigned)(h * ED + 384 * s + 32 * (cw + 6 * t) + (lane >> 1)), tag, acc, false);
#pragma unroll
                    for (int i = 0; i < 2; i++) { Qa[i] = Qb[i]; Sa[i] = Sb[i]; }
                }
                cs.P += WO_PK;
            }
            if (T) tl(4);
            continue;
        }
        // ================= the other 3-plane operators: w1|w3, w2, lm_head (one copy of the pass loop) =================
        unsigned* flag; unsigned target; const float* xs; int first, step, n_pass; float rstd;
        if (op == EOP_W13) { flag = &c->xs1_flag; target = (unsigned)l + 1u; xs = xs1; first = lane & 31; step = 32; n_pass = W13_PK; }
        else if (op == EOP_W2) { flag = &c->xa_flag; target = (unsigned)l + 1u; xs = xa; first = 12 * (cw % 3) + (lane & 3); step = 4; n_pass = W2_PK; }
        else { flag = &c->xs0_flag; target = (unsigned)p.n_layers + 1u; xs = xs0; first = lane & 31; step = 32; n_pass = lm_packets(p.vocab); }
        wait_ge(flag, target, c, p.err, ERR_STAGE);
        if (T) tl(op == EOP_W13 ? 5 : 7);
        const bool TP = T && op == EOP_W13;
        rstd = op == EOP_W13 ? c->rstd1 : op == EOP_W2 ? 1.0f : c->rstd0;
        XRegs<3> xr; xr.load(xs, first, step);
        uint4 Qa[3], Qb[3]; float Sa[3], Sb[3];
        if (TP) tl(24);
        cs.fetch<3>(cs.P, Qa, Sa, op != EOP_LM || cw < npass_lm);
        if (TP) tl(25);
#pragma unroll 1
        for (int t = 0; t < n_pass; t++) {
            const int q = cw + 6 * t;                                  // this wave's pass of packet t
            if (t + 1 < n_pass) cs.fetch<3>(cs.P + t + 1, Qb, Sb, op != EOP_LM || q + 6 < npass_lm);
            if (op != EOP_LM || q < npass_lm) {
                float acc = cs.pass_dot<3>(Qa, Sa, xr);
                if (op == EOP_W2) {
                    acc += dppf<0xB1>(acc); acc += dppf<0x4E>(acc);
                    if ((lane & 3) == 0) publish_b(p.P2, NP2 * ED * 8u, (unsigned)((3 * g + cw % 3) * ED + 96 * j + 48 * (cw / 3) + 16 * t + (lane >> 2)), tag, acc, false);
                } else {
                    float a, bq; two_rows(acc, a, bq); a *= rstd; bq *= rstd;
                    if (op == EOP_W13) {
                        if (lane == 0) publish_b(p.A, EF * 8u, (unsigned)(1152 * g + 36 * j + q), tag, silu_e(a) * bq, xloc);
                    } else {
                        const int n = row0_lm + 2 * q;
                        if (p.logits_out && lane < 2) p.logits_out[n + lane] = lane ? bq : a;
                        if (a > best || (a == best && n < best_i)) { best = a; best_i = n; }
                        if (bq > best || (bq == best && n + 1 < best_i)) { best = bq; best_i = n + 1; }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 3; i++) { Qa[i] = Qb[i]; Sa[i] = Sb[i]; }
            if (TP) tl(26 + t);
        }
        cs.P += (unsigned)n_pass;
        if (T) tl(op == EOP_W13 ? 6 : 14);
    }
    // ---------------- argmax partial of this CU ----------------
    if (lane0 == 0) { c->best_val[cw] = best; c->best_idx[cw] = best_i; }
    cs.cbarrier();
    if (cw == 0 && lane0 == 0) {
        float bv = c->best_val[0]; int bi = c->best_idx[0];
        for (int w = 1; w < NCONS; w++) { const float v = c->best_val[w]; const int ii = c->best_idx[w]; if (v > bv || (v == bv && ii < bi)) { bv = v; bi = ii; } }
        p.part_val[b] = bv; p.part_idx[b] = bi;
    }
    if (cw == 0) tl(15);
}

__global__ __launch_bounds__(NTHR, 1) void decode_engine_kernel(const EngParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    EngCtl* c = reinterpret_cast<EngCtl*>(lds + L_CTL);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < (int)(sizeof(EngCtl) / 4)) reinterpret_cast<unsigned*>(c)[tid] = 0u;
    for (int i = tid; i < p.n_layers * (int)(sizeof(EngLayerTab) / 8); i += NTHR) reinterpret_cast<u64*>(lds + L_TAB)[i] = reinterpret_cast<const u64*>(p.layers)[i];
    if (tid == 0) c->xcc_id = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID (id 20), bits 3:0
    __syncthreads();
    Tl tl; tl.on = p.tl != nullptr && lane == 0; tl.buf = p.tl ? p.tl + (size_t)blockIdx.x * 32 : nullptr;
#ifndef ENG_ROLES
#define ENG_ROLES 7
#endif
    if (wave == 0) {
        tl(19);
        if (ENG_ROLES & 1) eng_loader(p, c, (unsigned)(uintptr_t)(lds + L_RING), lane, tl);
    } else if (wave == 1) {
        if (ENG_ROLES & 2) eng_comm(p, c, lds, lane, tl);
        // the last workgroup-independent act of the launch: bump the serial (every workgroup has read it long before any lm_head input existed)
        if (blockIdx.x == 0 && lane == 0) { const unsigned sv = *p.serial; asm volatile("" ::: "memory"); *p.serial = sv + 1u; }
    } else {
        if (ENG_ROLES & 4) eng_consumer(p, c, lds, wave - 2, lane, tl);
    }
}

}  // namespace

bool eng_geometry_ok(int D, int n_heads, int n_kv, int hd, int ffn, int vocab, int max_seq) {
    return D == ED && n_heads == ENH && n_kv == ENKV && hd == EHD && ffn == EF && vocab > 0 && vocab % (2 * NCU) == 0 && max_seq > 0 && max_seq <= SC_MAX;
}
size_t eng_stream_bytes(int n_layers, int vocab) { return cu_stream_bytes(n_layers, vocab) * NCU; }
int eng_lds_bytes() { return L_TOTAL; }

// state block layout (bytes): H0, H1 [3072] | G [6144] | PW [32][3072] | A [9216] | P2 [24][3072] granules, then serial, err
static constexpr size_t ST_H0 = 0, ST_H1 = ST_H0 + (size_t)ED * 8, ST_G = ST_H1 + (size_t)ED * 8, ST_PW = ST_G + (size_t)(EQD + 2 * EKD) * 8,
                        ST_A = ST_PW + (size_t)NPW * ED * 8, ST_P2 = ST_A + (size_t)EF * 8, ST_SS0 = ST_P2 + (size_t)NP2 * ED * 8, ST_SS1 = ST_SS0 + (size_t)NCU * 8, ST_XC = ST_SS1 + (size_t)NCU * 8, ST_SERIAL = ST_XC + (size_t)NCU * 8, ST_ERR = ST_SERIAL + 256, ST_TOTAL = ST_ERR + 256;
size_t eng_state_bytes() { return ST_TOTAL; }
void eng_state_carve(unsigned char* st, EngParams* p) {
    p->H0 = reinterpret_cast<unsigned long long*>(st + ST_H0); p->H1 = reinterpret_cast<unsigned long long*>(st + ST_H1);
    p->XC = reinterpret_cast<unsigned long long*>(st + ST_XC);
    p->SS0 = reinterpret_cast<unsigned long long*>(st + ST_SS0); p->SS1 = reinterpret_cast<unsigned long long*>(st + ST_SS1);
    p->G = reinterpret_cast<unsigned long long*>(st + ST_G); p->PW = reinterpret_cast<unsigned long long*>(st + ST_PW);
    p->A = reinterpret_cast<unsigned long long*>(st + ST_A); p->P2 = reinterpret_cast<unsigned long long*>(st + ST_P2);
    p->serial = reinterpret_cast<unsigned*>(st + ST_SERIAL); p->err = reinterpret_cast<unsigned*>(st + ST_ERR);
}

hipError_t launch_eng_pack(const Q4W& w, int op, int layer, int n_layers, unsigned char* stream, int vocab, hipStream_t s) {
    if (w.fmt != WFMT_Q4_0 || !w.qs || !w.sc) return hipErrorInvalidValue;
    int passes, off_bytes, N, K;
    switch (op) {
    case EOP_QKV: passes = QKV_PK * NCONS; off_bytes = OFF_QKV; N = EQD + 2 * EKD; K = ED; break;
    case EOP_WO: passes = WO_PK * NCONS; off_bytes = OFF_WO; N = ED; K = EQD; break;
    case EOP_W13: passes = W13_PK * NCONS; off_bytes = OFF_W13; N = 2 * EF; K = ED; break;
    case EOP_W2: passes = W2_PK * NCONS; off_bytes = OFF_W2; N = ED; K = EF; break;
    case EOP_LM: passes = lm_passes(vocab); off_bytes = 0; N = vocab; K = ED; break;
    default: return hipErrorInvalidValue;
    }
    if (w.N != N || w.K != K) return hipErrorInvalidValue;
    const size_t op_off = op == EOP_LM ? (size_t)n_layers * LAYER_BYTES : (size_t)layer * LAYER_BYTES + off_bytes;
    eng_pack_kernel<<<dim3(passes, NCU), dim3(192), 0, s>>>(w, op, stream, cu_stream_bytes(n_layers, vocab), op_off, vocab);
    return hipGetLastError();
}

hipError_t launch_decode_engine(const EngParams& p, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    decode_engine_kernel<<<dim3(NCU), dim3(NTHR), L_TOTAL, s>>>(p);
    return hipGetLastError();
}

}  // namespace vox
