// vox_engine_b16.hip -- persistent decode-LAYER engine for a group of <= 16 sequences on gfx950 (MI355X): the 26 decoder layers of one batched decode step as ONE launch.
//
// What it replaces: the 26 x 5 dependent launches of the batched decode step (vox_api.cpp `group_chain`; reference: gguf/model.rs:938-960 run for B > 1 sequences -- the
// reference itself is batch-1, BASELINE configs[3] asks for 16 utterances).  Those launches run at 0.14 of the HBM peak: five kernel boundaries per layer, and every workgroup
// of every GEMM re-reads the whole 16-row activation block from L2.  Here the weights come through the SAME per-CU packet stream and LDS ring as the
// single-stream engine (vox_engine.hip, vox_engine_common.h: one loader wave per CU, LDS-DMA, consumption order), so no workgroup ever re-reads a weight or an activation
// from L2, and the CU <-> row decomposition is the same:
//   CU b = (g = b % 8: KV head / XCD, j = b / 8): query head h = 4 g + j / 8, slice s = j % 8
//     q|k|v   16 + 4 + 4 weight rows x K 3072 (K split over the 12 consumer waves)            -> RoPE at each sequence's own position -> granules G, KV cache
//     attn    head h of sequences 2 s, 2 s + 1 (six waves each: per-wave online softmax over its keys, combined through LDS)   -> XO (A fragments of wo, read inside the XCD group)
//     wo      rows [96 j, +96) x the 512 columns of the group's four heads (its OWN stream, EOP_WOB: the single-stream engine splits wo by head) -> partial plane g of PW;
//             the owner of columns [12 b, +12) sums the 8 planes + residual                                                                       -> XH1
//     w1|w3   72 interleaved rows (36 SwiGLU outputs) x K 3072                                  -> XA (A fragments of w2, read inside the XCD group)
//     w2      rows [96 j, +96) x the group's 1152 columns -> partial plane g of P2; the owner sums the 8 planes + residual                           -> XH0
// Arithmetic = the batched skinny kernels' (q4_skinny_kernel): v_mfma_f32_16x16x32_bf16, A = 16 sequences x 32 columns as bf16 hi + lo (x ~ hi + lo to 2^-17), B = the Q4
// nibbles as EXACT bf16 integers 128 + q (bits 0x4300 | q, two v_perm_b32 per four weights), the -136 sum(x) correction as the MFMA's C operand (one MFMA pair against a
// constant bf16(-136) per block, kept in LDS), the f16 block scale applied to the f32 result.  One step record (16 rows x 64 columns, written for the int8 MFMA of the
// single-stream engine) feeds two bf16 MFMA pairs: v_permlane32_swap moves the high nibbles of lane groups 0, 1 to groups 2, 3 and the low nibbles of groups 2, 3 to groups
// 0, 1, after which each register pair holds ONE Q4 block in natural column order (lane group y = columns 8 y .. 8 y + 7).
// The activations never touch LDS: a wave's K slice (256 columns x 16 sequences, hi + lo = 16 KB) is loaded straight into 64 VGPRs as MFMA A fragments, because the
// producers publish them in fragment order: XH / XA / XO = [32-column block][hi, lo][lane (y, m)] x 16 bytes.
// Edges carry kilobytes, not 8-byte granules: payload with write-through (sc1) stores, s_waitcnt vmcnt(0), then ONE flag word per producing CU (MI355X_MICROARCH.md
// handoff-flag / publish-large); the COMM wave polls the flags, the consumer waves then fetch their slices.  The all-gather flags sit 16 bytes apart: 256 CUs write
// theirs within a microsecond, and write-through stores into ONE 128-byte line complete one after the other at the memory side (profiles/r04_b16_flag_spacing.txt).
// The partial planes of wo / w2 need no flags at all: a slot is "empty" (all ones) until its producer stores it, the owner polls the slots and puts the marker back.
// The small q|k|v edge keeps the tagged 8-byte granules.  Edges inside one XCD group (q|k|v, attention output, SwiGLU output) use plain stores when the start-up check finds the group on one XCD.
// Tags / flags = (launch serial + 1) * 64 + layer + 1 as in the single-stream engine; every spin is bounded (20 ms) and fails the launch loudly (*err).
// TWO GROUPS PER LAUNCH (NG = 2, round 5; the wide batch's slot groups, vox_api.cpp transcribe_continuous_impl): every phase of a layer runs for group A, then for group B,
// on the same waves -- B computes while A's edge resolves and the other way round (the launch is bound by its hand-offs, not by the matrix pipe or HBM: section 3.3c of
// DESIGN.md).  The loader streams every operator's packets twice (the ring cannot hold an operator; the second pass comes out of L2 / MALL), each group has its own edge
// buffers, flags and launch serial (two state blocks), its own COMM-owned LDS (residual columns, RMSNorm scales, q / k / v of the step, RoPE factors, positions), and the
// COMM wave serves the two chains in the same A, B order.  Every counter that is a function of the layer for one group (ring packet index, barrier and publish counts)
// becomes a function of (layer, phase, group) in that one global order.  The time-shared LDS regions keep their single copy: a wave may run one phase ahead of the finishing
// waves of the previous phase (there is no edge between A's phase and B's), so before its first store into a time-shared region a wave waits until all twelve have
// published the previous phase (`guard`: one LDS poll, satisfied long before in the common case).  Cache slices are picked per sequence (EngBParams::kv_row, KVR = true).
// The final norm's input leaves the launch in the launch-based path's own format (XF planes of h * final_norm + 256 partial sums of squares), so the step's tail --
// the 16-row lm_head GEMM and the argmax / next-embedding kernel -- is unchanged.
#include "vox_kernels.h"

#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

namespace vox {
namespace {

#include "vox_engine_common.h"

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int BM = 16;                                   // sequences per launch (one MFMA m-tile)
constexpr int BNSLOT = 4;                                // ring slots (the partial-sum planes of w1|w3 need 54 KB of LDS at 16 rows)
constexpr int NBLK = ED / 32;                            // 96 column blocks of the residual stream
constexpr int FRAG = 1024;                               // one (block, plane) fragment line: 64 lanes x 16 bytes
constexpr int XH_BYTES = NBLK * 2 * FRAG;                // 196608
constexpr int XO_HEAD = 4 * 2 * FRAG;                    // 8192: one head's attention output (4 blocks)
constexpr int XA_GROUP = 36 * 2 * FRAG;                  // 73728: one XCD group's SwiGLU outputs (36 blocks)
constexpr int G_ROW = EQD + 2 * EKD;                     // 6144 granules per sequence
constexpr int PH = 5;                                    // publish phases per layer counted in pub_cnt: q|k|v, attention, wo, w1|w3, w2

struct BCtl {
    unsigned ring_ready[8], ring_done[8];
    unsigned cbar, dead, gathering, gw_flag;
    unsigned xcd_ok, xcc_id, pub_cnt, pad0;
    unsigned ag_flag[2], rs_flag[2], qkv_flag[2], wo_flag[2], xa_flag[2];      // per group; rs_flag: all-gather stages whose RMSNorm scales are in LDS
    unsigned tbar[2];
};
// ---- LDS map ----
constexpr int BL_RING = 0;
constexpr int BL_PART = BL_RING + BNSLOT * SLOT_BYTES;   // cross-wave partial sums of the operator in flight (time-shared; see the ordering argument at each use)
constexpr int PART_WAVE = 4 * 1024 + 512;                //   q|k|v / w1|w3: [wave][tile][lane] float4, the 8-row tile as [y][n < 8]
constexpr int PART_BYTES = NCONS * PART_WAVE;            //   55296; w2: [tile 6][K slice 6][lane] float4 = 36864
constexpr int BL_ATT = BL_PART + 24576;                  //   attention scratch (behind q|k|v's partials, which the two finishing waves may still be reading)
constexpr int BL_PO = BL_ATT;                            //     [12][128] f32 partial outputs
constexpr int BL_ML = BL_PO + NCONS * 128 * 4;           //     [12][2] {running max, running sum}
constexpr int BL_OF = BL_ML + 128;                       //     [2][128] f32 final outputs
constexpr int BL_QKVN = BL_PART + PART_BYTES;            // [2 sequences][q 128 | k 128 | v 128] f32 of this step
constexpr int BL_CB = BL_QKVN + 2 * 384 * 4;             // [12 waves][8 blocks][16 m] f32: -136 * sum over the block of x[m][.]; after w1|w3's barrier: SwiGLU outputs [16 m][36] f32
constexpr int BL_OWN = BL_CB + NCONS * 8 * 64;           // [12 n][16 m] f32 residual stream of the CU's columns, [12][16] post-attention stream
constexpr int BL_RSTD = BL_OWN + 2 * OWN * BM * 4;       // [2][16] RMSNorm scales of the two all-gathers
constexpr int BL_ROPE = BL_RSTD + 2 * BM * 4;            // q: cos [8 pairs][16 m], sin [8][16]; k: cos [2][16], sin [2][16]
constexpr int BL_POS = BL_ROPE + (2 * 8 * BM + 2 * 2 * BM) * 4;      // [16] int positions
constexpr int BL_TAB = BL_POS + BM * 4;
constexpr int MAX_LAYERS = 32;
constexpr int BL_GW = BL_TAB + MAX_LAYERS * (int)sizeof(EngLayerTab);      // [MAX_LAYERS + 1][2][16] norm weights of the CU's 12 columns
constexpr int BL_CTL = BL_GW + (MAX_LAYERS + 1) * 32 * 4;
constexpr int BL_KVR = BL_CTL + (int)sizeof(BCtl);      // [16] int cache slice of every sequence (KVR form)
constexpr int BL_TOTAL = BL_KVR + BM * 4;
// the second group's COMM-owned regions (NG = 2): q|k|v of the step, residual columns, RMSNorm scales, RoPE factors, positions, cache slices
constexpr int G1_QKVN = 0, G1_OWN = G1_QKVN + 2 * 384 * 4, G1_RSTD = G1_OWN + 2 * OWN * BM * 4, G1_ROPE = G1_RSTD + 2 * BM * 4, G1_POS = G1_ROPE + (2 * 8 * BM + 2 * 2 * BM) * 4,
              G1_KVR = G1_POS + BM * 4, G1_BYTES = G1_KVR + BM * 4;
constexpr int BL_G1 = BL_TOTAL, BL_TOTAL2 = BL_G1 + G1_BYTES;
static_assert(BL_TOTAL2 <= 160 * 1024, "LDS budget");
static_assert(BL_KVR % 16 == 0 && BL_G1 % 16 == 0 && G1_OWN % 16 == 0 && G1_RSTD % 16 == 0 && G1_ROPE % 16 == 0 && G1_POS % 16 == 0, "aligned carve (second group)");
// group q's copy of a COMM-owned region (q is wave-uniform; NG = 1: q is the constant 0)
__device__ __forceinline__ int lg(int base0, int rel1, int q) { return q == 0 ? base0 : BL_G1 + rel1; }
#define L_QKVN(q) lg(BL_QKVN, G1_QKVN, q)
#define L_OWN(q) lg(BL_OWN, G1_OWN, q)
#define L_RSTD(q) lg(BL_RSTD, G1_RSTD, q)
#define L_ROPE(q) lg(BL_ROPE, G1_ROPE, q)
#define L_POS(q) lg(BL_POS, G1_POS, q)
#define L_KVR(q) lg(BL_KVR, G1_KVR, q)
static_assert(BL_OF + 2 * 128 * 4 <= BL_PART + PART_BYTES && 6 * 6 * 1024 <= PART_BYTES && BM * 36 * 4 <= NCONS * 8 * 64, "time-shared regions");
static_assert(BL_QKVN % 16 == 0 && BL_CB % 16 == 0 && BL_OWN % 16 == 0 && BL_RSTD % 16 == 0 && BL_ROPE % 16 == 0 && BL_TAB % 16 == 0 && BL_CTL % 16 == 0 && BL_ATT % 16 == 0, "aligned carve");

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) { const f32x2_t v = {a, b}; union { bf16x2_t b; unsigned u; } c; c.b = __builtin_convertvector(v, bf16x2_t); return c.u; }
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {      // x ~ hi + lo (two bf16), as vox_kernels.hip
    hi = cvt_pk_bf16(a, b);
    lo = cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 ld_frag(srd_t sd, unsigned off) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)off, 0, 16)); }      // sc1: never this CU's L1
__device__ __forceinline__ void st_u32(srd_t sd, unsigned off, unsigned v, bool local) { if (local) __builtin_amdgcn_raw_buffer_store_b32(v, sd, (int)off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b32(v, sd, (int)off, 0, 16); }
__device__ __forceinline__ void st_u64(srd_t sd, unsigned off, unsigned a, unsigned b, bool local) {
    v2u_t x; x.x = a; x.y = b;
    if (local) __builtin_amdgcn_raw_buffer_store_b64(x, sd, (int)off, 0, 0); else __builtin_amdgcn_raw_buffer_store_b64(x, sd, (int)off, 0, 16);
}
__device__ __forceinline__ void st_f4(srd_t sd, unsigned off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sd, (int)off, 0, 16); }
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// fragment address of (block, plane, lane group y, sequence m) inside an XH / XA / XO buffer
__device__ __forceinline__ unsigned frag_off(int blk, int plane, int y, int m) { return (unsigned)(((blk * 2 + plane) * 64 + y * 16 + m) * 16); }

// one step record (16 or 8 weight rows x 64 columns = Q4 blocks 2 T, 2 T + 1) as a lane holds it: 8 bytes of nibbles + the two f16 block scales of its row.  Records do not
// depend on the step's activations: a wave copies them from the ring into REGISTERS while it waits for an edge and frees the ring slots for the loader.
struct RawRec { uint2 q; unsigned sc; };      // sc: the two f16 scale bit patterns (block 2 T | block 2 T + 1 << 16)
__device__ __forceinline__ RawRec rec_load(const unsigned char* rec, bool half, int lane) {
    const int n = lane & 15, g = lane >> 4;
    RawRec r; r.q = make_uint2(0u, 0u); r.sc = 0u;
    if (!half) {
        r.q = reinterpret_cast<const uint2*>(rec)[lane];
        const unsigned short* scp = reinterpret_cast<const unsigned short*>(rec + REC_SC);
        r.sc = (unsigned)scp[n] | ((unsigned)scp[16 + n] << 16);
    } else if (n < 8) {
        r.q = reinterpret_cast<const uint2*>(rec)[g * 8 + n];
        const unsigned short* scp = reinterpret_cast<const unsigned short*>(rec + REC_H_SC);
        r.sc = (unsigned)scp[n] | ((unsigned)scp[8 + n] << 16);
    }
    return r;
}
// the record against the step's two activation blocks: acc[m][n] += sum_k x[m][k] * w[n][k]
__device__ __forceinline__ f32x4 rec_mma(const RawRec& r, bf16x8 ah0, bf16x8 al0, bf16x8 ah1, bf16x8 al1, f32x4 c0, f32x4 c1, f32x4 acc) {
    const uint2 q = r.q;
    const unsigned lox = q.x & 0x0F0F0F0Fu, loy = q.y & 0x0F0F0F0Fu, hix = (q.x >> 4) & 0x0F0F0F0Fu, hiy = (q.y >> 4) & 0x0F0F0F0Fu;
    // lanes 32..63 of `lo` <-> lanes 0..31 of `hi`: first result = block 2 T in every lane group (y < 2: own low nibbles = columns 8 y ..; y >= 2: the high nibbles of group
    // y - 2 = columns 16 + 8 (y - 2) ..), second result = block 2 T + 1 (y < 2: the low nibbles of group y + 2; y >= 2: own high nibbles)
    const auto sx = __builtin_amdgcn_permlane32_swap(lox, hix, false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(loy, hiy, false, false);
    const unsigned c43 = 0x43434343u;
    u32x4 b0, b1;
    b0.x = __builtin_amdgcn_perm(sx[0], c43, 0x00050004u); b0.y = __builtin_amdgcn_perm(sx[0], c43, 0x00070006u);
    b0.z = __builtin_amdgcn_perm(sy[0], c43, 0x00050004u); b0.w = __builtin_amdgcn_perm(sy[0], c43, 0x00070006u);
    b1.x = __builtin_amdgcn_perm(sx[1], c43, 0x00050004u); b1.y = __builtin_amdgcn_perm(sx[1], c43, 0x00070006u);
    b1.z = __builtin_amdgcn_perm(sy[1], c43, 0x00050004u); b1.w = __builtin_amdgcn_perm(sy[1], c43, 0x00070006u);
    f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, as_bf16x8(b0), c0, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, as_bf16x8(b0), d0, 0, 0, 0);
    f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, as_bf16x8(b1), c1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, as_bf16x8(b1), d1, 0, 0, 0);
    const float s0 = __half2float(__ushort_as_half((unsigned short)(r.sc & 0xFFFFu))), s1 = __half2float(__ushort_as_half((unsigned short)(r.sc >> 16)));
    acc = __builtin_elementwise_fma((f32x4){s0, s0, s0, s0}, d0, acc);
    acc = __builtin_elementwise_fma((f32x4){s1, s1, s1, s1}, d1, acc);
    return acc;
}
// -136 * sum_k x[m][k] of one block, in the accumulator layout (the same value in every column n)
// (mm = bf16(-136) x 8 comes from the caller, materialised per load_a: as a constant in here hipcc hoisted the four registers out of the layer loop, spilled them and
// reloaded them -- scratch_load + s_waitcnt vmcnt(0) -- in front of the first MFMA of every operator)
__device__ __forceinline__ f32x4 corr16(bf16x8 ah, bf16x8 al, u32x4 mm) {
    const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, as_bf16x8(mm), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, as_bf16x8(mm), s, 0, 0, 0);
}

// the kernel's arguments: one parameter block per group, back to back in the kernarg segment
template <int NG> struct EngBArgs { EngBParams g[NG]; };
__device__ __forceinline__ EngBParams kparams(int q) {      // group q's block by value from the constant address space: the fields a phase uses become scalar loads
#if defined(__HIP_DEVICE_COMPILE__)
    return ((const __attribute__((address_space(4))) EngBParams*)__builtin_amdgcn_kernarg_segment_ptr())[q];
#else
    (void)q; return EngBParams{};
#endif
}
__device__ __forceinline__ unsigned char* lds_base() { extern __shared__ __attribute__((aligned(16))) unsigned char lds_[]; return lds_; }
__device__ __forceinline__ unsigned b16_tag_base(const EngBParams& p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)((*p.serial + 1u) * 64u)); }

// ------------------------------------------------------------------------------------------------
// LOADER wave: the layer packets of the single-stream engine's stream, four ring slots
// ------------------------------------------------------------------------------------------------
template <int NG>
__device__ __forceinline__ void b16_loader(const EngBParams& p, BCtl* c, unsigned ring_lds, int lane, const Tl& tl) {
    Loader<BCtl, BNSLOT> ld; ld.c = c; ld.err = p.err; ld.ring_lds = ring_lds; ld.voff = (unsigned)lane * 16u; ld.thin = (p.flags & 1) != 0; ld.pace = 0;
    if (p.flags & 1024) ld.depth = 2;
    if (p.flags & 2048) ld.depth = 1;
    if (p.flags & 64) ld.pause_ticks = 300;      // nothing new in flight while this CU's COMM wave polls / sums / publishes (bounded: 3 us)
    const bool nodma = (p.flags & 32) != 0;
    const u64 base = (u64)p.stream;
    if (NG == 1) {
        const unsigned n_pk = (unsigned)p.n_layers * PK_LAYER;
        unsigned l = 0, r = 0; u64 off = 0;
#pragma unroll 1
        for (unsigned pk = 0; pk < n_pk; pk++) {
            const int bytes = r < PK_LAYER_M ? PK_M : PK_A;
            if (r == 0 && (int)l == p.tl_layer) tl(16);
            const bool wo_pk = r >= (unsigned)QKV_PK && r < (unsigned)(QKV_PK + WO_PK);      // wo: the XCD-group split lives in its own stream, [layer][packet][CU][bytes]
            const u64 src = wo_pk ? (u64)p.stream_wo + (u64)NCU * ((u64)l * WOB_LAYER_BYTES + (u64)(r - QKV_PK) * PK_M) + (u64)blockIdx.x * (u64)PK_M
                                  : base + (u64)NCU * off + (u64)blockIdx.x * (u64)bytes;
            if (bytes == PK_M) ld.issue<PK_M>(src, lane, nodma); else ld.issue<PK_A>(src, lane, nodma);
            off += (u64)bytes;
            if (++r == PK_LAYER) { if ((int)l == p.tl_layer) tl(17); r = 0; l++; }
        }
    } else {
        // NG groups: every operator's packets NG times in a row (group A's pass, then group B's: the consumption order of the interleaved phases)
#pragma unroll 1
        for (unsigned l = 0; l < (unsigned)p.n_layers; l++) {
            if ((int)l == p.tl_layer) tl(16);
#pragma unroll 1
            for (int seg = 0; seg < 4; seg++) {      // q|k|v, wo, w1|w3, w2
                const int r0 = seg == 0 ? 0 : seg == 1 ? QKV_PK : seg == 2 ? QKV_PK + WO_PK : PK_LAYER_M, npk = seg == 0 ? QKV_PK : seg == 1 ? WO_PK : seg == 2 ? W13_PK : W2_PK;
                const u64 seg_src = seg == 1 ? (u64)p.stream_wo + (u64)NCU * ((u64)l * WOB_LAYER_BYTES) + (u64)blockIdx.x * (u64)PK_M
                                  : seg == 3 ? base + (u64)NCU * ((u64)l * LAYER_BYTES + (u64)OFF_W2) + (u64)blockIdx.x * (u64)PK_A
                                             : base + (u64)NCU * ((u64)l * LAYER_BYTES + (u64)r0 * PK_M) + (u64)blockIdx.x * (u64)PK_M;
#pragma unroll 1
                for (int q = 0; q < NG; q++) {
#pragma unroll 1
                    for (int i = 0; i < npk; i++) {
                        if (seg == 3) ld.issue<PK_A>(seg_src + (u64)NCU * (u64)i * PK_A, lane, nodma);
                        else ld.issue<PK_M>(seg_src + (u64)NCU * (u64)i * PK_M, lane, nodma);
                    }
                }
            }
            if ((int)l == p.tl_layer) tl(17);
        }
    }
    ld.flush();
    tl(18);
}

// ------------------------------------------------------------------------------------------------
// COMM wave
// ------------------------------------------------------------------------------------------------
// poll n flag words (n <= 256, a multiple of 4 or < 4 ... one 16-byte load per lane covers four) until every one equals `tag`.  idx(i) = word index of flag i.
template <class IdxF>
__device__ __forceinline__ bool poll_flags(const unsigned* base_, unsigned words, int n, IdxF idx, unsigned tag, int lane, BCtl* c, unsigned* err, unsigned stride = 4u) {
    const srd_t sd = make_srd(base_, words * stride);
    u64 t0 = 0;
    const unsigned off = (unsigned)idx(min(lane, n - 1)) * stride;
    for (;;) {
        const unsigned f = __builtin_amdgcn_raw_buffer_load_b32(sd, (int)off, 0, 16);
        if (__all(f == tag)) return true;
        if (sweep_bail(t0, tag, c, err)) return false;
    }
}
// all 256 flags of an all-gather, FSTRIDE bytes apart (one dword each): 256 CUs write them within a microsecond of each other and every CU polls all of them; packed
// into 1 KB (32 writers per 128-byte line) the last flag became visible 3.0 us after its store, as bytes in 256 B (128 writers per line) 7 us -- same-line write-through
// stores serialise at the memory side (profiles/r04_b16_flag_spacing.txt)
constexpr int FSTRIDE = 16;
__device__ __forceinline__ bool poll_flags256(const unsigned* base_, unsigned tag, int lane, BCtl* c, unsigned* err) {
    const srd_t sd = make_srd(base_, NCU * FSTRIDE);
    u64 t0 = 0;
    for (;;) {
        unsigned f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) f[u] = __builtin_amdgcn_raw_buffer_load_b32(sd, (lane + 64 * u) * FSTRIDE, 0, 16);
        if (__all(f[0] == tag && f[1] == tag && f[2] == tag && f[3] == tag)) return true;
        if (sweep_bail(t0, tag, c, err)) return false;
    }
}
__device__ __forceinline__ void st_flag_ag(const unsigned* base_, unsigned idx, unsigned tag) { __builtin_amdgcn_raw_buffer_store_b32(tag, make_srd(base_, NCU * FSTRIDE), (int)(idx * FSTRIDE), 0, 16); }
// RMSNorm scales of an all-gathered stream: 256 per-CU partial sums of squares per sequence, summed in a FIXED order (u ascending, then the butterfly) -> rstd[16] in LDS
__device__ __forceinline__ void comm_rstd(const float* ss, float eps, int lane, float* rstd_out) {
    const srd_t sd = make_srd(ss, NCU * BM * 4u);
    const int q = lane >> 2, mq = lane & 3;
    u32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = ld_frag(sd, (unsigned)(((16 * q + u) * BM + 4 * mq) * 4));
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 16; u++) { a[0] += __uint_as_float(v[u].x); a[1] += __uint_as_float(v[u].y); a[2] += __uint_as_float(v[u].z); a[3] += __uint_as_float(v[u].w); }
#pragma unroll
    for (int sft = 4; sft < 64; sft <<= 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] += __shfl_xor(a[i], sft, 64);
    }
    if (q == 0) {
#pragma unroll
        for (int i = 0; i < 4; i++) rstd_out[4 * mq + i] = 1.0f / sqrtf(a[i] / (float)ED + eps);
    }
}
// the owner's side of an all-gather: own[n][m] (raw residual stream of columns 12 b + n) -> fragment pieces of own * gw (bf16 hi + lo, four columns = 8 bytes per store),
// the CU's partial sums of squares, drained, flag
__device__ __forceinline__ void comm_publish_rows(const EngBParams& p, int lane, const float* own, const float* gw, unsigned char* xh, float* ss, unsigned* flags, unsigned tag) {
    const int b = blockIdx.x, m = lane & 15, t = lane >> 4;
    const srd_t xd = make_srd(xh, XH_BYTES), sd = make_srd(ss, NCU * BM * 4u);
    float sq = 0.f;
    if (t < 3) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const float raw = own[(4 * t + i) * BM + m]; sq = fmaf(raw, raw, sq); v[i] = raw * gw[4 * t + i]; }
        unsigned h0, l0, h1, l1; split_pair(v[0], v[1], h0, l0); split_pair(v[2], v[3], h1, l1);
        const int col = OWN * b + 4 * t, blk = col >> 5, y = (col & 31) >> 3, hf = (col >> 2) & 1;
        st_u64(xd, frag_off(blk, 0, y, m) + 8u * hf, h0, h1, false);
        st_u64(xd, frag_off(blk, 1, y, m) + 8u * hf, l0, l1, false);
    }
    sq += __shfl(sq, (lane + 16) & 63, 64) + __shfl(sq, (lane + 32) & 63, 64);      // lanes 0..15: t = 0, 1, 2 (t = 3 contributes 0 through lane + 48 -> not added)
    if (lane < 16) st_u32(sd, (unsigned)((b * BM + m) * 4), __float_as_uint(sq), false);
    drain_vm();
    if (lane == 0) st_flag_ag(flags, (unsigned)b, tag);
}
// the layer stack's output: h * final_norm in the launch-based lm_head's XF format (xf_store4 in vox_kernels.hip) + the CU's partial sums of squares (plain stores: the
// kernel boundary publishes them)
__device__ __forceinline__ void comm_publish_final(const EngBParams& p, int lane, const float* own, const float* gw) {
    const int b = blockIdx.x, m = lane & 15, t = lane >> 4;
    float sq = 0.f;
    if (t < 3) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { const float raw = own[(4 * t + i) * BM + m]; sq = fmaf(raw, raw, sq); v[i] = raw * gw[4 * t + i]; }
        const int k = OWN * b + 4 * t, q = k >> 7, j = (k >> 5) & 3, e = k & 31, hf = e >> 4, g = (e & 15) >> 2;
        const size_t base = ((size_t)((q * 4 + j) * 64 + g * 16 + m)) * 8 + 2 * hf, plane = (size_t)(ED >> 7) * 256 * 8;
        unsigned hi, lo;
        split_pair(v[0], v[2], hi, lo); *reinterpret_cast<unsigned*>(p.xf_out + base) = hi; *reinterpret_cast<unsigned*>(p.xf_out + plane + base) = lo;
        split_pair(v[1], v[3], hi, lo); *reinterpret_cast<unsigned*>(p.xf_out + base + 4) = hi; *reinterpret_cast<unsigned*>(p.xf_out + plane + base + 4) = lo;
    }
    sq += __shfl(sq, (lane + 16) & 63, 64) + __shfl(sq, (lane + 32) & 63, 64);
    if (lane < 16) p.ssq_out[b * BM + m] = sq;
}
// sum NP partial planes of the CU's 12 columns x 16 sequences (planes[pl][col][m], plane pl's producer-side index given by plane_idx) + the residual -> dst[n][m]
// The partial planes carry their own validity: a slot holds PLANE_EMPTY (all ones: a NaN no product can be) until its producer stores the tile, and the owner -- the
// only reader -- puts PLANE_EMPTY back once it has summed the slot.  The owner polls the DATA: one round trip instead of flag sweep + payload load, and the producers
// neither drain their stores nor write flags (a 16-byte store becomes visible as a whole; every dword is checked anyway).  The owner's resets are drained by the
// publish that follows them, long before the next layer's producers can write the slots again.
constexpr unsigned PLANE_EMPTY = 0xFFFFFFFFu;
template <int NP, class PlaneF>
__device__ __forceinline__ bool comm_reduce_poll(float* planes, unsigned bytes, PlaneF plane_base, int lane, const float* resid, float* dst, unsigned tag, BCtl* c, unsigned* err) {
    static_assert(NP <= 8, "one batch of loads");
    const srd_t sd = make_srd(planes, bytes);
    const int n12 = min(lane >> 2, OWN - 1), mq = lane & 3;
    const unsigned lo_ = (unsigned)((OWN * blockIdx.x + n12) * BM + 4 * mq) * 4u;
    u32x4 v[NP]; u64 t0 = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < NP; u++) v[u] = ld_frag(sd, (unsigned)plane_base(u) + lo_);
#pragma unroll
        for (int u = 0; u < NP; u++) ok &= v[u].x != PLANE_EMPTY && v[u].y != PLANE_EMPTY && v[u].z != PLANE_EMPTY && v[u].w != PLANE_EMPTY;
        if (__all(ok)) break;
        if (sweep_bail(t0, tag, c, err)) return false;
    }
    f32x4 a = *reinterpret_cast<const f32x4*>(resid + n12 * BM + 4 * mq);
#pragma unroll
    for (int u = 0; u < NP; u++) a += __builtin_bit_cast(f32x4, v[u]);      // fixed order
    if (lane < 4 * OWN) {
        *reinterpret_cast<f32x4*>(dst + n12 * BM + 4 * mq) = a;
        const u32x4 e = {PLANE_EMPTY, PLANE_EMPTY, PLANE_EMPTY, PLANE_EMPTY};
#pragma unroll
        for (int u = 0; u < NP; u++) __builtin_amdgcn_raw_buffer_store_b128(e, sd, (int)((unsigned)plane_base(u) + lo_), 0, 16);
    }
    return true;
}
template <int NP, class PlaneF>
__device__ __forceinline__ void comm_reduce(const float* planes, unsigned bytes, PlaneF plane_base, int lane, const float* resid, float* dst) {
    const srd_t sd = make_srd(planes, bytes);
    const int n12 = min(lane >> 2, OWN - 1), mq = lane & 3;
    const unsigned lo_ = (unsigned)((OWN * blockIdx.x + n12) * BM + 4 * mq) * 4u;
    f32x4 a = *reinterpret_cast<const f32x4*>(resid + n12 * BM + 4 * mq);
#pragma unroll
    for (int r0 = 0; r0 < NP; r0 += 16) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16 && r0 + u < NP; u++) v[u] = ld_frag(sd, (unsigned)plane_base(r0 + u) + lo_);
#pragma unroll
        for (int u = 0; u < 16 && r0 + u < NP; u++) a += __builtin_bit_cast(f32x4, v[u]);      // fixed order
    }
    if (lane < 4 * OWN) *reinterpret_cast<f32x4*>(dst + n12 * BM + 4 * mq) = a;
}

// One function per edge, each of (group q, layer l): the COMM wave serves the groups' chains in the order the consumer waves run their phases (A, B per phase).
// pub(k) = the publish count at which all twelve consumer waves of THIS CU have left phase k (0 q|k|v .. 4 w2) of (l, q).
template <int NG>
__device__ __forceinline__ void b16_comm(BCtl* c, unsigned char* lds, const int lane0, const Tl& tl0) {
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7;
    const float* gwt = reinterpret_cast<const float*>(lds + BL_GW);
    const EngBParams p0 = kparams(0);
    const int L = p0.n_layers;
#pragma unroll 1
    for (int q = 0; q < NG; q++) {   // the step's input rows of this CU's 12 columns (rows >= n_rows: zeros)
        const EngBParams p = kparams(q);
        float* own0 = reinterpret_cast<float*>(lds + L_OWN(q));
        const unsigned tag_base = b16_tag_base(p);
        const int n12 = min(lane0 >> 2, OWN - 1), mq = lane0 & 3;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; r++) if (4 * mq + r < p.n_rows) v[r] = as_g(p.h_in)[(size_t)(4 * mq + r) * p.h_stride + OWN * b + n12];
        if (lane0 < 4 * OWN) *reinterpret_cast<f32x4*>(own0 + n12 * BM + 4 * mq) = v;
        wait_ge(&c->gw_flag, (unsigned)NCONS, c, p.err, ERR_STAGE);
        ENG_CFENCE();
        if (L > 0) comm_publish_rows(p, lane0, own0, gwt, p.XH0, p.SS0, p.F0, tag_base);
        else comm_publish_final(p, lane0, own0, gwt);
    }
    const bool xloc = (p0.flags & 128) != 0 && lds_ld(&c->xcd_ok) != 0;
    const unsigned tb0 = b16_tag_base(p0), tb1 = NG > 1 ? b16_tag_base(kparams(NG - 1)) : 0u;      // the groups' launch serials (two state blocks)
    auto tagb = [&](int q) { return q == 0 ? tb0 : tb1; };
    auto pub = [&](int l, int k, int q) { return (unsigned)NCONS * ((unsigned)(NG * PH) * (unsigned)l + (unsigned)(NG * k + q) + 1u); };
    // ---- all-gather (stage st = 2 l + odd): every owner's flag, the RMSNorm scales, release the consumer waves
    auto ev_gather = [&](int l, bool odd, int q) {
        int lane = lane0; asm volatile("" : "+v"(lane));      // opaque per edge: nothing lane-derived is carried from one edge to the next (the wave has no registers to spare)
        const EngBParams p = kparams(q);
        const unsigned tag = tagb(q) + (unsigned)l + 1u, st = 2u * (unsigned)l + (odd ? 1u : 0u);
        const bool T = l == p.tl_layer; Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32);
        float* rstd = reinterpret_cast<float*>(lds + L_RSTD(q));
        if (T) tl(odd ? 20 : 8);
        lds_st(&c->gathering, 1u);
        poll_flags256(odd ? p.F1 : p.F0, odd ? tag : tag - 1u, lane, c, p.err);
        if (T) tl(odd ? 21 : 9);
        lds_st(&c->ag_flag[q], st + 1u);
        comm_rstd(odd ? p.SS1 : p.SS0, p.eps, lane, rstd + (odd ? BM : 0));
        ENG_CFENCE(); lds_st(&c->rs_flag[q], st + 1u);
        lds_st(&c->gathering, 0u);
    };
#pragma unroll 1
    for (int l = 0; l < L; l++) {
#pragma unroll 1
        for (int q = 0; q < NG; q++) ev_gather(l, false, q);
#pragma unroll 1
        for (int q = 0; q < NG; q++) {   // this step's q (head h), k, v (KV head g) rows of the CU's two sequences
            int lane = lane0; asm volatile("" : "+v"(lane));
            const EngBParams p = kparams(q);
            const unsigned tag = tagb(q) + (unsigned)l + 1u;
            float* qkvn = reinterpret_cast<float*>(lds + L_QKVN(q));
            wait_ge(&c->pub_cnt, pub(l, 0, q), c, p.err, ERR_STAGE);
            lds_st(&c->gathering, 1u);
            float v[12];
            sweep<12>(p.G, BM * G_ROW * 8u, tag, [&](int u) { const int i = lane + 64 * u, t = i / 384, r = i - 384 * t, seg = r >> 7, e = r & 127;
                                                             return (2 * s + t) * G_ROW + (seg == 0 ? 128 * h + e : (seg == 1 ? EQD : EQD + EKD) + 128 * g + e); },
                      [&]() { return 0; }, false, v, c, p.err);
            lds_st(&c->gathering, 0u);
#pragma unroll
            for (int u = 0; u < 12; u++) qkvn[lane + 64 * u] = v[u];
            ENG_CFENCE(); lds_st(&c->qkv_flag[q], (unsigned)l + 1u);
            if (l == p.tl_layer) { Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32); tl(10); }
        }
#pragma unroll 1
        for (int q = 0; q < NG; q++) {   // attention outputs of head h: one flag per sequence, written by the team that computed it
            int lane = lane0; asm volatile("" : "+v"(lane));
            const EngBParams p = kparams(q);
            const unsigned tag = tagb(q) + (unsigned)l + 1u;
            wait_ge(&c->pub_cnt, pub(l, 1, q), c, p.err, ERR_STAGE);      // (own CU through: nothing can be complete much earlier -- no polling while it computes)
            lds_st(&c->gathering, 1u);
            poll_flags(p.FO, 512, 4 * BM, [&](int i) { return 4 * g * BM + i; }, tag, lane, c, p.err);      // the group's four heads x 16 sequences: wo's K range on this CU
            lds_st(&c->gathering, 0u);
            lds_st(&c->wo_flag[q], (unsigned)l + 1u);
            if (l == p.tl_layer) { Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32); tl(11); }
        }
#pragma unroll 1
        for (int q = 0; q < NG; q++) {   // wo: the 8 planes (one per XCD group) of the CU's 12 columns, fixed-order sum + residual -> the post-attention stream, published as the w1|w3 input
            int lane = lane0; asm volatile("" : "+v"(lane));
            const EngBParams p = kparams(q);
            const unsigned tag = tagb(q) + (unsigned)l + 1u;
            float* own0 = reinterpret_cast<float*>(lds + L_OWN(q)); float* own1 = own0 + OWN * BM;
            const bool T = l == p.tl_layer; Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32);
            wait_ge(&c->pub_cnt, pub(l, 2, q), c, p.err, ERR_STAGE);
            lds_st(&c->gathering, 1u);
            comm_reduce_poll<NPWB>(p.PW, NPWB * ED * BM * 4u, [&](int pl) { return (unsigned)pl * (ED * BM * 4u); }, lane, own0, own1, tag, c, p.err);
            if (T) tl(13);
            lds_st(&c->gathering, 0u);
            ENG_CFENCE();
            if (!((p.flags & 16384) && b == 7 && l == 1 && q == 0))      // (flag 16384 = FAULT INJECTION: workgroup 7 loses a publish)
                comm_publish_rows(p, lane, own1, gwt + (l * 2 + 1) * 16, p.XH1, p.SS1, p.F1, tag);
            if (T) tl(14);
        }
#pragma unroll 1
        for (int q = 0; q < NG; q++) ev_gather(l, true, q);
#pragma unroll 1
        for (int q = 0; q < NG; q++) {   // SwiGLU outputs of the XCD group: this CU's flag, then the group's 32
            int lane = lane0; asm volatile("" : "+v"(lane));
            const EngBParams p = kparams(q);
            const unsigned tag = tagb(q) + (unsigned)l + 1u;
            wait_ge(&c->pub_cnt, pub(l, 3, q), c, p.err, ERR_STAGE);
            if (lane == 0) st_u32(make_srd(p.FA, NCU * 4u), (unsigned)(g * 32 + j) * 4u, tag, xloc);
            lds_st(&c->gathering, 1u);
            poll_flags(p.FA, NCU, 32, [&](int i) { return g * 32 + i; }, tag, lane, c, p.err);
            lds_st(&c->gathering, 0u);
            lds_st(&c->xa_flag[q], (unsigned)l + 1u);
            if (l == p.tl_layer) { Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32); tl(22); }
        }
#pragma unroll 1
        for (int q = 0; q < NG; q++) {   // w2: the 8 planes -> the layer's output; published as the next layer's q|k|v input, or (last layer) as the lm_head launch's input
            int lane = lane0; asm volatile("" : "+v"(lane));
            const EngBParams p = kparams(q);
            const unsigned tag = tagb(q) + (unsigned)l + 1u;
            float* own0 = reinterpret_cast<float*>(lds + L_OWN(q)); float* own1 = own0 + OWN * BM;
            const bool T = l == p.tl_layer; Tl tl = tl0; tl.buf += (size_t)q * (NCU * 32);
            wait_ge(&c->pub_cnt, pub(l, 4, q), c, p.err, ERR_STAGE);
            if (T) tl(23);
            lds_st(&c->gathering, 1u);
            comm_reduce_poll<NP2>(p.P2, NP2 * ED * BM * 4u, [&](int pl) { return (unsigned)pl * (ED * BM * 4u); }, lane, own1, own0, tag, c, p.err);
            if (T) tl(24);
            lds_st(&c->gathering, 0u);
            ENG_CFENCE();
            if (l + 1 < L) comm_publish_rows(p, lane, own0, gwt + ((l + 1) * 2) * 16, p.XH0, p.SS0, p.F0, tag);
            else comm_publish_final(p, lane, own0, gwt + (L * 2) * 16);
            if (T) tl(25);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// CONSUMER waves
// ------------------------------------------------------------------------------------------------
// Every phase of a layer is its own function of (consumer wave, lane, layer) and NOTHING else: packet and barrier counters are functions of the layer, the launch
// parameters are re-read from the kernarg segment with scalar loads, LDS is addressed from its base, and the lane index is made opaque at every phase entry.  Written as
// one loop body with shared state, hipcc hoisted every phase's loop-invariant, lane-derived values out of the layer loop and carried them across all the other phases:
// 180 VGPRs spilled into the hot loops (each phase alone fits: q|k|v and w1|w3 128 VGPRs with the wave's 64 registers of A fragments, attention 114).  Calling the
// phases (noinline) also ends every live range but costs 48 callee-saved VGPRs stored and reloaded per call -- 1.4 MB of scratch traffic per CU and layer; -DB16_NOINLINE
// keeps that form for comparison.
#ifndef B16_NOINLINE
#define B16_PHASE __device__ __forceinline__
#else
#define B16_PHASE __device__ __attribute__((noinline))
#endif
constexpr int CB_LAYER = 5;      // workgroup barriers among the consumer waves per layer: q|k|v, wo, w1|w3 (2), w2

struct BCons {
    const EngBParams& p; BCtl* c; unsigned char* lds; int cw, lane;
    __device__ __forceinline__ BCons(const EngBParams& p_, unsigned char* lds_, int cw_, int lane_) : p(p_), c(reinterpret_cast<BCtl*>(lds_ + BL_CTL)), lds(lds_), cw(cw_), lane(lane_) {}
    __device__ __forceinline__ void cbarrier(unsigned target) {      // target = NCONS * (barriers so far)
        ENG_CFENCE();
        if (lane == 0) __hip_atomic_fetch_add(&c->cbar, 1u, RLX, WG);
        wait_ge(&c->cbar, target, c, p.err, ERR_CBAR);
    }
    __device__ __forceinline__ void tbarrier(int team, unsigned target) {      // the six waves of an attention team
        ENG_CFENCE();
        if (lane == 0) __hip_atomic_fetch_add(&c->tbar[team], 1u, RLX, WG);
        wait_ge(&c->tbar[team], target, c, p.err, ERR_CBAR);
    }
    __device__ __forceinline__ void published() { ENG_CFENCE(); if (lane == 0) __hip_atomic_fetch_add(&c->pub_cnt, 1u, RLX, WG); }
    __device__ __forceinline__ const unsigned char* slot_wait(unsigned pk, int share_bytes, int& slot) {
        slot = (int)(pk % BNSLOT); const unsigned k = pk / BNSLOT;
        wait_ge(&c->ring_ready[slot], k + 1u, c, p.err, ERR_RING);
        return lds + BL_RING + slot * SLOT_BYTES + cw * share_bytes;
    }
    __device__ __forceinline__ void slot_release(int slot) { ENG_CFENCE(); if (lane == 0) __hip_atomic_fetch_add(&c->ring_done[slot], 1u, RLX, WG); }
    // NB activation blocks blk0 .. blk0 + NB - 1 of a fragment buffer -> registers (hi, lo), and their -136 sum(x) rows -> this wave's CB lines.  Two halves: the loads
    // (a_issue) and what needs their data (a_finish).  One group per launch: both behind the records, which wait in registers for the edge.  Two groups: the edge has
    // usually resolved while the other group's phase ran, so the loads go out FIRST and the records are fetched from the ring under their latency.
    template <int NB>
    __device__ __forceinline__ void a_issue(const unsigned char* buf, unsigned bytes, int blk0, u32x4 (&rh)[NB], u32x4 (&rl)[NB]) {
        const srd_t sd = make_srd(buf, bytes);
#pragma unroll
        for (int i = 0; i < NB; i++) { rh[i] = ld_frag(sd, (unsigned)(((blk0 + i) * 2) * FRAG + lane * 16)); rl[i] = ld_frag(sd, (unsigned)(((blk0 + i) * 2 + 1) * FRAG + lane * 16)); }
    }
    template <int NB>
    __device__ __forceinline__ void a_finish(const u32x4 (&rh)[NB], const u32x4 (&rl)[NB], bf16x8 (&ah)[NB], bf16x8 (&al)[NB]) {
        float* cb = reinterpret_cast<float*>(lds + BL_CB) + cw * 8 * BM;
        unsigned k136 = 0xC308C308u; asm volatile("" : "+v"(k136));      // bf16(-136) x 2, opaque: rebuilt here (4 v_mov), never carried across phases
        u32x4 mm; mm.x = mm.y = mm.z = mm.w = k136;
#pragma unroll
        for (int i = 0; i < NB; i++) {
            ah[i] = as_bf16x8(rh[i]); al[i] = as_bf16x8(rl[i]);
            const f32x4 cs = corr16(ah[i], al[i], mm);
            if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(cb + i * BM + 4 * (lane >> 4)) = cs;
        }
    }
    __device__ __forceinline__ f32x4 cb_read(int i) const { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(lds + BL_CB) + (cw * 8 + i) * BM + 4 * (lane >> 4)); }
};
#define B16_PROLOGUE                                                                                                          \
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));      /* the lane index from the EXEC mask (all 64 lanes are active here), as VOLATILE asm: the builtin form was computed once before the layer loop, spilled, and reloaded (scratch_load + vmcnt(0)) at every phase entry */ \
    const int q = NG > 1 ? __builtin_amdgcn_readfirstlane(q_) : 0;                                                            \
    const EngBParams p = kparams(q); unsigned char* lds = lds_base();                                                         \
    const int cw = __builtin_amdgcn_readfirstlane(cw_), l = __builtin_amdgcn_readfirstlane(l_);                               \
    BCtl* c = reinterpret_cast<BCtl*>(lds + BL_CTL); BCons cs(p, lds, cw, lane);                                             \
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7, n = lane & 15, y = lane >> 4;           \
    const unsigned tag = b16_tag_base(p) + (unsigned)l + 1u;                                                                  \
    /* the one global order of the launch: phases, consumer barriers and ring packets before layer l (NG groups per phase) */  \
    const unsigned ordL = (unsigned)l * (NG * PH), cbL = (unsigned)l * (NG * CB_LAYER), pkL = (unsigned)l * (NG * PK_LAYER);  \
    const bool xloc = (p.flags & 128) != 0 && lds_ld(&c->xcd_ok) != 0;                                                       \
    const EngLayerTab* Lt = reinterpret_cast<const EngLayerTab*>(lds + BL_TAB) + l;                                           \
    Tl tl; tl.on = p.tl != nullptr && lane == 0 && l == p.tl_layer && cw == 0; tl.buf = p.tl ? p.tl + (size_t)blockIdx.x * 32 + (size_t)q * (NCU * 32) : nullptr;      /* (group B's stamps: a second [256][32] table) */ \
    (void)g; (void)j; (void)h; (void)s; (void)n; (void)y; (void)tag; (void)ordL; (void)cbL; (void)pkL; (void)xloc; (void)Lt;
// NG > 1: before a wave's first store into a time-shared LDS region of phase k, all twelve waves have left the previous phase of the global order (its finishing waves
// may still have been reading the region: there is no cross-CU edge between group A's phase and group B's).  One LDS poll; the records, the edge and the MFMAs come first.
#define B16_GUARD(k) do { if (NG > 1) wait_ge(&c->pub_cnt, (unsigned)NCONS * (ordL + (unsigned)(NG * (k)) + (unsigned)q), c, p.err, ERR_CBAR); } while (0)

// ================= q|k|v: 16 + 8 weight rows x K 3072, K split over the 12 waves =================
template <int NG, bool KVR>
B16_PHASE void ph_qkv(int cw_, int lane, int l_, int q_) {
    B16_PROLOGUE
    unsigned char* part = lds + BL_PART;
    const unsigned P0 = pkL + (unsigned)(q * QKV_PK);
    {
        // this wave's 8 step records (K-steps 4 cw .. 4 cw + 3 of the 16-row q tile and of the 8-row k|v tile) -> registers BEFORE the input exists; the slots are free again
        RawRec rq[4]; u32x4 xh[8], xl[8];
        if (NG > 1) { wait_ge(&c->ag_flag[q], 2u * (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XH0, XH_BYTES, 8 * cw, xh, xl); }
#pragma unroll
        for (int pk = 0; pk < 2; pk++) {
            int sl; const unsigned char* bb = cs.slot_wait(P0 + pk, 2 * REC, sl);
            rq[2 * pk] = rec_load(bb, false, lane); rq[2 * pk + 1] = rec_load(bb + REC, false, lane);
            cs.slot_release(sl);
        }
        if (NG == 1) { wait_ge(&c->ag_flag[q], 2u * (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XH0, XH_BYTES, 8 * cw, xh, xl); }
        bf16x8 ah[8], al[8];
        cs.a_finish<8>(xh, xl, ah, al);
        tl(0);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        int sl2; const unsigned char* b2 = cs.slot_wait(P0 + 2, 4 * REC_H, sl2);      // the 8-row tile's records stay in the ring (landed long ago)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const f32x4 c0 = cs.cb_read(2 * r), c1 = cs.cb_read(2 * r + 1);
            a0 = rec_mma(rq[r], ah[2 * r], al[2 * r], ah[2 * r + 1], al[2 * r + 1], c0, c1, a0);
            a1 = rec_mma(rec_load(b2 + r * REC_H, true, lane), ah[2 * r], al[2 * r], ah[2 * r + 1], al[2 * r + 1], c0, c1, a1);
        }
        cs.slot_release(sl2);
        B16_GUARD(0);
        *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + lane * 16) = a0;
        if (n < 8) *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 1024 + (y * 8 + n) * 16) = a1;
    }
    cs.cbarrier((unsigned)NCONS * (cbL + (unsigned)q + 1u));
    if (cw < 2) {      // wave 0: the 16 q rows; wave 1: 4 k + 4 v rows -- sum the 12 K slices (fixed order), RMSNorm scale, RoPE at each sequence's position, publish
        const float* rstd = reinterpret_cast<const float*>(lds + L_RSTD(q));
        const float* ropef = reinterpret_cast<const float*>(lds + L_ROPE(q));      // q cos [8][16] | q sin [8][16] | k cos [2][16] | k sin [2][16]
        const int* posl = reinterpret_cast<const int*>(lds + L_POS(q));
        const int* kvr = reinterpret_cast<const int*>(lds + L_KVR(q));
        const int nn = cw == 0 ? n : (n & 7);
        wait_ge(&c->rs_flag[q], 2u * (unsigned)l + 1u, c, p.err, ERR_STAGE);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NCONS; w++) a += *reinterpret_cast<const f32x4*>(part + w * PART_WAVE + (cw == 0 ? lane * 16 : 1024 + (y * 8 + nn) * 16));
        a *= *reinterpret_cast<const f32x4*>(rstd + 4 * y);
        const bool rot = cw == 0 || nn < 4;
        const float* rcp = ropef + (cw == 0 ? 0 : 16 * BM) + (nn >> 1) * BM + 4 * y;
        const f32x4 rc = rot ? *reinterpret_cast<const f32x4*>(rcp) : (f32x4){1.f, 1.f, 1.f, 1.f};
        const f32x4 rs = rot ? *reinterpret_cast<const f32x4*>(rcp + (cw == 0 ? 8 * BM : 2 * BM)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const srd_t gd = make_srd(p.G, BM * G_ROW * 8u);
        const gf_p kc = as_g(Lt->kc), vc = as_g(Lt->vc);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float o = dppf<0xB1>(a[r]);
            const float yv = (nn & 1) ? fmaf(o, rs[r], a[r] * rc[r]) : fmaf(-o, rs[r], a[r] * rc[r]);      // interleaved-pair RoPE (rope.rs:99-141); identity for v
            const int m = 4 * y + r;
            if (cw == 0) st_u64(gd, (unsigned)(m * G_ROW + 128 * h + 16 * s + n) * 8u, __float_as_uint(yv), tag, xloc);
            else if (n < 8) {
                const int col = 128 * g + 4 * j + (nn & 3);
                st_u64(gd, (unsigned)(m * G_ROW + (nn < 4 ? EQD : EQD + EKD) + col) * 8u, __float_as_uint(yv), tag, xloc);
                if (m < p.n_rows) (nn < 4 ? kc : vc)[(size_t)(KVR ? kvr[m] : m) * p.kv_seq_stride + (size_t)g * p.max_seq * EHD + (size_t)posl[m] * EHD + 4 * j + (nn & 3)] = yv;      // the cache rows later steps read
            }
        }
    }
    cs.published();
    tl(1);
}

// ================= attention: head h of sequence 2 s + team, six waves, per-wave online softmax over the wave's keys =================
template <int NG, bool KVR>
B16_PHASE void ph_attn(int cw_, int lane, int l_, int q_) {
    B16_PROLOGUE
    const int* posl = reinterpret_cast<const int*>(lds + L_POS(q));
    const float* qkvn = reinterpret_cast<const float*>(lds + L_QKVN(q));
    float* po = reinterpret_cast<float*>(lds + BL_PO); float* ml = reinterpret_cast<float*>(lds + BL_ML); float* ofin = reinterpret_cast<float*>(lds + BL_OF);
    const int team = cw / 6, tw = cw - 6 * team, msq = 2 * s + team;
    const bool seq_ok = msq < p.n_rows;
    const float scale = 1.0f / sqrtf((float)EHD);
    const int pos = __builtin_amdgcn_readfirstlane(posl[msq]);
    const int j_lo = p.window >= 0 ? max(0, pos - p.window) : 0, n_old = seq_ok ? pos - j_lo : 0, last_old = max(n_old - 1, 0);
    // uniform bases in SGPRs (the table lives in LDS: a ds_read result is a VGPR to the compiler) -> one 32-bit VGPR offset per load
    const int kvrow = KVR ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lds + L_KVR(q))[msq]) : min(msq, max(p.n_rows - 1, 0));      // (KVR: rows >= n_rows hold a valid slice too, see the start-up)
    const size_t kvo = (size_t)kvrow * p.kv_seq_stride + (size_t)g * p.max_seq * EHD;
    const u64 kcb = (u64)(uintptr_t)Lt->kc, vcb = (u64)(uintptr_t)Lt->vc;
    const gf_p kc = (gf_p)(uintptr_t)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kcb >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kcb)) + kvo;
    const gf_p vc = (gf_p)(uintptr_t)(((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(vcb >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)vcb)) + kvo;
    const int part_ = lane & 7, kk = lane >> 3;
    // a ROUND = 96 keys of the team = 16 per wave (two sub-passes of 8): K as 8 lanes x 16 columns per key, V as 64 lanes x 2 columns per key; all 64 registers
    // of a round are requested at once.  Round 0 does not depend on this step: it is in flight while the q|k|v edge resolves.
    const int nround = (n_old + 95) / 96;
    float4 kA[2][4]; float2 vA[2][8];
    auto kvload = [&](int r) {
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const int k0 = 96 * r + 48 * sp + 8 * tw;
            const unsigned ko = (unsigned)(j_lo + min(k0 + kk, last_old)) * EHD + part_ * 16;
#pragma unroll
            for (int e = 0; e < 4; e++) kA[sp][e] = ldg4(kc + (ko + 4 * e));
#pragma unroll
            for (int e = 0; e < 8; e++) { const fv2 v = *(const __attribute__((address_space(1))) fv2*)(vc + ((unsigned)(j_lo + min(k0 + e, last_old)) * EHD + lane * 2)); vA[sp][e] = make_float2(v.x, v.y); }
        }
    };
    if (nround > 0) kvload(0);
    wait_ge(&c->qkv_flag[q], (unsigned)l + 1u, c, p.err, ERR_STAGE);
    tl(2);
    const float* qn = qkvn + team * 384;
    float qv[16];
#pragma unroll
    for (int e = 0; e < 4; e++) { const float4 v = *reinterpret_cast<const float4*>(qn + part_ * 16 + 4 * e); qv[4 * e] = v.x; qv[4 * e + 1] = v.y; qv[4 * e + 2] = v.z; qv[4 * e + 3] = v.w; }
    auto dot16 = [&](const float4 (&kq)[4]) {
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { sacc = fmaf(qv[4 * e], kq[e].x, sacc); sacc = fmaf(qv[4 * e + 1], kq[e].y, sacc); sacc = fmaf(qv[4 * e + 2], kq[e].z, sacc); sacc = fmaf(qv[4 * e + 3], kq[e].w, sacc); }
        return group8_sum_e(sacc);
    };
    float m_run = -INFINITY, l_run = 0.f; float2 o = make_float2(0.f, 0.f);
#pragma unroll 1
    for (int r = 0; r < nround; r++) {
        if (r > 0) kvload(r);
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            const int i = 96 * r + 48 * sp + 8 * tw + kk;
            const float sv = i < n_old ? dot16(kA[sp]) * scale : -INFINITY;
            const float pm = wave_max_e(sv);
            if (pm > -INFINITY) {
                const float mn = fmaxf(m_run, pm), alpha = expf(m_run - mn), pe = expf(sv - mn);
                l_run *= alpha; o.x *= alpha; o.y *= alpha;
#pragma unroll
                for (int e = 0; e < 8; e++) { const float pr_ = rlf(pe, 8 * e); l_run += pr_; o.x = fmaf(pr_, vA[sp][e].x, o.x); o.y = fmaf(pr_, vA[sp][e].y, o.y); }
                m_run = mn;
            }
        }
    }
    if (tw == 0) {      // the new key (this step's k / v row)
        float4 kq[4];
#pragma unroll
        for (int e = 0; e < 4; e++) kq[e] = *reinterpret_cast<const float4*>(qn + 128 + part_ * 16 + 4 * e);
        const float sv = dot16(kq) * scale;      // the same in every lane group
        const float mn = fmaxf(m_run, sv), alpha = expf(m_run - mn), pe = expf(sv - mn);
        const float2 vv = *reinterpret_cast<const float2*>(qn + 256 + lane * 2);
        l_run = l_run * alpha + pe; o.x = fmaf(pe, vv.x, o.x * alpha); o.y = fmaf(pe, vv.y, o.y * alpha);
        m_run = mn;
    }
    B16_GUARD(1);
    *reinterpret_cast<float2*>(po + cw * 128 + lane * 2) = o;
#ifdef VOX_PK_AS_COMPILED
    if (lane == 0) { ml[2 * cw] = m_run; ml[2 * cw + 1] = l_run; }
#else
    if (lane == 0) { volatile float* mlv = ml; mlv[2 * cw] = m_run; mlv[2 * cw + 1] = l_run; }      // (two 4-byte stores: as one 8-byte store hipcc swapped the pair with v_pk_mov_b32 op_sel:[1,0] -- vox_kernels.h, VOX_NO_PK_F32)
#endif
    cs.tbarrier(team, 6u * ((unsigned)(NG * l + q) + 1u));
    if (tw == 0) {      // combine the team's six partials (fixed order), normalise, publish head h's output of sequence msq as wo's A fragments (hi + lo)
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 6; w++) M = fmaxf(M, ml[2 * (6 * team + w)]);
        float Ls = 0.f; float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 6; w++) {
            const float f = expf(ml[2 * (6 * team + w)] - M);
            const float2 ov = *reinterpret_cast<const float2*>(po + (6 * team + w) * 128 + lane * 2);
            Ls = fmaf(ml[2 * (6 * team + w) + 1], f, Ls); acc.x = fmaf(ov.x, f, acc.x); acc.y = fmaf(ov.y, f, acc.y);
        }
        const float inv = seq_ok ? 1.0f / Ls : 0.f;
        *reinterpret_cast<float2*>(ofin + team * 128 + lane * 2) = make_float2(acc.x * inv, acc.y * inv);
        ENG_CFENCE();
        if (lane < 16) {      // lane = octet of the head's 128 columns
            const float4 v0 = *reinterpret_cast<const float4*>(ofin + team * 128 + 8 * lane), v1 = *reinterpret_cast<const float4*>(ofin + team * 128 + 8 * lane + 4);
            unsigned hq[4], lq[4];
            split_pair(v0.x, v0.y, hq[0], lq[0]); split_pair(v0.z, v0.w, hq[1], lq[1]); split_pair(v1.x, v1.y, hq[2], lq[2]); split_pair(v1.z, v1.w, hq[3], lq[3]);
            u32x4 hi, lo; hi.x = hq[0]; hi.y = hq[1]; hi.z = hq[2]; hi.w = hq[3]; lo.x = lq[0]; lo.y = lq[1]; lo.z = lq[2]; lo.w = lq[3];
            const srd_t od = make_srd(p.XO + (size_t)h * XO_HEAD, XO_HEAD);
            const unsigned o0 = frag_off(lane >> 2, 0, lane & 3, msq), o1 = frag_off(lane >> 2, 1, lane & 3, msq);
            if (xloc) { __builtin_amdgcn_raw_buffer_store_b128(hi, od, (int)o0, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(lo, od, (int)o1, 0, 0); }
            else { __builtin_amdgcn_raw_buffer_store_b128(hi, od, (int)o0, 0, 16); __builtin_amdgcn_raw_buffer_store_b128(lo, od, (int)o1, 0, 16); }
        }
        drain_vm();
        if (lane == 0) st_u32(make_srd(p.FO, 512 * 4u), (unsigned)(h * BM + msq) * 4u, tag, xloc);
    }
    cs.published();
    tl(3);
}

// ================= wo: rows [96 j, +96) x the 512 columns of the XCD group's four heads -> plane g =================
template <int NG, bool KVR>
B16_PHASE void ph_wo(int cw_, int lane, int l_, int q_) {
    B16_PROLOGUE
    unsigned char* part = lds + BL_PART;
    const unsigned P0 = pkL + (unsigned)(NG * QKV_PK + q * WO_PK);
    {
        const int tile = cw % 6, kh = cw / 6;
        RawRec rw[4]; u32x4 xh[8], xl[8];      // tile cw % 6, K-steps 4 (cw / 6) .. + 3: in registers before the attention outputs exist
        if (NG > 1) { wait_ge(&c->wo_flag[q], (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XO + (size_t)(4 * g) * XO_HEAD, 4 * XO_HEAD, 8 * kh, xh, xl); }
#pragma unroll
        for (int i = 0; i < WO_PK; i++) {
            int sl; const unsigned char* bb = cs.slot_wait(P0 + i, 2 * REC, sl);
            rw[2 * i] = rec_load(bb, false, lane); rw[2 * i + 1] = rec_load(bb + REC, false, lane);
            cs.slot_release(sl);
        }
        if (NG == 1) { wait_ge(&c->wo_flag[q], (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XO + (size_t)(4 * g) * XO_HEAD, 4 * XO_HEAD, 8 * kh, xh, xl); }      // heads 4 g .. 4 g + 3 are contiguous: 16 blocks, this wave's half
        bf16x8 ah[8], al[8];
        cs.a_finish<8>(xh, xl, ah, al);
        tl(4);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; r++) a = rec_mma(rw[r], ah[2 * r], al[2 * r], ah[2 * r + 1], al[2 * r + 1], cs.cb_read(2 * r), cs.cb_read(2 * r + 1), a);
        // (q|k|v's partials in this region are dead: its finishing waves published before attention could start; w1|w3 writes here only after the h1 all-gather, which
        // needs every wo tile of every CU stored)
        B16_GUARD(2);
        *reinterpret_cast<f32x4*>(part + (tile * 2 + kh) * 1024 + lane * 16) = a;
    }
    cs.cbarrier((unsigned)NCONS * (cbL + (unsigned)(NG + q) + 1u));
    if (cw < 6) {      // wave t finishes tile t: the two K halves in a fixed order
        const f32x4 a = *reinterpret_cast<const f32x4*>(part + (cw * 2) * 1024 + lane * 16) + *reinterpret_cast<const f32x4*>(part + (cw * 2 + 1) * 1024 + lane * 16);
        st_f4(make_srd(p.PW + (size_t)g * ED * BM, ED * BM * 4u), (unsigned)((96 * j + 16 * cw + n) * BM + 4 * y) * 4u, a);      // the owner polls the slots themselves (comm_reduce_poll)
    }
    cs.published();
    tl(5);
}

// ================= w1|w3: 72 interleaved gate / up rows x K 3072 -> 36 SwiGLU outputs x 16 sequences =================
template <int NG, bool KVR>
B16_PHASE void ph_w13(int cw_, int lane, int l_, int q_) {
    B16_PROLOGUE
    unsigned char* part = lds + BL_PART;
    // SwiGLU outputs [16 m][36].  One group: in the CB lines, which are dead behind the first barrier.  NG > 1: the other group's pass writes ITS CB lines while waves 0..2
    // still read these -> in the group's own q / k / v block instead (3 KB, dead between the group's attention and the next layer's granule sweep)
    float* sg = reinterpret_cast<float*>(lds + (NG == 1 ? BL_CB : L_QKVN(q)));
    {
        // tile 0's and tile 1's records (8 of the wave's 20) wait in registers while the post-attention stream is still on its way; afterwards one tile is fetched ahead
        // of the one being multiplied
        const unsigned PW13 = pkL + (unsigned)(NG * (QKV_PK + WO_PK) + q * W13_PK);
        RawRec ra[4], rb[4];
        // (the two f16 scales of a record are kept as ONE register -- opaque to the compiler, which otherwise holds the halves in two: with eight records waiting in
        // registers next to the 64 registers of A fragments that was 8 VGPRs too many, spilled and reloaded inside the multiply loop)
        auto fetch_tile = [&](int ti, RawRec (&rr)[4]) {
#pragma unroll
            for (int pk = 0; pk < 2; pk++) {
                int sl; const unsigned char* bb = cs.slot_wait(PW13 + 2 * ti + pk, 2 * REC, sl);
                rr[2 * pk] = rec_load(bb, false, lane); rr[2 * pk + 1] = rec_load(bb + REC, false, lane);
                asm volatile("" : "+v"(rr[2 * pk].sc)); asm volatile("" : "+v"(rr[2 * pk + 1].sc));
                cs.slot_release(sl);
            }
        };
        auto fetch_half = [&](RawRec (&rr)[4]) {
            int sl; const unsigned char* bb = cs.slot_wait(PW13 + 8, 4 * REC_H, sl);
#pragma unroll
            for (int r = 0; r < 4; r++) { rr[r] = rec_load(bb + r * REC_H, true, lane); asm volatile("" : "+v"(rr[r].sc)); }
            cs.slot_release(sl);
        };
        u32x4 xh[8], xl[8];
        if (NG > 1) { wait_ge(&c->ag_flag[q], 2u * (unsigned)l + 2u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XH1, XH_BYTES, 8 * cw, xh, xl); }
        fetch_tile(0, ra); fetch_tile(1, rb);
        if (NG == 1) { wait_ge(&c->ag_flag[q], 2u * (unsigned)l + 2u, c, p.err, ERR_STAGE); cs.a_issue<8>(p.XH1, XH_BYTES, 8 * cw, xh, xl); }
        bf16x8 ah[8], al[8];
        cs.a_finish<8>(xh, xl, ah, al);
        tl(6);
        B16_GUARD(3);
        auto mul_tile = [&](const RawRec (&rr)[4]) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; r++) a = rec_mma(rr[r], ah[2 * r], al[2 * r], ah[2 * r + 1], al[2 * r + 1], cs.cb_read(2 * r), cs.cb_read(2 * r + 1), a);
            return a;
        };
        { const f32x4 a = mul_tile(ra); *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 0 * 1024 + lane * 16) = a; }
        fetch_tile(2, ra);
        { const f32x4 a = mul_tile(rb); *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 1 * 1024 + lane * 16) = a; }
        fetch_tile(3, rb);
        { const f32x4 a = mul_tile(ra); *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 2 * 1024 + lane * 16) = a; }
        fetch_half(ra);
        { const f32x4 a = mul_tile(rb); *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 3 * 1024 + lane * 16) = a; }
        { const f32x4 a = mul_tile(ra); if (n < 8) *reinterpret_cast<f32x4*>(part + cw * PART_WAVE + 4096 + (y * 8 + n) * 16) = a; }
    }
    tl(27);
    cs.cbarrier((unsigned)NCONS * (cbL + (unsigned)(2 * NG + 2 * q) + 1u));      // (also: every wave is through with its CB lines -> the region becomes the SwiGLU scratch)
    tl(28);
    if (cw < 5) {       // wave t finishes tile t: K slices summed in a fixed order, RMSNorm scale, SiLU(gate) * up -> sg[m][col]
        const float* rstd = reinterpret_cast<const float*>(lds + L_RSTD(q));
        const bool hf = cw == 4;
        wait_ge(&c->rs_flag[q], 2u * (unsigned)l + 2u, c, p.err, ERR_STAGE);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NCONS; w++) a += *reinterpret_cast<const f32x4*>(part + w * PART_WAVE + (hf ? 4096 + (y * 8 + (n & 7)) * 16 : cw * 1024 + lane * 16));
        a *= *reinterpret_cast<const f32x4*>(rstd + BM + 4 * y);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float up = dppf<0xB1>(a[r]);
            if ((n & 1) == 0 && (!hf || n < 8)) sg[(4 * y + r) * 36 + 8 * cw + (n >> 1)] = silu_e(a[r]) * up;
        }
    }
    tl(29);
    cs.cbarrier((unsigned)NCONS * (cbL + (unsigned)(2 * NG + 2 * q) + 2u));
    tl(30);
    if (cw < 3 && lane < 48) {      // 144 pieces of four columns x one sequence -> bf16 hi + lo, 8 bytes each, into the group's fragment buffer
        const int it = 48 * cw + lane, m = it & 15, hc = it >> 4;
        const float4 v = *reinterpret_cast<const float4*>(sg + m * 36 + 4 * hc);
        unsigned h0, l0, h1, l1; split_pair(v.x, v.y, h0, l0); split_pair(v.z, v.w, h1, l1);
        const int col = 36 * j + 4 * hc, blk = col >> 5, yy = (col & 31) >> 3, hh = (col >> 2) & 1;
        const srd_t ad = make_srd(p.XA + (size_t)g * XA_GROUP, XA_GROUP);
        st_u64(ad, frag_off(blk, 0, yy, m) + 8u * hh, h0, h1, xloc);
        st_u64(ad, frag_off(blk, 1, yy, m) + 8u * hh, l0, l1, xloc);
        drain_vm();
    }
    cs.published();
    tl(7);
}

// ================= w2: rows [96 j, +96) x the XCD group's 1152 columns -> plane g =================
template <int NG, bool KVR>
B16_PHASE void ph_w2(int cw_, int lane, int l_, int q_) {
    B16_PROLOGUE
    unsigned char* part = lds + BL_PART;
    const unsigned P0 = pkL + (unsigned)(NG * PK_LAYER_M + q * W2_PK);
    {
        const int ksl = cw % 6, tg = cw / 6;
        RawRec r2[9]; u32x4 xh[6], xl[6];      // this wave's 3 K-steps of its 3 tiles -> registers while the SwiGLU outputs of the XCD group are still being exchanged
        if (NG > 1) { wait_ge(&c->xa_flag[q], (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<6>(p.XA + (size_t)g * XA_GROUP, XA_GROUP, 6 * ksl, xh, xl); }
#pragma unroll
        for (int i = 0; i < W2_PK; i++) {
            int sl; const unsigned char* bb = cs.slot_wait(P0 + i, 3 * REC, sl);
#pragma unroll
            for (int k = 0; k < 3; k++) r2[3 * i + k] = rec_load(bb + k * REC, false, lane);
            cs.slot_release(sl);
        }
        if (NG == 1) { wait_ge(&c->xa_flag[q], (unsigned)l + 1u, c, p.err, ERR_STAGE); cs.a_issue<6>(p.XA + (size_t)g * XA_GROUP, XA_GROUP, 6 * ksl, xh, xl); }
        bf16x8 ah[6], al[6];
        cs.a_finish<6>(xh, xl, ah, al);
        tl(26);
        B16_GUARD(4);
#pragma unroll
        for (int i = 0; i < W2_PK; i++) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 3; k++) a = rec_mma(r2[3 * i + k], ah[2 * k], al[2 * k], ah[2 * k + 1], al[2 * k + 1], cs.cb_read(2 * k), cs.cb_read(2 * k + 1), a);
            *reinterpret_cast<f32x4*>(part + ((3 * tg + i) * 6 + ksl) * 1024 + lane * 16) = a;
        }
    }
    cs.cbarrier((unsigned)NCONS * (cbL + (unsigned)(4 * NG + q) + 1u));
    if (cw < 6) {      // wave t finishes tile t
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 6; w++) a += *reinterpret_cast<const f32x4*>(part + (cw * 6 + w) * 1024 + lane * 16);
        st_f4(make_srd(p.P2 + (size_t)g * ED * BM, ED * BM * 4u), (unsigned)((96 * j + 16 * cw + n) * BM + 4 * y) * 4u, a);
    }
    cs.published();
    tl(15);
}

template <int NG, bool KVR>
__device__ __forceinline__ void b16_consumer(BCtl* c, unsigned char* lds, int cw, const int lane0_) {
    const int b = blockIdx.x, g = b & 7, j = b >> 3, s = j & 7;
    const EngBParams p = kparams(0);
    const unsigned tag_base = b16_tag_base(p);
    const int half = EHD / 2;
    {   // start-up: positions, RoPE factors (wave 0: the CU's 8 q pairs; wave 1: its 2 k pairs), cache slices, norm-weight table, XCD check
#pragma unroll 1
        for (int q = 0; q < NG; q++) {
            int lane0 = lane0_; asm volatile("" : "+v"(lane0));
            const EngBParams pq = kparams(q);
            float* ropef = reinterpret_cast<float*>(lds + L_ROPE(q));      // q cos [8][16] | q sin [8][16] | k cos [2][16] | k sin [2][16]
            int* posl = reinterpret_cast<int*>(lds + L_POS(q));
            if (cw == 0) {
                const int m = lane0 & 15;
                const int pm = m < pq.n_rows ? pq.pos[m] : 0;
                if (lane0 < 16) posl[m] = pm;
#pragma unroll
                for (int u = 0; u < 2; u++) { const int pr = 2 * (lane0 >> 4) + u; ropef[pr * BM + m] = pq.rope_cos[(size_t)pm * half + 8 * s + pr]; ropef[8 * BM + pr * BM + m] = pq.rope_sin[(size_t)pm * half + 8 * s + pr]; }
            }
            if (cw == 1 && lane0 < 32) {
                const int m = lane0 & 15, pr = lane0 >> 4;
                const int pm = m < pq.n_rows ? pq.pos[m] : 0;
                ropef[16 * BM + pr * BM + m] = pq.rope_cos[(size_t)pm * half + 2 * j + pr]; ropef[18 * BM + pr * BM + m] = pq.rope_sin[(size_t)pm * half + 2 * j + pr];
            }
            if (KVR && cw == 2 && lane0 < 16) reinterpret_cast<int*>(lds + L_KVR(q))[lane0] = pq.kv_row[min(lane0, max(pq.n_rows - 1, 0))];      // (rows past n_rows: the last live row's slice -- their attention reads key 0 of it and is discarded)
        }
        int lane0 = lane0_; asm volatile("" : "+v"(lane0));
        const EngLayerTab* tab = reinterpret_cast<const EngLayerTab*>(lds + BL_TAB);
        float* gwt = reinterpret_cast<float*>(lds + BL_GW);
        const bool xchg = cw == NCONS - 1 && (p.flags & 128) != 0;
        const unsigned my = c->xcc_id;
        if (xchg && lane0 == 0) publish(p.XC + b, tag_base, __uint_as_float(my));
        const int l = cw + NCONS * (lane0 >> 4), r = lane0 & 15, lc = min(l, max(p.n_layers - 1, 0));
        const bool lok = r < OWN && l <= p.n_layers && (lane0 >> 4) < 3, lay = l < p.n_layers;
        const unsigned k = (unsigned)(OWN * b + min(r, OWN - 1));
        const gcf_p pa = as_g(lay ? tab[lc].attn_norm : p.final_norm);
        float wa = 0.f, wf = 0.f, wd = 0.f;
        if (lok) { wa = pa[k]; if (p.n_layers > 0) { wf = as_g(tab[lc].ffn_norm)[k]; wd = as_g(tab[lc].ada_mul)[k]; } }
        if (lok) { gwt[(l * 2) * 16 + r] = wa; if (lay) gwt[(l * 2 + 1) * 16 + r] = wf * wd; }
        if (cw == NCONS - 1) {
            unsigned ok = 0;
            if (xchg) {
                float v[1];
                const bool got = sweep<1>(p.XC, NCU * 8u, tag_base, [&](int) { return 8 * (lane0 & 31) + g; }, [&]() { return 0; }, false, v, c, p.err);
                ok = got && __all(__float_as_uint(v[0]) == my) ? 1u : 0u;
                if (!ok && lane0 == 0) __hip_atomic_store(p.err + 1, 9u | ((unsigned)b << 8) | (my << 16), RLX, AG);
            }
            lds_st(&c->xcd_ok, ok);
        }
        ENG_CFENCE();
        if (lane0 == 0) __hip_atomic_fetch_add(&c->gw_flag, 1u, RLX, WG);
    }
    wait_ge(&c->gw_flag, (unsigned)NCONS, c, p.err, ERR_STAGE);
    const int lane0 = lane0_;
#pragma unroll 1
    for (int l = 0; l < p.n_layers; l++) {
        if (NG == 1) {
            ph_qkv<1, KVR>(cw, lane0, l, 0);
            ph_attn<1, KVR>(cw, lane0, l, 0);
            ph_wo<1, KVR>(cw, lane0, l, 0);
            ph_w13<1, KVR>(cw, lane0, l, 0);
            ph_w2<1, KVR>(cw, lane0, l, 0);
        } else {      // group A's phase, then group B's: one computes while the other's edge resolves
#pragma unroll 1
            for (int q = 0; q < NG; q++) ph_qkv<NG, KVR>(cw, lane0, l, q);
#pragma unroll 1
            for (int q = 0; q < NG; q++) ph_attn<NG, KVR>(cw, lane0, l, q);
#pragma unroll 1
            for (int q = 0; q < NG; q++) ph_wo<NG, KVR>(cw, lane0, l, q);
#pragma unroll 1
            for (int q = 0; q < NG; q++) ph_w13<NG, KVR>(cw, lane0, l, q);
#pragma unroll 1
            for (int q = 0; q < NG; q++) ph_w2<NG, KVR>(cw, lane0, l, q);
        }
    }
}

template <int NG, bool KVR>
__global__ __launch_bounds__(NTHR, 1) void decode_engine_b16_kernel(const EngBArgs<NG> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    BCtl* c = reinterpret_cast<BCtl*>(lds + BL_CTL);
    const EngBParams& p = a.g[0];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < (int)(sizeof(BCtl) / 4)) reinterpret_cast<unsigned*>(c)[tid] = 0u;
    for (int i = tid; i < p.n_layers * (int)(sizeof(EngLayerTab) / 8); i += NTHR) reinterpret_cast<u64*>(lds + BL_TAB)[i] = reinterpret_cast<const u64*>(p.layers)[i];
    if (tid == 0) c->xcc_id = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));      // HW_REG_XCC_ID
    __syncthreads();
    Tl tl; tl.on = p.tl != nullptr && lane == 0; tl.buf = p.tl ? p.tl + (size_t)blockIdx.x * 32 : nullptr;
#ifndef B16_ROLES
#define B16_ROLES 7
#endif
    if (wave == 0) { tl(19); if (B16_ROLES & 1) b16_loader<NG>(p, c, (unsigned)(uintptr_t)(lds + BL_RING), lane, tl); }
    else if (wave == 1) {
        if (B16_ROLES & 2) b16_comm<NG>(c, lds, lane, tl);
        if (blockIdx.x == 0 && lane == 0) {
#pragma unroll
            for (int q = 0; q < NG; q++) { unsigned* sp = a.g[q].serial; const unsigned sv = *sp; asm volatile("" ::: "memory"); *sp = sv + 1u; }
        }
    } else if (B16_ROLES & 4) b16_consumer<NG, KVR>(c, lds, wave - 2, lane);
}

// 256 per-CU partial sums of squares -> 16 (fixed order), the count the launch-based lm_head GEMM's prologue takes
__global__ __launch_bounds__(256) void engb_ssq_fold_kernel(const float* __restrict__ in, float* __restrict__ out) {
    const int m = threadIdx.x & 15, q = threadIdx.x >> 4;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 16; u++) a += in[(16 * q + u) * BM + m];
    out[q * BM + m] = a;
}

}  // namespace

hipError_t launch_engb_ssq_fold(const float* in, float* out, hipStream_t s) { engb_ssq_fold_kernel<<<dim3(1), dim3(256), 0, s>>>(in, out); return hipGetLastError(); }

// state block: XH0 | XH1 | SS0 | SS1 | G | XO | PW | XA | P2 | F0 F1 FO FW FA F2 | XC | serial | err
static constexpr size_t BS_XH0 = 0, BS_XH1 = BS_XH0 + XH_BYTES, BS_SS0 = BS_XH1 + XH_BYTES, BS_SS1 = BS_SS0 + (size_t)NCU * BM * 4, BS_G = BS_SS1 + (size_t)NCU * BM * 4,
                        BS_XO = BS_G + (size_t)BM * G_ROW * 8, BS_PW = BS_XO + (size_t)ENH * XO_HEAD, BS_XA = BS_PW + (size_t)NPWB * ED * BM * 4, BS_P2 = BS_XA + (size_t)ENKV * XA_GROUP,
                        BS_F = BS_P2 + (size_t)NP2 * ED * BM * 4, BS_XC = BS_F + 21248 * 4, BS_SERIAL = BS_XC + (size_t)NCU * 8, BS_ERR = BS_SERIAL + 256, BS_TOTAL = BS_ERR + 256;
size_t engb_state_bytes() { return BS_TOTAL; }
size_t engb_wo_stream_bytes(int n_layers) { return (size_t)n_layers * WOB_LAYER_BYTES * NCU + 1024; }
void engb_state_carve(unsigned char* st, EngBParams* p) {
    p->XH0 = st + BS_XH0; p->XH1 = st + BS_XH1; p->SS0 = reinterpret_cast<float*>(st + BS_SS0); p->SS1 = reinterpret_cast<float*>(st + BS_SS1);
    p->G = reinterpret_cast<unsigned long long*>(st + BS_G); p->XO = st + BS_XO; p->PW = reinterpret_cast<float*>(st + BS_PW); p->XA = st + BS_XA; p->P2 = reinterpret_cast<float*>(st + BS_P2);
    unsigned* f = reinterpret_cast<unsigned*>(st + BS_F);
    p->FO = f; p->FA = f + 512; p->FW = f + 768; p->F2 = f + 6912; p->F0 = f + 13056; p->F1 = f + 17152;      // words: FO [32 heads][16 sequences], FA [256], FW / F2 [256 CUs][6 tiles] x 16 B, F0 / F1 [256] x <= 64 B
    p->XC = reinterpret_cast<unsigned long long*>(st + BS_XC); p->serial = reinterpret_cast<unsigned*>(st + BS_SERIAL); p->err = reinterpret_cast<unsigned*>(st + BS_ERR);
}
// zero the block, then mark every partial-plane slot empty (comm_reduce_poll)
hipError_t engb_state_init(unsigned char* st, hipStream_t s) {
    hipError_t e = hipMemsetAsync(st, 0, BS_TOTAL, s);
    if (e == hipSuccess) e = hipMemsetAsync(st + BS_PW, 0xFF, (size_t)NPWB * ED * BM * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(st + BS_P2, 0xFF, (size_t)NP2 * ED * BM * 4, s);
    return e;
}
int engb_lds_bytes() { return BL_TOTAL; }
int engb_lds_bytes2() { return BL_TOTAL2; }

// the three forms raise their dynamic-LDS limit once, all together (hipFuncSetAttribute is not something to do inside a stream capture: vox_api.cpp calls this from
// engb_prepare, before any graph holds a launch)
static DevOnce g_engb_attr_done;
hipError_t engb_prepare_kernels() {
    const int dev = vox_current_device();
    if (g_engb_attr_done.done(dev)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_b16_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_b16_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_b16_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) g_engb_attr_done.set(dev);
    return e;
}
template <int NG, bool KVR>
static hipError_t engb_launch(const EngBArgs<NG>& a, int lds_bytes, hipStream_t s) {
    hipError_t e = engb_prepare_kernels();
    if (e != hipSuccess) return e;
    decode_engine_b16_kernel<NG, KVR><<<dim3(NCU), dim3(NTHR), lds_bytes, s>>>(a);
    return hipGetLastError();
}
static bool engb_params_ok(const EngBParams& p) { return p.n_rows >= 1 && p.n_rows <= BM && p.n_layers >= 0 && p.n_layers <= MAX_LAYERS && p.stream_wo != nullptr; }
hipError_t launch_decode_engine_b16(const EngBParams& p, hipStream_t s) {
    if (!engb_params_ok(p)) return hipErrorInvalidValue;
    EngBArgs<1> a; a.g[0] = p;
    return p.kv_row ? engb_launch<1, true>(a, BL_TOTAL, s) : engb_launch<1, false>(a, BL_TOTAL, s);
}
// two groups through one launch: the same weights, layer table (cache slabs: slices are picked through kv_row), RoPE tables and flags; each group its own state block
hipError_t launch_decode_engine_b16x2(const EngBParams& ga, const EngBParams& gb, hipStream_t s) {
    if (!engb_params_ok(ga) || !engb_params_ok(gb) || !ga.kv_row || !gb.kv_row) return hipErrorInvalidValue;
    if (ga.stream != gb.stream || ga.stream_wo != gb.stream_wo || ga.layers != gb.layers || ga.n_layers != gb.n_layers || ga.flags != gb.flags || ga.serial == gb.serial ||
        ga.kv_seq_stride != gb.kv_seq_stride || ga.max_seq != gb.max_seq) return hipErrorInvalidValue;
    EngBArgs<2> a; a.g[0] = ga; a.g[1] = gb; a.g[1].err = ga.err; a.g[1].tl = ga.tl; a.g[1].tl_layer = ga.tl_layer;      // one error word per launch (group A's block)
    return engb_launch<2, true>(a, BL_TOTAL2, s);
}
// resident workgroups per CU the runtime grants this kernel (the engine needs its 256 workgroups co-resident: >= 1 on a 256-CU device)
hipError_t engb_occupancy(int* blocks_per_cu) {
    hipError_t e = engb_prepare_kernels();
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(decode_engine_b16_kernel<2, true>), NTHR, BL_TOTAL2);
}

}  // namespace vox
