// vox_engine.hip -- persistent decode-step engine for gfx950 (MI355X): one launch per decoded token.
//
// What it replaces: the 26 x 4 + 1 dependent launches of one single-stream decode step (reference: gguf/model.rs:938-960 -> forward_hidden_with_cache
// + lm_head, model.rs:566-691).  Why: a decode layer streams 65.5 MB of Q4 weights -- 10 us at HBM speed -- but every launch pays a kernel
// boundary, a ramp, and the round trip of its activation vector before the first weight byte is consumed.  Here the
// weight stream never stops: it does not depend on anything, so a loader wave per CU runs AHEAD of every dependency edge, and the edges
// themselves are 8-byte {value, tag} write-through granules swept by one wave per CU (MI355X_MICROARCH.md "Persistent kernels" price list).
//
// Geometry (fixed: the real Voxtral decoder -- D 3072, 32 query heads / 8 KV heads x 128, FFN 9216; other shapes keep the per-operator path):
//   grid = 256 workgroups (one per CU, all resident) x 896 threads = 14 waves (<= 128 VGPRs):
//     wave 0       LOADER   global_load_lds_dwordx4 ... nt: this CU's step records of q|k|v, wo, w1|w3, w2 of every layer, then the lm_head rows, as 13.5 / 20 KiB
//                           packets into a ring of six LDS slots; 1-3 packets in flight; waits only for free slots.  The wave is ISSUE-bound (a lone wave issues about
//                           one instruction per 8 cycles): four 1 KiB lines per M0 write, the packet's line loop is straight-line code.
//     wave 1       COMM     sweeps granules written by other CUs into LDS (q|k|v of the head, the XCD group's SwiGLU outputs, partial planes of the CU's 12 residual
//                           rows), reduces partial sums in a FIXED order (deterministic), publishes the CU's 12 rows (already multiplied by the next norm weight).
//     waves 2..13  CONSUMERS  layer operators as EXACT integer dot products on v_mfma_i32_16x16x64_i8: B = Q4 nibbles as int8, A = the activation in per-block fixed
//                           point split into four digits (rows 4 p + c = digit c of block p), K split over the 12 waves, cross-wave sums through LDS; step records
//                           are moved into registers while a wave waits for an edge; attention; lm_head (fp8-trick VALU path, one row per wave per packet).
//   CU b = (g = b % 8: KV head / XCD, j = b / 8): query head h = 4 g + j / 8, slice s = j % 8.  Per layer:
//     q|k|v   CU computes q rows [128 h + 16 s, +16), k rows [4 j, +4) and v rows [4 j, +4) of KV head g (K = 3072, RMSNorm scale applied at the sum)  -> granules G
//     attn    CU (h, s) gathers q_h, k_g, v_g (new row) and runs head h's single-query attention over the cache (redundantly per slice)
//     wo      CU (h, s): rows [384 s, +384) x columns of head h (K = 128)  -> 32 partial planes PW; owner of rows [12 b, +12) sums them + residual -> H1
//     w1|w3   CU (g, j): SwiGLU outputs [1152 g + 36 j, +36) (K = 3072)     -> granules A (read inside the XCD group only)
//     w2      CU (g, j): rows [96 j, +96) x the XCD group's K slice [1152 g, +1152) -> 8 partial planes P2; owner sums + residual -> H0 (next layer's input)
//   An all-gather (H0, H1) is 4 granules per lane for each consumer wave (its own K slice); the other edges are <= 1152 granules.
// Tags = (launch serial + 1) * 64 + layer + 1: unique per (launch, layer), so no buffer is ever re-initialised and a stale granule can never match.
// Every spin is bounded (20 ms): on timeout the workgroup sets *err, marks itself dead and runs to completion without waiting.
// History, measurements and everything that was tried: DESIGN.md section 3.0, profiles/r03_engine_experiments.md.
#include "vox_kernels.h"

#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstdlib>

namespace vox {
namespace {

#include "vox_engine_common.h"

// lm_head pass q (= 12 * packet + consumer wave) on CU b: the row; lane holds block `lane` and half (lane & 1) of block 64 + lane / 2
__global__ __launch_bounds__(64) void eng_pack_lm_kernel(Q4W w, unsigned char* __restrict__ stream, size_t op_off, int vocab) {
    const int q = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int row = lm_rows_per_cu(vocab) * b + q;
    unsigned char* dst = stream + (size_t)NCU * (op_off + (size_t)(q / NCONS) * PK_A) + (size_t)b * PK_A + (size_t)(q % NCONS) * PASS_A;
    const size_t src = (size_t)row * w.nb + lane;
    reinterpret_cast<uint4*>(dst)[lane] = w.qs[src];
    reinterpret_cast<uint16_t*>(dst + PA_S0)[lane] = w.sc[src];
    const size_t sh = (size_t)row * w.nb + 64 + (lane >> 1);
    const uint4 qh = w.qs[sh];
    reinterpret_cast<uint2*>(dst + PA_Q1)[lane] = (lane & 1) ? make_uint2(qh.z, qh.w) : make_uint2(qh.x, qh.y);      // bytes [8 half, +8): elements [8 half, +8) and 16 + the same
    if ((lane & 1) == 0) reinterpret_cast<uint16_t*>(dst + PA_S1)[lane >> 1] = w.sc[sh];
}
// layer operators: one workgroup per (record, CU).  Packet-major stream: packet k of all 256 CUs is contiguous ([k][cu][bytes]) -- at any moment the 256
// loaders read one contiguous ~5 MB window, spread over every HBM channel (CU-major streams 7.3 MB apart put all loaders on the same channels at the same time)
__global__ __launch_bounds__(64) void eng_pack_kernel(Q4W w, int op, unsigned char* __restrict__ stream, size_t op_off) {
    const int pk = blockIdx.x / (NCONS * 4), wv = (blockIdx.x / 4) % NCONS, s = blockIdx.x % 4, b = blockIdx.y, lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int steps = pk_steps(op, pk); const bool half = pk_half_tile(op, pk);
    if (s >= steps || (half && n >= 8)) return;
    int tile, T; rec_src(op, pk, wv, s, &tile, &T);
    const int row = tile_row(op, b, tile, n), blk0 = step_blk0(op, b, T);
    const size_t pk_bytes = (size_t)op_pk_bytes(op), rec_bytes = half ? REC_H : REC;
    unsigned char* dst = stream + (size_t)NCU * (op_off + (size_t)pk * pk_bytes) + (size_t)b * pk_bytes + (size_t)wv * steps * rec_bytes + (size_t)s * rec_bytes;
    const uint4 q = w.qs[(size_t)row * w.nb + blk0 + (g >> 1)];
    reinterpret_cast<uint2*>(dst)[half ? g * 8 + n : lane] = (g & 1) ? make_uint2(q.z, q.w) : make_uint2(q.x, q.y);
    if (g < 2) reinterpret_cast<uint16_t*>(dst + (half ? REC_H_SC : REC_SC))[(half ? 8 : 16) * g + n] = w.sc[(size_t)row * w.nb + blk0 + g];
}

// ------------------------------------------------------------------------------------------------
// LDS map
// ------------------------------------------------------------------------------------------------
struct EngCtl {
    unsigned ring_ready[8], ring_done[8];         // monotonic per slot: fills landed / passes consumed
    unsigned xs0_flag, xs1_flag, xa_flag, qkv_flag;   // layer + 1 of the staged content (monotonic)
    unsigned cbar, dead, gathering, gw_flag;
    unsigned xcd_ok, xcc_id, ag_flag, pub_cnt;    // pub_cnt: consumer waves that have issued an operator's publishes (12 per operator, 4 operators per layer)       // ag_flag: all-gather stages whose probe has resolved (the consumer waves then sweep the vector themselves)
    float rstd0, rstd1, pad1, pad2;
    float best_val[16]; int best_idx[16];
    float h_own[16], h1_own[16];
    unsigned ag_done, pad_q[3];                   // consumer waves through with their all-gather sweep (12 per stage)
    float rope_c[24], rope_s[24];                 // RoPE factors of the CU's 24 q|k|v rows (the same in every layer; wave 0 reads them back per layer instead of holding two VGPRs all step)
};
constexpr int L_RING = 0;
constexpr int L_XS = L_RING + NSLOT * SLOT_BYTES;       // 12 KB: the all-gathered (or XCD-gathered) input vector of the next operator, as FIXED-POINT DIGIT PLANES for the MFMA operators
                                                        //   ([block][digit 0..3][half][16 B]: 128 bytes per 32-element block -- the same 4 bytes per element as f32), or as swizzled f32 chunks for
                                                        //   the lm_head.  ONE buffer: it is re-staged only after every consumer wave has published results computed from the previous content
constexpr int L_U = L_XS + ED * 4;                      // time-shared: cross-wave partial sums of the operator in flight | the XCD group's 1152 SwiGLU outputs (f32, before conversion) | attention scratch
constexpr int L_XA = L_U;                               //   [1152] f32 staged by the COMM wave after w1|w3, converted to digit planes (in L_XS) by the consumer waves
constexpr int L_PART = L_U;                             //   [rows][12 or 6] f32 partial sums (<= 72 x 12)
constexpr int L_SC = L_U;                               //   [SC_MAX] scores; after the softmax: [128] attention output (f32) on its way to digits
constexpr int L_PO = L_SC + SC_MAX * 4;                 //   [12][128] partial attention outputs
constexpr int L_PL = L_PO + 12 * 128 * 4;               //   [16] partial softmax sums
constexpr int L_QKVN = L_PL + 64;                  // q_h[128] k_g[128] v_g[128] of this step (plain order)
constexpr int L_TMP = L_QKVN + 384 * 4;                 // [384] partial sums swept by the comm wave
constexpr int L_SSQ = L_TMP + 384 * 4;                  // [256] per-CU partial sums of squares of the vector being all-gathered
constexpr int L_BLK = L_SSQ + NCU * 4;                  // [96] per 32-element block of the staged vector: {-8 * sum of its fixed-point values, 2^-shift}
constexpr int L_TAB = L_BLK + 96 * 8;                  // [MAX_LAYERS] copy of the layer table: pointer reads never touch VMEM (a vector load behind a publish waits for the store)
constexpr int MAX_LAYERS = 32;
constexpr int L_GW = L_TAB + MAX_LAYERS * (int)sizeof(EngLayerTab);      // [MAX_LAYERS + 1][2][16] norm weight of this CU's 12 rows: [l][0] attn_norm (l = L: final norm * 512 -- the lm_head's fp8 trick), [l][1] ffn_norm * Ada
constexpr int L_CTL = L_GW + (MAX_LAYERS + 1) * 32 * 4;
constexpr int L_TOTAL = L_CTL + (int)sizeof(EngCtl);
static_assert(L_TOTAL <= 160 * 1024, "LDS budget");
static_assert(L_TAB % 16 == 0 && sizeof(EngLayerTab) == 40, "layer table");
static_assert(L_XS % 16 == 0 && L_XA % 16 == 0 && L_QKVN % 16 == 0 && L_PO % 16 == 0 && L_CTL % 16 == 0 && L_BLK % 8 == 0 && 72 * 12 * 4 <= SC_MAX * 4 && 1152 * 4 + 96 * 6 * 4 <= SC_MAX * 4 + 12 * 128 * 4, "aligned carve");

// ONE rolled loop over the step's packets (the kernel's code must stay small: the instruction cache is shared by two CUs and every layer walks
// through all three roles' code -- a 100 KB kernel ran 13 % slower than a 68 KB one with the same structure)
__device__ __forceinline__ void eng_loader(const EngParams& p, EngCtl* c, unsigned ring_lds, int lane, const Tl& tl) {
    Loader<EngCtl, NSLOT> ld; ld.c = c; ld.err = p.err; ld.ring_lds = ring_lds; ld.voff = (unsigned)lane * 16u; ld.thin = (p.flags & 1) != 0; ld.pace = (u64)p.pace_ticks;
    const bool fake = (p.flags & 2) != 0;      // diagnostic: every packet re-reads one packet (L2 hits, no HBM traffic; results wrong)
    const bool nodma = (p.flags & 32) != 0;    // diagnostic: no LDS-DMA at all inside the layers (results wrong)
    if (p.flags & 64) ld.pause_ticks = 300;
    if (p.flags & 1024) ld.depth = 2;
    if (p.flags & 2048) ld.depth = 1;
    const u64 base = (u64)p.stream;
    const unsigned n_layer_pk = (unsigned)p.n_layers * PK_LAYER, n_pk = n_layer_pk + (unsigned)lm_packets(p.vocab);
    unsigned l = 0, r = 0;                       // layer, packet within the layer
    u64 off = 0;                                 // byte offset of the next packet in this CU's stream (packets are stored in consumption order)
#pragma unroll 1
    for (unsigned pk = 0; pk < n_pk; pk++) {
        int bytes = PK_A;
        if (pk < n_layer_pk) {
            if (r < PK_LAYER_M) bytes = PK_M;
            if (r == 0 && (int)l == p.tl_layer) tl(16);
        } else { ld.pace = 0; ld.pause_ticks = 0; ld.depth = 3; }      // no edge left to protect: the lm_head streams at full depth
        const u64 src = fake ? base + (u64)blockIdx.x * PK_A : base + (u64)NCU * off + (u64)blockIdx.x * (u64)bytes;
        if (bytes == PK_M) ld.issue<PK_M>(src, lane, nodma); else ld.issue<PK_A>(src, lane, nodma);
        off += (u64)bytes;
        if (++r == PK_LAYER) { if ((int)l == p.tl_layer) tl(17); r = 0; l++; }
    }
    ld.flush();
    tl(18);
}


// All-gather of a staged activation vector: the owners publish their 12 rows ALREADY multiplied by the consumer's norm weight (* Ada scale) * 512,
// plus one partial sum of squares per CU, so the sweep is granules -> LDS with no other memory operand (the per-layer norm vectors take microseconds to
// arrive and would sit in front of the granule loads: VMEM returns in order).  The COMM wave only probes (one row of every 4th producer, one granule per
// lane); once the probe resolves, the 12 consumer waves sweep 4 (+ 1) granules per lane each -- one round trip, no 100-register sweep in one wave.
__device__ __forceinline__ void comm_probe_x(const EngParams& p, EngCtl* c, int lane, const u64* src, unsigned tag, const Tl& tl, bool T) {
    const srd_t sd = make_srd(src, ED * 8u);
    if (T) tl(20);
    u64 t0 = 0;
    if (p.flags & 512) {      // no probe: the owners publish within a fraction of a microsecond of each other, so wait out the store-to-visibility latency once
        const u64 tw = wall_clock64();      // and go straight for the full sweep (a miss costs one more round trip)
        while (wall_clock64() - tw < (u64)p.ag_delay_ticks) __builtin_amdgcn_s_sleep(1);
    } else
    for (;;) {
        const u64 gq = ld_gran(sd, 48u * (unsigned)lane);
        if (__all((unsigned)(gq >> 32) == tag)) break;
        if (sweep_bail(t0, tag, c, p.err)) break;
    }
    if (T) tl(21);
}
// the owner's side: rows [12 b, +12) of a residual stream -> granules of row * (next norm weight) * 512, the CU's partial sum of squares, raw rows kept in LDS
__device__ __forceinline__ void comm_publish_rows(const EngParams& p, int lane, float hraw, float gw, u64* dst, u64* ssq, unsigned tag, float* own) {
    const int b = blockIdx.x;
    float sq = lane < OWN ? hraw * hraw : 0.f;
    sq = row16_sum_e(sq);                                   // lanes 0..11 live in the first DPP row
    if (lane < OWN) { publish(dst + OWN * b + lane, tag, hraw * gw); own[lane] = hraw; }
    if (lane == 0) publish(ssq + b, tag, sq);
}

// flags 65536: the step's input is formed HERE instead of by a separate launch behind every step (argmax_embed_kernel: 4.7 us + a kernel boundary per token).  Every CU
// reduces the previous launch's 256 argmax partials itself (4 per lane, lowest vocabulary index wins ties: argmax_embed_kernel's order) and builds its own 12 rows
// of audio[pos] + embed(token) with embed_row's arithmetic ((nibble - 8) * d, then the audio row added: separate roundings).  Returns lane r's raw row (r < 12).
constexpr int ENGF_ARGMAX_IN = 65536;
__device__ __forceinline__ float comm_next_input(const EngParams& p, int lane, unsigned own_k) {
    float xv[4]; int xi[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { xv[u] = as_g(p.part_val)[lane + 64 * u]; xi[u] = ((const __attribute__((address_space(1))) int*)(uintptr_t)p.part_idx)[lane + 64 * u]; }
    const int cur = __builtin_amdgcn_readfirstlane(*p.pos_rw) + 1;
    const float au = as_g(p.audio)[(size_t)cur * ED + own_k];      // does not depend on the token: in flight with the partials
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < 4; u++) if (xv[u] > bv || (xv[u] == bv && xi[u] < bi)) { bv = xv[u]; bi = xi[u]; }
#define ENG_AMAX_STEP(CTRL) { const float ov = dppf<CTRL>(bv); const int oi = dppi<CTRL>(bi); if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
    ENG_AMAX_STEP(0xB1) ENG_AMAX_STEP(0x4E) ENG_AMAX_STEP(0x141) ENG_AMAX_STEP(0x140)      // every lane of a 16-lane row holds the row's best (the order is total: any tree gives the same result)
#undef ENG_AMAX_STEP
    float v0 = rlf(bv, 0); int token = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 1; r < 4; r++) { const float ov = rlf(bv, 16 * r); const int oi = __builtin_amdgcn_readlane(bi, 16 * r); if (ov > v0 || (ov == v0 && oi < token)) { v0 = ov; token = oi; } }
    if (blockIdx.x == 0 && lane == 0) p.tokens[cur] = token;
    token = (unsigned)token < (unsigned)p.vocab ? token : 0;      // the previous launch may have timed out before it wrote its partials: never read the embedding out of bounds
    const unsigned c = own_k >> 5, e = own_k & 31;
    const size_t blk = (size_t)token * (size_t)p.tok_nb + c;
    const unsigned byte = ((const __attribute__((address_space(1))) unsigned char*)(uintptr_t)p.tok_qs)[blk * 16 + (e & 15)];
    const float d = __half2float(__ushort_as_half(((const __attribute__((address_space(1))) unsigned short*)(uintptr_t)p.tok_sc)[blk]));
    const int nib = e < 16 ? (int)(byte & 15u) : (int)(byte >> 4);
    return __fadd_rn(au, __fmul_rn(__fsub_rn((float)nib, 8.0f), d));
}

// ONE rolled loop over the 2 L + 1 all-gathers of the step: stage 2 l = layer l's input (-> q|k|v), stage 2 l + 1 = its post-attention stream (-> w1|w3),
// stage 2 L = the final norm's input (-> lm_head); the small edges that follow each all-gather hang off the loop body.
__device__ __forceinline__ void eng_comm(const EngParams& p, EngCtl* c, unsigned char* lds, const int lane0, const Tl& tl) {
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3);
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
        const float g0 = ld_gf(make_srd(p.n_layers > 0 ? tab[0].attn_norm : p.final_norm, ED * 4u), own_k) * (p.n_layers > 0 ? 1.0f : 512.0f);
        const float hraw = (p.flags & ENGF_ARGMAX_IN) ? comm_next_input(p, lane0, own_k) : ld_gf(make_srd(p.h_in, ED * 4u), own_k);
        comm_publish_rows(p, lane0, hraw, g0, p.H0, p.SS0, tag_base, c->h_own);
    }
#pragma unroll 1
    for (int st = 0; st <= 2 * p.n_layers; st++) {
        int lane = lane0; asm volatile("" : "+v"(lane));      // opaque per stage: lane-derived addresses are recomputed, not carried around the loop in VGPRs
        const int l = st >> 1; const bool odd = st & 1, last = st == 2 * p.n_layers, T = l == p.tl_layer;
        const unsigned tag = tag_base + (unsigned)l + 1u;       // written during layer l
        if (odd) comm_probe_x(p, c, lane, p.H1, tag, tl, T);
        else comm_probe_x(p, c, lane, p.H0, tag - 1u, tl, false);
        lds_st(&c->gathering, 1u);      // "this CU is polling memory": the loader (flags 1 / 64) keeps its LDS-DMA traffic out of the CU's memory pipeline meanwhile
        lds_st(&c->ag_flag, (unsigned)st + 1u);
        if (T) tl(odd ? 11 : 8);
        if (p.flags & (1 | 64)) wait_ge(&c->ag_done, NCONS * ((unsigned)st + 1u), c, p.err, ERR_STAGE);
        lds_st(&c->gathering, 0u);
        if (last) break;
        if (!odd) {
            wait_ge(&c->gw_flag, (unsigned)NCONS, c, p.err, ERR_STAGE);
            const bool xloc_nt = (p.flags & 128) != 0 && (p.flags & 4096) != 0 && lds_ld(&c->xcd_ok) != 0;      // (measurement knob 4096: XCD-local edges polled with nt loads)
            {   // this step's q_h, k_g, v_g rows
                wait_ge(&c->pub_cnt, NCONS * (4u * (unsigned)l + 1u), c, p.err, ERR_STAGE);
                lds_st(&c->gathering, 1u);
                float v[6];
                sweep<6>(p.G, (EQD + 2 * EKD) * 8u, tag, [&](int u) { const int i = lane + 64 * u, seg = i >> 7, e = i & 127; return seg == 0 ? 128 * h + e : (seg == 1 ? EQD : EQD + EKD) + 128 * g + e; },
                         [&]() { return lane < 32 ? EQD + 128 * g + 4 * lane : EQD + EKD + 128 * g + 4 * (lane - 32); }, PROBE_SMALL, v, c, p.err, xloc_nt);      // probe: a k / v row of each of the group's 32 CUs
                lds_st(&c->gathering, 0u);
#pragma unroll
                for (int u = 0; u < 6; u++) qkvn[lane + 64 * u] = v[u];
                ENG_CFENCE(); lds_st(&c->qkv_flag, (unsigned)l + 1u);
            }
            if (T) tl(9);
            {   // wo: 32 partial planes of this CU's 12 rows -> residual stream after attention, published as the w1|w3 input (gw = ffn_norm * Ada * 512)
                wait_ge(&c->pub_cnt, NCONS * (4u * (unsigned)l + 2u), c, p.err, ERR_STAGE);
                lds_st(&c->gathering, 1u);
                float v[6];
                sweep<6>(p.PW, NPW * ED * 8u, tag, [&](int u) { const int i = lane + 64 * u, hh = i / OWN, r = i - hh * OWN; return hh * ED + OWN * b + r; }, [&]() { return (lane & 31) * ED + OWN * b; }, PROBE_SMALL, v, c, p.err);
#pragma unroll
                for (int u = 0; u < 6; u++) tmp[lane + 64 * u] = v[u];
                lds_st(&c->gathering, 0u);
                ENG_CFENCE();
                float a = 0.f;
                const int r = min(lane, OWN - 1);
#pragma unroll 4
                for (int hh = 0; hh < NPW; hh++) a += tmp[hh * OWN + r];      // fixed order
                wait_ge(&c->gw_flag, (unsigned)NCONS, c, p.err, ERR_STAGE);
                if (!((p.flags & 16384) && blockIdx.x == 7 && l == 1))      // (flag 16384 = FAULT INJECTION for the test of the bounded waits: workgroup 7 loses a publish)
                    comm_publish_rows(p, lane, c->h_own[r] + a, gwt[(l * 2 + 1) * 16 + r], p.H1, p.SS1, tag, c->h1_own);
            }
            if (T) tl(10);
        } else {
            const bool xloc_nt = (p.flags & 128) != 0 && (p.flags & 4096) != 0 && lds_ld(&c->xcd_ok) != 0;
            {   // the XCD group's 1152 SwiGLU outputs -> w2 input
                wait_ge(&c->pub_cnt, NCONS * (4u * (unsigned)l + 3u), c, p.err, ERR_STAGE);
                lds_st(&c->gathering, 1u);
                float v[18];
                sweep<18>(p.A, EF * 8u, tag, [&](int u) { return 1152 * g + lane + 64 * u; }, [&]() { return 1152 * g + 36 * (lane & 31) + 35; }, PROBE_SMALL, v, c, p.err, xloc_nt);      // probe: the last output of each CU of the group
#pragma unroll
                for (int u = 0; u < 18; u++) xa[lane + 64 * u] = v[u];      // plain f32: the consumer waves turn it into digit planes
                ENG_CFENCE(); lds_st(&c->xa_flag, (unsigned)l + 1u);
                lds_st(&c->gathering, 0u);
            }
            if (T) tl(12);
            {   // w2: 8 partial planes (one per XCD group) of this CU's 12 rows -> the layer's output, published as the next layer's q|k|v input (gw = next attn_norm)
                wait_ge(&c->pub_cnt, NCONS * (4u * (unsigned)l + 4u), c, p.err, ERR_STAGE);
                lds_st(&c->gathering, 1u);
                float v[2];
                sweep<2>(p.P2, NP2 * ED * 8u, tag, [&](int u) { const int i = min(lane + 64 * u, NP2 * OWN - 1), pp = i / OWN, r = i - pp * OWN; return pp * ED + OWN * b + r; }, [&]() { return min(lane, NP2 - 1) * ED + OWN * b; }, PROBE_SMALL, v, c, p.err);
#pragma unroll
                for (int u = 0; u < 2; u++) if (lane + 64 * u < NP2 * OWN) tmp[lane + 64 * u] = v[u];
                lds_st(&c->gathering, 0u);
                ENG_CFENCE();
                float a = 0.f;
                const int r = min(lane, OWN - 1);
#pragma unroll
                for (int pp = 0; pp < NP2; pp++) a += tmp[pp * OWN + r];      // fixed order
                comm_publish_rows(p, lane, c->h1_own[r] + a, gwt[((l + 1) * 2) * 16 + r], p.H0, p.SS0, tag, c->h_own);
            }
            if (T) tl(13);
        }
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

// the 16 weights of half a block (bytes [8 hb, +8) of its nibbles: elements [8 hb, +8) and 16 + the same); x[4 d .. 4 d + 3] = the activations of dword d
__device__ __forceinline__ float half_dot(const uint2 q, const f2* __restrict__ x, float init) {
    f2 a0 = {init, 0.f}, a1 = {0.f, 0.f};
    const unsigned w[2] = {q.x, q.y};
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const unsigned lo = w[d] & 0x0F0F0F0Fu, hi = (w[d] >> 4) & 0x0F0F0F0Fu;
        a0 = __builtin_elementwise_fma(cvt2(lo, false), x[4 * d], a0);
        a1 = __builtin_elementwise_fma(cvt2(lo, true), x[4 * d + 1], a1);
        a0 = __builtin_elementwise_fma(cvt2(hi, false), x[4 * d + 2], a0);
        a1 = __builtin_elementwise_fma(cvt2(hi, true), x[4 * d + 3], a1);
    }
    return (a0.x + a0.y) + (a1.x + a1.y);
}

// activation registers of one lane: a whole 32-element chunk and half of a second one (A-type passes: 96 blocks over 64 lanes), or one chunk (wo)
struct XA {
    f2 x[16], xh[8]; float m8, m8h;
    __device__ __forceinline__ void load_chunk(const float* xs, int chunk) {
        asm volatile("" : "+v"(chunk));      // opaque per call: keeps the swizzled piece addresses out of the loop invariants
        const float4* x4 = reinterpret_cast<const float4*>(xs);
        float s = 0.f;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) {
            const float4 v = x4[sw_piece(chunk, jj)];
            x[2 * jj] = f2{v.x, v.y}; x[2 * jj + 1] = f2{v.z, v.w};
            s += (v.x + v.y) + (v.z + v.w);
        }
        m8 = s * (-1.0f / 64.0f);            // -8 * sum x = -(8 / 512) * sum x'
    }
    __device__ __forceinline__ void load(const float* xs, int chunk, int chunk_h, int hb) {
        load_chunk(xs, chunk);
        asm volatile("" : "+v"(chunk_h));
        const float4* x4 = reinterpret_cast<const float4*>(xs);
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                const float4 v = x4[sw_piece(chunk_h, 4 * hh + 2 * hb + d)];
                xh[4 * d + 2 * hh] = f2{v.x, v.y}; xh[4 * d + 2 * hh + 1] = f2{v.z, v.w};
                s += (v.x + v.y) + (v.z + v.w);
            }
        m8h = s * (-1.0f / 64.0f);
    }
};
struct PassA { uint4 q0; uint2 q1; float s0, s1; };

// ---- MFMA side -----------------------------------------------------------------------------------------------------------------------------
// The activation vector lives in LDS as fixed-point digit planes: per 32-element block a power-of-two scale 2^shift (block maximum < 2^26 after scaling),
// x_int = rint(x * 2^shift) = d0 + 128 d1 + 128^2 d2 + 128^3 d3 (d0..d2 in 0..127, d3 signed), and per block {-8 * sum(x_int), 2^-shift}.
// MFMA operand A, row m = 4 p + c: digit c of the step's block p, zero in the other block's columns -> D[4 p + c][n] = sum over block p of digit_c(x_k) * q[n][k]
// EXACTLY (int32).  Lane group g < 2 then holds the four digit sums of block g for tile row n: Horner in f32, the -8 offset, the two scales.
typedef int i4v __attribute__((ext_vector_type(4)));
struct StepCtx { i4v A; float m8sx, sxinv; };
__device__ __forceinline__ StepCtx load_ctx(const unsigned char* planes, const float2* binfo, int blk0, int lane) {
    const int n = lane & 15, g = lane >> 4;
    StepCtx cx; cx.A = i4v{0, 0, 0, 0};
    if (n < 8 && (g >> 1) == (n >> 2)) cx.A = *reinterpret_cast<const i4v*>(planes + (((blk0 + (n >> 2)) * 4 + (n & 3)) * 2 + (g & 1)) * 16);
    const float2 bi = binfo[blk0 + (g & 1)];      // lane groups 0 / 1 finish blocks 0 / 1 (groups 2, 3 compute along on the same values; their results are dropped)
    cx.m8sx = bi.x; cx.sxinv = bi.y;
    return cx;
}
// one step record as this lane holds it: 8 bytes of nibbles + the f16 scale bits of its block.  Records do not depend on the step's activations, so a wave
// moves them from the ring into REGISTERS while it waits for an edge (and frees the ring slots for the loader: the register file is the bigger buffer)
struct RecR { i4v B; float sc; };      // the MFMA B operand (nibbles unpacked to int8 -- done while the wave waits, off the critical path) and the block scale as f32
__device__ __forceinline__ RecR rec_read(const unsigned char* rec, bool half, int lane) {
    const int n = lane & 15, g = lane >> 4;
    uint2 q = make_uint2(0u, 0u);
    if (!half) q = reinterpret_cast<const uint2*>(rec)[lane];
    else if (n < 8) q = reinterpret_cast<const uint2*>(rec)[g * 8 + n];
    const unsigned short sb = reinterpret_cast<const unsigned short*>(rec + (half ? REC_H_SC : REC_SC))[(half ? 8 : 16) * (g & 1) + (half ? (n & 7) : n)];
    RecR r;
    r.B[0] = (int)(q.x & 0x0F0F0F0Fu); r.B[1] = (int)(q.y & 0x0F0F0F0Fu); r.B[2] = (int)((q.x >> 4) & 0x0F0F0F0Fu); r.B[3] = (int)((q.y >> 4) & 0x0F0F0F0Fu);
    r.sc = __half2float(__ushort_as_half(sb));
    asm volatile("" : "+v"(r.B[0]), "+v"(r.B[1]), "+v"(r.B[2]), "+v"(r.B[3]), "+v"(r.sc));      // pin the unpack HERE (hipcc sinks it to the MFMA otherwise: 7 VALU back on the critical path)
    return r;
}
// one MFMA step of a 16-row (or 8-row) tile: acc += (block scale) * (x block scale) * sum_k x_int[k] * (q[k] - 8) for this lane's block of tile row n
__device__ __forceinline__ float mstep(const RecR& r, const StepCtx& cx, float acc) {
    const i4v D = __builtin_amdgcn_mfma_i32_16x16x64_i8(cx.A, r.B, i4v{0, 0, 0, 0}, 0, 0, 0);
    // digits -> value: (D0 + 128 D1) and (D2 + 128 D3) exactly in int32 (v_lshl_add_u32; |D| <= 32 * 15 * 128), then one f32 FMA (one rounding, as before)
    const int lo = (int)((unsigned)D[0] + ((unsigned)D[1] << 7)), hi = (int)((unsigned)D[2] + ((unsigned)D[3] << 7));
    float t = fmaf((float)hi, 16384.0f, (float)lo);
    t += cx.m8sx;
    return fmaf(t, r.sc * cx.sxinv, acc);
}
__device__ __forceinline__ float mstep(const unsigned char* rec, bool half, int lane, const StepCtx& cx, float acc) { return mstep(rec_read(rec, half, lane), cx, acc); }
__device__ __forceinline__ float g01_sum(float v, int lane) { return v + __shfl(v, (lane + 16) & 63); }      // block 0 + block 1 partial of tile row n (valid in lanes 0..15)
// Four consecutive elements of a 32-element block (8 consecutive lanes = one block, this lane: elements [4 (lane & 7), +4)) -> digit planes + block info
__device__ __forceinline__ void to_digits(const float (&v)[4], int lane, unsigned char* plane_blk, float2* binfo_blk) {
    float mx = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    mx = fmaxf(mx, dppf<0xB1>(mx)); mx = fmaxf(mx, dppf<0x4E>(mx)); mx = fmaxf(mx, dppf<0x141>(mx));      // over the block's 8 lanes
    int e = __builtin_amdgcn_frexp_expf(mx);          // mx < 2^e (0 for mx == 0)
    e = max(e, -100);
    const float f = ldexpf(1.0f, 26 - e);
    int xi[4];
#pragma unroll
    for (int i = 0; i < 4; i++) xi[i] = (int)rintf(v[i] * f);
    int s = (xi[0] + xi[1]) + (xi[2] + xi[3]);
    s += dppi<0xB1>(s); s += dppi<0x4E>(s); s += dppi<0x141>(s);      // exact: |x_int| <= 2^26, 32 of them
    if ((lane & 7) == 0) *binfo_blk = make_float2(-8.0f * (float)s, ldexpf(1.0f, e - 26));
    const int e0 = 4 * (lane & 7), h = (e0 >> 3) & 1, off = (e0 & 4) + 8 * (e0 >> 4);
#pragma unroll
    for (int cdig = 0; cdig < 4; cdig++) {
        unsigned d = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int dv = cdig < 3 ? (xi[i] >> (7 * cdig)) & 127 : xi[i] >> 21; d |= (unsigned)(dv & 0xFF) << (8 * i); }
        *reinterpret_cast<unsigned*>(plane_blk + (cdig * 2 + h) * 16 + off) = d;
    }
}

struct Cons {
    const EngParams& p; EngCtl* c; unsigned char* lds; int cw, lane; unsigned P; unsigned cbar_n;
    __device__ __forceinline__ Cons(const EngParams& p_, EngCtl* c_, unsigned char* lds_, int cw_, int lane_) : p(p_), c(c_), lds(lds_), cw(cw_), lane(lane_), P(0), cbar_n(0) {}
    __device__ __forceinline__ void cbarrier() {
        cbar_n += NCONS;
        ENG_CFENCE();
        if (lane == 0) __hip_atomic_fetch_add(&c->cbar, 1u, RLX, WG);
        wait_ge(&c->cbar, cbar_n, c, p.err, ERR_CBAR);
    }
    // this wave's publishes of the current operator are issued: the COMM wave polls memory for an edge only once the CU's own 12 waves are through (every CU
    // runs the same schedule, so nothing can be complete much earlier; polling during the operator took VALU slots and fabric bandwidth from it)
    __device__ __forceinline__ void published(const Tl& tl, int evt) {      // evt >= 0: timeline stamp by the LAST of the 12 waves
        ENG_CFENCE();
        if (lane == 0) { const unsigned old = __hip_atomic_fetch_add(&c->pub_cnt, 1u, RLX, WG); if (evt >= 0 && tl.buf && old % NCONS == NCONS - 1) tl.buf[evt] = wall_clock64(); }
    }
    // this wave's share (share_bytes per wave) of packet pk, once it has landed
    __device__ __forceinline__ const unsigned char* slot_wait(unsigned pk, int share_bytes, int& slot) {
        slot = (int)(pk % NSLOT); const unsigned k = pk / NSLOT;
        wait_ge(&c->ring_ready[slot], k + 1u, c, p.err, ERR_RING);
        return lds + L_RING + slot * SLOT_BYTES + cw * share_bytes;
    }
    __device__ __forceinline__ void slot_release(int slot) {
        ENG_CFENCE();      // the LDS pipeline executes a wave's instructions in order: the reads above have been served when this add is
        if (lane == 0) __hip_atomic_fetch_add(&c->ring_done[slot], 1u, RLX, WG);
    }
    // lm_head: fetch this wave's pass of packet pk into registers and release the slot (the last packet has fewer passes than waves: the surplus waves read
    // whatever the slot holds and drop the result -- one straight-line body, no partially defined registers around the loop)
    __device__ __forceinline__ void fetch(unsigned pk, PassA& P_) {
        int slot; const unsigned char* base = slot_wait(pk, PASS_A, slot);
        P_.q0 = reinterpret_cast<const uint4*>(base)[lane];
        P_.q1 = reinterpret_cast<const uint2*>(base + PA_Q1)[lane];
        P_.s0 = __half2float(__ushort_as_half(reinterpret_cast<const unsigned short*>(base + PA_S0)[lane]));
        P_.s1 = __half2float(__ushort_as_half(reinterpret_cast<const unsigned short*>(base + PA_S1)[lane >> 1]));
        slot_release(slot);
    }
    static __device__ __forceinline__ float dot(const PassA& P_, const XA& xr) { return fmaf(P_.s1, half_dot(P_.q1, xr.xh, xr.m8h), P_.s0 * block_dot(P_.q0, xr.x, xr.m8)); }

    // This wave's share of an all-gather (after the COMM wave's probe has resolved): 4 consecutive granules per lane (elements [4 q, +4), q = 64 * region + lane; the
    // region is rotated by CU so that the 256 CUs do not walk the 24 KB in the same order), waves 0..3 also the 256 per-CU partial sums of squares.
    // DIGITS: the vector goes to LDS as digit planes (MFMA operators); otherwise as swizzled f32 chunks (lm_head).  Returns the RMSNorm scale (same fixed order everywhere).
    template <bool DIGITS>
    __device__ __forceinline__ float all_gather(const u64* src, const u64* ssq, unsigned tag, unsigned stage, unsigned char* xs, float* ssl, float2* binfo, const Tl& tl, bool T) {
        wait_ge(&c->ag_flag, stage + 1u, c, p.err, ERR_STAGE);
        const srd_t sd = make_srd(src, ED * 8u), qd = make_srd(ssq, NCU * 8u);
        int ln = lane; asm volatile("" : "+v"(ln));
        if (DIGITS) {
            // MFMA operators split K over the waves: wave cw multiplies columns [256 cw, +256) only -- it gathers exactly those 256 rows (4 per lane), turns them into
            // ITS 8 blocks of digit planes and goes on; no workgroup barrier, nobody waits for the slowest wave's granules.  Waves 0..3 also stage the 256 partial
            // sums of squares: the RMSNorm scale is applied where the K-slices are summed, behind the operator's own barrier.
            const unsigned q = 64u * (unsigned)cw + (unsigned)ln;
            u32x4 ra, rb; u64 rq = 0, t0 = 0;
            for (;;) {
                ra = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)(q * 32u), 0, 16));
                rb = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)(q * 32u + 16u), 0, 16));
                if (cw < 4) rq = ld_gran(qd, 64u * (unsigned)cw + (unsigned)ln);
                bool ok = ra.y == tag && ra.w == tag && rb.y == tag && rb.w == tag;
                if (cw < 4) ok &= (unsigned)(rq >> 32) == tag;
                if (__all(ok)) break;
                if (sweep_bail(t0, tag, c, p.err)) break;
            }
            if (T) tl(22);
            if (ln == 0) __hip_atomic_fetch_add(&c->ag_done, 1u, RLX, WG);
            const float v[4] = {__uint_as_float(ra.x), __uint_as_float(ra.z), __uint_as_float(rb.x), __uint_as_float(rb.z)};
            to_digits(v, ln, xs + (q >> 3) * 128u, binfo + (q >> 3));
            if (cw < 4) ssl[64 * cw + ln] = __uint_as_float((unsigned)rq);      // read back (rstd_staged) by the waves that finish the operator, behind its cross-wave barrier
            return 0.f;
        }
        // lm_head (VALU path): every wave needs the whole vector -- 4 consecutive granules per lane of a region rotated by CU, waves 0..3 the partial sums of squares
        const unsigned q = 64u * (((unsigned)cw + blockIdx.x) % NCONS) + (unsigned)ln;
        u32x4 ra, rb; u64 rq = 0, t0 = 0;
        for (;;) {
            ra = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)(q * 32u), 0, 16));
            rb = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sd, (int)(q * 32u + 16u), 0, 16));
            if (cw < 4) rq = ld_gran(qd, 64u * (unsigned)cw + (unsigned)ln);
            bool ok = ra.y == tag && ra.w == tag && rb.y == tag && rb.w == tag;
            if (cw < 4) ok &= (unsigned)(rq >> 32) == tag;
            if (__all(ok)) break;
            if (sweep_bail(t0, tag, c, p.err)) break;
        }
        if (ln == 0) __hip_atomic_fetch_add(&c->ag_done, 1u, RLX, WG);
        const float v[4] = {__uint_as_float(ra.x), __uint_as_float(ra.z), __uint_as_float(rb.x), __uint_as_float(rb.z)};
#pragma unroll
        for (int i = 0; i < 4; i++) reinterpret_cast<float*>(xs)[sw_dword((int)(4u * q) + i)] = v[i];
        if (cw < 4) ssl[64 * cw + ln] = __uint_as_float((unsigned)rq);
        cbarrier();
        return rstd_staged(ssl);
    }
    __device__ __forceinline__ float rstd_staged(const float* ssl) {
        const float ss = wave_sum_e((ssl[lane] + ssl[lane + 64]) + (ssl[lane + 128] + ssl[lane + 192]));      // fixed order: bit-identical on every wave of every CU
        return 1.0f / sqrtf(ss / (float)ED + p.eps);
    }
};

// ONE rolled loop over the step's 3 L + 1 items: per layer [q|k|v -> attention -> wo], w1|w3, w2; then the lm_head.
__device__ __forceinline__ void eng_consumer(const EngParams& p, EngCtl* c, unsigned char* lds, int cw, const int lane0, const Tl& tl) {
    Cons cs(p, c, lds, cw, lane0);
    const int b = blockIdx.x, g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7;
    unsigned char* xs = lds + L_XS; float* ssl = reinterpret_cast<float*>(lds + L_SSQ);
    float2* binfo = reinterpret_cast<float2*>(lds + L_BLK);
    const float* xa = reinterpret_cast<const float*>(lds + L_XA);
    float* part = reinterpret_cast<float*>(lds + L_PART);
    float* part2 = reinterpret_cast<float*>(lds + L_XA + 1152 * 4);      // w2's partial sums: behind the f32 SwiGLU outputs, which slower waves may still be converting
    const float* qkvn = reinterpret_cast<const float*>(lds + L_QKVN);
    float* sc = reinterpret_cast<float*>(lds + L_SC); float2* po = reinterpret_cast<float2*>(lds + L_PO); float* pl = reinterpret_cast<float*>(lds + L_PL);
    const unsigned tag_base = (*p.serial + 1u) * 64u;
    const int pos = *p.pos_ptr + p.pos_off + ((p.flags & ENGF_ARGMAX_IN) ? 1 : 0);      // (65536: this launch advances the position itself)
    const int j_lo = p.window >= 0 ? max(0, pos - p.window) : 0, n_old = pos - j_lo, last_old = max(n_old - 1, 0);
    // q|k|v epilogue (wave 0, lane r < 24 = the CU's row r: 16 of q, 4 of k, 4 of v): the RoPE factor of the row's pair -- the same in every layer
    const int half = EHD / 2;
    if (cw == 0 && lane0 < 24) {
        float rc = 1.0f, rs = 0.0f;
        if (lane0 < 20) { const int pr = lane0 < 16 ? 8 * s + (lane0 >> 1) : 2 * j + ((lane0 - 16) >> 1); rc = p.rope_cos[(size_t)pos * half + pr]; rs = p.rope_sin[(size_t)pos * half + pr]; }
        c->rope_c[lane0] = rc; c->rope_s[lane0] = rs;
    }
    const float scale = 1.0f / sqrtf((float)EHD);
    const int n_items = 3 * p.n_layers, npass_lm = lm_passes(p.vocab), row0_lm = lm_rows_per_cu(p.vocab) * b;
    {   // norm weights (* Ada scale) of this CU's 12 rows for every layer -> LDS, once per launch.  Wave cw fills the layers l = cw, cw + 12, cw + 24 (lane = 16 * slot + row):
        // three loads per lane, all in flight at once -- ONE round trip, in parallel on the twelve waves and under the COMM wave's first publish.  (One wave walking
        // all 27 layers in a loop took ~6 us, and every consumer waited for it before it even looked at the ring: "x staged" of layer 0 came at 10.6 us.)
        const EngLayerTab* tab = reinterpret_cast<const EngLayerTab*>(lds + L_TAB);
        float* gwt = reinterpret_cast<float*>(lds + L_GW);
        const bool xchg = cw == NCONS - 1 && (p.flags & 128) != 0;
        const unsigned my = c->xcc_id;
        // XCD-local edges (q|k|v -> attention, SwiGLU -> w2) go through the shared L2 only if the 32 workgroups of group g really sit on ONE XCD: workgroup b is
        // observed on XCD (b + rotation) % 8 (the dispatcher's round-robin carries over from the previous launch), so the ids are EXCHANGED and compared
        if (xchg && lane0 == 0) publish(p.XC + b, tag_base, __uint_as_float(my));      // first: the store is on its way while the table loads are
        const int l = cw + NCONS * (lane0 >> 4), r = lane0 & 15, lc = min(l, max(p.n_layers - 1, 0));
        const bool lok = r < OWN && l <= p.n_layers && (lane0 >> 4) < 3, lay = l < p.n_layers;
        const unsigned k = (unsigned)(OWN * b + min(r, OWN - 1));
        const gcf_p pa = as_g(lay ? tab[lc].attn_norm : p.final_norm);
        float wa = 0.f, wf = 0.f, wd = 0.f;
        if (lok) { wa = pa[k]; if (p.n_layers > 0) { wf = as_g(tab[lc].ffn_norm)[k]; wd = as_g(tab[lc].ada_mul)[k]; } }
        if (lok) {
            gwt[(l * 2) * 16 + r] = lay ? wa : wa * 512.0f;      // l = L: the final norm * 512 -- the lm_head's VALU path reads nibble bytes as e4m3 (q / 512)
            if (lay) gwt[(l * 2 + 1) * 16 + r] = wf * wd;
        }
        if (cw == NCONS - 1) {
            unsigned ok = 0;
            if (xchg) {
                float v[1];
                const bool got = sweep<1>(p.XC, NCU * 8u, tag_base, [&](int) { return 8 * (lane0 & 31) + g; }, [&]() { return 0; }, false, v, c, p.err);
                ok = got && __all(__float_as_uint(v[0]) == my) ? 1u : 0u;
                if (!ok && lane0 == 0) __hip_atomic_store(p.err + 1, 9u | ((unsigned)b << 8) | (my << 16), RLX, AG);      // informational (err[1]): the fast edges are off for this launch
            }
            lds_st(&c->xcd_ok, ok);
        }
        ENG_CFENCE();
        if (lane0 == 0) __hip_atomic_fetch_add(&c->gw_flag, 1u, RLX, WG);      // NCONS arrivals = table complete and xcd_ok decided (the exchanging wave arrives after its sweep)
    }
    float best = -INFINITY; int best_i = 0x7fffffff;
    bool xloc = false;      // this workgroup runs on XCD blockIdx % 8, like (by the same check) the group's other 31: decided by the exchange above, read behind layer 0's q|k|v steps

#pragma unroll 1
    for (int it = 0; it <= n_items; it++) {
        int lane = lane0; asm volatile("" : "+v"(lane));      // opaque per item: lane-derived addresses are recomputed, not carried around the loop in VGPRs
        const int l = it / 3, op = it == n_items ? (int)EOP_LM : (it - 3 * l == 0 ? (int)EOP_QKV : it - 3 * l == 1 ? (int)EOP_W13 : (int)EOP_W2);
        const bool T = l == p.tl_layer && cw == 0 && op != EOP_LM;
        const int tlev = l == p.tl_layer ? 1 : 0;
        const unsigned tag = tag_base + (unsigned)l + 1u;
        const EngLayerTab* L = reinterpret_cast<const EngLayerTab*>(lds + L_TAB) + (op == EOP_LM ? 0 : l);
        if (op == EOP_QKV) {
            // ================= q|k|v  ->  attention of head h  ->  wo =================
            const gf_p kc = as_g(L->kc) + (size_t)g * p.max_seq * EHD, vc = as_g(L->vc) + (size_t)g * p.max_seq * EHD;
            const int t12 = cw * 64 + lane;
            const int part_ = t12 & 7, ks = t12 >> 3;           // scores: 8 lanes per key, 96 keys per pass
            // this wave's 8 step records (K-steps 4 cw .. 4 cw + 3 of the 16-row q tile and of the 8-row k|v tile) -> registers, BEFORE the input vector exists
            RecR rq[8];
            {
                int s0, s1, s2;
                const unsigned char* b0 = cs.slot_wait(cs.P, 2 * REC, s0);
                rq[0] = rec_read(b0, false, lane); rq[1] = rec_read(b0 + REC, false, lane);
                cs.slot_release(s0);
                const unsigned char* b1 = cs.slot_wait(cs.P + 1, 2 * REC, s1);
                rq[2] = rec_read(b1, false, lane); rq[3] = rec_read(b1 + REC, false, lane);
                cs.slot_release(s1);
                const unsigned char* b2 = cs.slot_wait(cs.P + 2, 4 * REC_H, s2);
#pragma unroll
                for (int i = 0; i < 4; i++) rq[4 + i] = rec_read(b2 + i * REC_H, true, lane);
                cs.slot_release(s2);
                cs.P += QKV_PK;
            }
            cs.all_gather<true>(p.H0, p.SS0, tag - 1u, 2u * (unsigned)l, xs, ssl, binfo, tl, false);
            if (T) tl(0);
            // the old K rows do not depend on this step: requested BEFORE the q|k|v steps, so they are home before the q|k|v edge is polled
            constexpr int NKP = 2;      // 192 keys in registers (32 VGPRs): 96 requested here, 96 behind the q|k|v steps; later keys take the loop below
            float4 kpre[NKP][4];
            auto kload = [&](int u) {      // only keys that exist
                const unsigned ko = (unsigned)(j_lo + min(ks + 96 * u, last_old)) * EHD + part_ * 16;      // 32-bit lane offset + uniform base: one VGPR per address
#pragma unroll
                for (int e = 0; e < 4; e++) kpre[u][e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ks + 96 * u < n_old) {
#pragma unroll
                    for (int e = 0; e < 4; e++) kpre[u][e] = ldg4(kc + (ko + 4 * e));
                }
            };
            kload(0);
            {
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const StepCtx cx = load_ctx(xs, binfo, 2 * (4 * cw + i), lane);
                    a0 = mstep(rq[i], cx, a0); a1 = mstep(rq[4 + i], cx, a1);
                }
                a0 = g01_sum(a0, lane); a1 = g01_sum(a1, lane);
                if (lane < 16) part[lane * NCONS + cw] = a0;
                if (lane < 8) part[(16 + lane) * NCONS + cw] = a1;
            }
            cs.cbarrier();
            if (it == 0) { wait_ge(&c->gw_flag, (unsigned)NCONS, c, p.err, ERR_STAGE); xloc = (p.flags & 128) != 0 && lds_ld(&c->xcd_ok) != 0; }      // (long complete by now)
            if (cw == 0) {      // rows 0..15: q, 16..19: k, 20..23: v -- sum the 12 K-slices (fixed order), RMSNorm scale, RoPE on (even, odd) row pairs, publish
                const int r = min(lane, 23);
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NCONS; w++) a += part[r * NCONS + w];
                a *= cs.rstd_staged(ssl);
                const float rc = c->rope_c[r], rs = c->rope_s[r];
                const float o = dppf<0xB1>(a);                                   // the pair's other row
                const float y = (lane & 1) ? fmaf(o, rs, a * rc) : fmaf(-o, rs, a * rc);      // interleaved-pair RoPE (rope.rs:99-141); identity (rc 1, rs 0) for v
                if (lane < 24) {
                    const int n = lane < 16 ? 128 * h + 16 * s + lane : lane < 20 ? EQD + 128 * g + 4 * j + (lane - 16) : EQD + EKD + 128 * g + 4 * j + (lane - 20);
                    publish_b(p.G, (EQD + 2 * EKD) * 8u, (unsigned)n, tag, y, xloc);
                    if (lane >= 16) (lane < 20 ? kc : vc)[(size_t)pos * EHD + (n & 127)] = y;      // k / v rows also go to the cache (read by later steps)
                }
            }
            cs.published(tl, tlev ? 28 : -1);
            if (T) tl(1);
            kload(1);
            float2 vpre[16];      // P.V: wave = key group (keys cw + 12 u), lane = float2 column
#pragma unroll
            for (int u = 0; u < 16; u++) {
                vpre[u] = make_float2(0.f, 0.f);
                if (cw + 12 * u < n_old) {      // wave-uniform: rows that do not exist are not requested
                    const fv2 v = *(const __attribute__((address_space(1))) fv2*)(vc + ((unsigned)(j_lo + cw + 12 * u) * EHD + lane * 2));
                    vpre[u] = make_float2(v.x, v.y);
                }
            }
            wait_ge(&c->qkv_flag, (unsigned)l + 1u, c, p.err, ERR_STAGE);
            if (T) tl(2);
            {
                float qv[16];
#pragma unroll
                for (int e = 0; e < 4; e++) { const float4 v = *reinterpret_cast<const float4*>(qkvn + part_ * 16 + 4 * e); qv[4 * e] = v.x; qv[4 * e + 1] = v.y; qv[4 * e + 2] = v.z; qv[4 * e + 3] = v.w; }
                auto dot16 = [&](const float4 (&kk)[4]) {
                    float sacc = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; e++) { sacc = fmaf(qv[4 * e], kk[e].x, sacc); sacc = fmaf(qv[4 * e + 1], kk[e].y, sacc); sacc = fmaf(qv[4 * e + 2], kk[e].z, sacc); sacc = fmaf(qv[4 * e + 3], kk[e].w, sacc); }
                    return group8_sum_e(sacc);
                };
#pragma unroll
                for (int u = 0; u < NKP; u++) {
                    const float sv = dot16(kpre[u]);
                    const int i = ks + 96 * u;
                    if (part_ == 0 && i < n_old) sc[i] = sv * scale;
                }
                for (int i0 = 96 * NKP; i0 < n_old; i0 += 96 * NKP) {      // later keys, 192 per round: ALL loads of a round first (one exposed round trip per 192 keys, not per 96)
#pragma unroll
                    for (int u = 0; u < NKP; u++) {
                        const int i = i0 + ks + 96 * u;
                        const unsigned ko = (unsigned)(j_lo + min(i, last_old)) * EHD + part_ * 16;
#pragma unroll
                        for (int e = 0; e < 4; e++) kpre[u][e] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (i < n_old) {
#pragma unroll
                            for (int e = 0; e < 4; e++) kpre[u][e] = ldg4(kc + (ko + 4 * e));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NKP; u++) {
                        const float sv = dot16(kpre[u]);
                        const int i = i0 + ks + 96 * u;
                        if (part_ == 0 && i < n_old) sc[i] = sv * scale;
                    }
                }
                if (cw == 0) {                                       // the new key (this step's k row)
                    float4 kk[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) kk[e] = *reinterpret_cast<const float4*>(qkvn + 128 + part_ * 16 + 4 * e);
                    const float sv = dot16(kk);
                    if (t12 == 0) sc[n_old] = sv * scale;
                }
            }
            if (T) tl(26);
            cs.cbarrier();
            if (T) tl(27);
            {
                const int n = n_old + 1;
                float mx = -INFINITY;
                for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sc[i]);
                mx = wave_max_e(mx);
                float2 o = make_float2(0.f, 0.f); float lsum = 0.f;
                // this wave's keys cw + 12 u: lane u computes the key's weight ONCE (the 64 lanes used to repeat every exponential), the P.V loop reads it back lane by lane
                const int iu = cw + 12 * (lane & 15);
                const float pv = iu < n_old ? expf(sc[iu] - mx) : 0.f;
#pragma unroll
                for (int u = 0; u < 16; u++) {
                    const float pr_ = rlf(pv, u); const float2 vv = vpre[u];      // rows that do not exist: weight 0, value 0
                    o.x = fmaf(pr_, vv.x, o.x); o.y = fmaf(pr_, vv.y, o.y); lsum += pr_;
                }
                for (int b0 = 192; b0 < n_old; b0 += 192) {      // later keys, 16 per wave and round: all of a round's rows requested at once, one exponential per key (lane = key)
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        vpre[u] = make_float2(0.f, 0.f);
                        if (b0 + cw + 12 * u < n_old) {
                            const fv2 v = *(const __attribute__((address_space(1))) fv2*)(vc + ((unsigned)(j_lo + b0 + cw + 12 * u) * EHD + lane * 2));
                            vpre[u] = make_float2(v.x, v.y);
                        }
                    }
                    const int ib = b0 + cw + 12 * (lane & 15);
                    const float pb = ib < n_old ? expf(sc[ib] - mx) : 0.f;
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const float pr_ = rlf(pb, u); const float2 vv = vpre[u];
                        o.x = fmaf(pr_, vv.x, o.x); o.y = fmaf(pr_, vv.y, o.y); lsum += pr_;
                    }
                }
                if (cw == 0) {
                    const float pr_ = expf(sc[n_old] - mx); const float2 vv = *reinterpret_cast<const float2*>(qkvn + 256 + lane * 2);
                    o.x = fmaf(pr_, vv.x, o.x); o.y = fmaf(pr_, vv.y, o.y); lsum += pr_;
                }
                po[cw * 64 + lane] = o;
                if (lane == 0) pl[cw] = lsum;
            }
            if (T) tl(23);
            cs.cbarrier();
            if (T) tl(25);
            if (lane < 32) {      // head h's attention output: lane = 4 consecutive columns, summed over the 12 key groups (fixed order) -> digit planes (4 blocks).
                // EVERY wave does this for itself, into its own (dead since the q|k|v steps) region of the plane buffer: no barrier before wo
                const float4* po4 = reinterpret_cast<const float4*>(po);
                float4 so = make_float4(0.f, 0.f, 0.f, 0.f); float sl = 0.f;
#pragma unroll
                for (int q = 0; q < 12; q++) { const float4 a = po4[q * 32 + lane]; so.x += a.x; so.y += a.y; so.z += a.z; so.w += a.w; sl += pl[q]; }
                const float inv = 1.0f / sl;
                const float v[4] = {so.x * inv, so.y * inv, so.z * inv, so.w * inv};
                to_digits(v, lane, xs + (8 * cw + (lane >> 3)) * 128, binfo + 8 * cw + (lane >> 3));
            }
            if (T) tl(3);
            {   // ---------------- wo: rows [384 s, +384) x head h's 128 columns: this wave's tiles 2 cw, 2 cw + 1, two K-steps each ----------------
                StepCtx cx[2];
#pragma unroll
                for (int i = 0; i < 2; i++) cx[i] = load_ctx(xs, binfo, 8 * cw + 2 * i, lane);
#pragma unroll
                for (int i = 0; i < WO_PK; i++) {
                    int sl_; const unsigned char* bb = cs.slot_wait(cs.P + i, 2 * REC, sl_);
                    float a = mstep(bb, false, lane, cx[0], 0.f); a = mstep(bb + REC, false, lane, cx[1], a);
                    cs.slot_release(sl_);
                    a = g01_sum(a, lane);
                    if (lane < 16) publish_b(p.PW, NPW * ED * 8u, (unsigned)(h * ED + 384 * s + 16 * (2 * cw + i) + lane), tag, a, false);
                }
                cs.P += WO_PK;
                cs.published(tl, tlev ? 29 : -1);
            }
            if (T) tl(4);
            continue;
        }
        if (op == EOP_W13) {
            // ================= w1|w3: 72 rows (36 SwiGLU outputs) x K 3072: this wave's K-steps of 4 full tiles + the 8-row tile =================
            constexpr int NPRE = 3;      // tiles whose records wait in registers (36 VGPRs) while the post-attention stream is still on its way; their ring slots refill meanwhile
            RecR rw[4 * NPRE];
#pragma unroll
            for (int ti = 0; ti < NPRE; ti++) {
                int s0, s1;
                const unsigned char* b0 = cs.slot_wait(cs.P + 2 * ti, 2 * REC, s0);
                rw[4 * ti] = rec_read(b0, false, lane); rw[4 * ti + 1] = rec_read(b0 + REC, false, lane);
                cs.slot_release(s0);
                const unsigned char* b1 = cs.slot_wait(cs.P + 2 * ti + 1, 2 * REC, s1);
                rw[4 * ti + 2] = rec_read(b1, false, lane); rw[4 * ti + 3] = rec_read(b1 + REC, false, lane);
                cs.slot_release(s1);
            }
            cs.all_gather<true>(p.H1, p.SS1, tag, 2u * (unsigned)l + 1u, xs, ssl, binfo, tl, T);
            if (T) tl(5);
            StepCtx cx[4];
#pragma unroll
            for (int i = 0; i < 4; i++) cx[i] = load_ctx(xs, binfo, 2 * (4 * cw + i), lane);
            if (T) tl(24);
#pragma unroll
            for (int ti = 0; ti < NPRE; ti++) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) a = mstep(rw[4 * ti + i], cx[i], a);
                a = g01_sum(a, lane);
                if (lane < 16) part[(16 * ti + lane) * NCONS + cw] = a;
            }
#pragma unroll 1
            for (int ti = NPRE; ti < 4; ti++) {
                int s0, s1;
                const unsigned char* b0 = cs.slot_wait(cs.P + 2 * ti, 2 * REC, s0);
                const unsigned char* b1 = cs.slot_wait(cs.P + 2 * ti + 1, 2 * REC, s1);
                float a = mstep(b0, false, lane, cx[0], 0.f); a = mstep(b0 + REC, false, lane, cx[1], a);
                a = mstep(b1, false, lane, cx[2], a); a = mstep(b1 + REC, false, lane, cx[3], a);
                cs.slot_release(s0); cs.slot_release(s1);
                a = g01_sum(a, lane);
                if (lane < 16) part[(16 * ti + lane) * NCONS + cw] = a;
            }
            {
                int s2; const unsigned char* b2 = cs.slot_wait(cs.P + 8, 4 * REC_H, s2);
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < 4; i++) a = mstep(b2 + i * REC_H, true, lane, cx[i], a);
                cs.slot_release(s2);
                a = g01_sum(a, lane);
                if (lane < 8) part[(64 + lane) * NCONS + cw] = a;
            }
            cs.P += W13_PK;
            cs.cbarrier();
            if (cw < 2) {      // rows 2 i (gate), 2 i + 1 (up): sum the 12 K-slices (fixed order), RMSNorm scale, SwiGLU, publish to the XCD group
                const int r = min(64 * cw + lane, 71);
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < NCONS; w++) a += part[r * NCONS + w];
                a *= cs.rstd_staged(ssl);
                const float up = dppf<0xB1>(a);
                if ((lane & 1) == 0 && 64 * cw + lane < 72) publish_b(p.A, EF * 8u, (unsigned)(1152 * g + 36 * j + (r >> 1)), tag, silu_e(a) * up, xloc);
            }
            cs.published(tl, tlev ? 30 : -1);
            if (T) tl(6);
            continue;
        }
        if (op == EOP_W2) {
            // ================= w2: rows [96 j, +96) x the XCD group's 1152 columns: this wave's 3 K-steps (of 18) of 3 tiles (of 6) =================
            RecR r2[9];      // this wave's 3 K-steps of its 3 tiles -> registers while the SwiGLU outputs of the XCD group are still being exchanged
#pragma unroll
            for (int i = 0; i < W2_PK; i++) {
                int sl_; const unsigned char* bb = cs.slot_wait(cs.P + i, 3 * REC, sl_);
#pragma unroll
                for (int k = 0; k < 3; k++) r2[3 * i + k] = rec_read(bb + k * REC, false, lane);
                cs.slot_release(sl_);
            }
            wait_ge(&c->xa_flag, (unsigned)l + 1u, c, p.err, ERR_STAGE);
            if (T) tl(7);
            const int ksl = cw % 6, tg = cw / 6;
            if (lane < 48) {      // this wave's 6 blocks (K-steps 3 ksl .. 3 ksl + 2) of the group's SwiGLU outputs: f32 -> digit planes in the wave's own region -- no barrier
                const float4 v4 = *reinterpret_cast<const float4*>(xa + 32 * (6 * ksl + (lane >> 3)) + 4 * (lane & 7));
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                to_digits(v, lane, xs + (8 * cw + (lane >> 3)) * 128, binfo + 8 * cw + (lane >> 3));
            }
            StepCtx cx[3];
#pragma unroll
            for (int i = 0; i < 3; i++) cx[i] = load_ctx(xs, binfo, 8 * cw + 2 * i, lane);
#pragma unroll
            for (int i = 0; i < W2_PK; i++) {
                float a = mstep(r2[3 * i], cx[0], 0.f); a = mstep(r2[3 * i + 1], cx[1], a); a = mstep(r2[3 * i + 2], cx[2], a);
                a = g01_sum(a, lane);
                if (lane < 16) part2[(16 * (3 * tg + i) + lane) * 6 + ksl] = a;
            }
            cs.P += W2_PK;
            cs.cbarrier();
            if (cw < 2) {
                const int r = min(64 * cw + lane, 95);
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 6; w++) a += part2[r * 6 + w];      // fixed order
                if (64 * cw + lane < 96) publish_b(p.P2, NP2 * ED * 8u, (unsigned)(g * ED + 96 * j + r), tag, a, false);
            }
            cs.published(tl, tlev ? 31 : -1);
            if (T) tl(14);
            continue;
        }
        // ================= lm_head (VALU path: one row per wave per packet) =================
        const float rstd = cs.all_gather<false>(p.H0, p.SS0, tag_base + (unsigned)p.n_layers, 2u * (unsigned)p.n_layers, xs, ssl, binfo, tl, false);
        XA xr; xr.load(reinterpret_cast<const float*>(xs), lane, 64 + (lane >> 1), lane & 1);
        const int n_pass = lm_packets(p.vocab);
        PassA Qa, Qb;
        cs.fetch(cs.P, Qa);
#pragma unroll 1
        for (int t = 0; t < n_pass; t++) {
            const int q = cw + NCONS * t;                              // this wave's pass of packet t
            if (t + 1 < n_pass) cs.fetch(cs.P + t + 1, Qb);
            if (q < npass_lm) {
                const float r = wave_sum_e(Cons::dot(Qa, xr)) * rstd;
                const int n = row0_lm + q;
                if (p.logits_out && lane == 0) p.logits_out[n] = r;
                if (r > best || (r == best && n < best_i)) { best = r; best_i = n; }
            }
            Qa = Qb;
        }
        cs.P += (unsigned)n_pass;
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
#ifdef ENG_PRIO_LOADER
        __builtin_amdgcn_s_setprio(ENG_PRIO_LOADER);
#endif
        if (ENG_ROLES & 1) eng_loader(p, c, (unsigned)(uintptr_t)(lds + L_RING), lane, tl);
    } else if (wave == 1) {
#ifdef ENG_PRIO_COMM
        __builtin_amdgcn_s_setprio(ENG_PRIO_COMM);
#endif
        if (ENG_ROLES & 2) eng_comm(p, c, lds, lane, tl);
        // the last workgroup-independent act of the launch: bump the serial (every workgroup has read it long before any lm_head input existed)
        if (blockIdx.x == 0 && lane == 0) {
            const unsigned sv = *p.serial; asm volatile("" ::: "memory"); *p.serial = sv + 1u;
            if (p.flags & ENGF_ARGMAX_IN) { const int pv_ = *p.pos_rw; asm volatile("" ::: "memory"); *p.pos_rw = pv_ + 1; }      // likewise: every workgroup read the position when it started
        }
    } else {
#ifdef ENG_PRIO_CONS
        __builtin_amdgcn_s_setprio(ENG_PRIO_CONS);
#endif
        if (ENG_ROLES & 4) eng_consumer(p, c, lds, wave - 2, lane, tl);
    }
}

// the final norm's output of the launch that has just ended (see launch_eng_hidden)
__global__ __launch_bounds__(256) void eng_hidden_kernel(const u64* __restrict__ H, const u64* __restrict__ SS, float eps, float* __restrict__ out) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    red[t] = __uint_as_float((unsigned)SS[t]);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (t < s) red[t] += red[t + s]; __syncthreads(); }
    const float rstd = 1.0f / sqrtf(red[0] / (float)ED + eps);
    const int k = blockIdx.x * 256 + t;
    out[k] = __uint_as_float((unsigned)H[k]) * (1.0f / 512.0f) * rstd;
}

}  // namespace

hipError_t launch_eng_hidden(const EngParams& p, float* out, hipStream_t s) {
    const bool even = ((2 * p.n_layers) & 1) == 0;      // stage 2 L is always even: H0 / SS0
    eng_hidden_kernel<<<dim3(ED / 256), dim3(256), 0, s>>>(even ? p.H0 : p.H1, even ? p.SS0 : p.SS1, p.eps, out);
    return hipGetLastError();
}

bool eng_geometry_ok(int D, int n_heads, int n_kv, int hd, int ffn, int vocab, int max_seq) {
    return D == ED && n_heads == ENH && n_kv == ENKV && hd == EHD && ffn == EF && vocab > 0 && vocab % NCU == 0 && max_seq > 0 && max_seq <= SC_MAX;
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
    int off_bytes, N, K;
    switch (op) {
    case EOP_QKV: off_bytes = OFF_QKV; N = EQD + 2 * EKD; K = ED; break;
    case EOP_WO: off_bytes = OFF_WO; N = ED; K = EQD; break;
    case EOP_W13: off_bytes = OFF_W13; N = 2 * EF; K = ED; break;
    case EOP_W2: off_bytes = OFF_W2; N = ED; K = EF; break;
    case EOP_LM: off_bytes = 0; N = vocab; K = ED; break;
    case EOP_WOB: off_bytes = 0; N = ED; K = EQD; break;      // `stream` = the batched engine's wo stream ([layer][packet][CU][bytes])
    default: return hipErrorInvalidValue;
    }
    if (w.N != N || w.K != K) return hipErrorInvalidValue;
    if (op == EOP_LM) eng_pack_lm_kernel<<<dim3(lm_passes(vocab), NCU), dim3(64), 0, s>>>(w, stream, (size_t)n_layers * LAYER_BYTES, vocab);
    else if (op == EOP_WOB) eng_pack_kernel<<<dim3(op_packets(op) * NCONS * 4, NCU), dim3(64), 0, s>>>(w, op, stream, (size_t)layer * WOB_LAYER_BYTES);
    else eng_pack_kernel<<<dim3(op_packets(op) * NCONS * 4, NCU), dim3(64), 0, s>>>(w, op, stream, (size_t)layer * LAYER_BYTES + off_bytes);
    return hipGetLastError();
}

// resident workgroups per CU the runtime grants this kernel: the engine's 256 workgroups spin on each other and must all be resident (>= 1 per CU on a 256-CU device)
hipError_t eng_occupancy(int* blocks_per_cu) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, reinterpret_cast<const void*>(decode_engine_kernel), NTHR, L_TOTAL);
}

hipError_t launch_decode_engine(const EngParams& p, hipStream_t s) {
    static DevOnce attr_done;
    const int dev = vox_current_device();
    if (!attr_done.done(dev)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(decode_engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.set(dev);
    }
    decode_engine_kernel<<<dim3(NCU), dim3(NTHR), L_TOTAL, s>>>(p);
    return hipGetLastError();
}

}  // namespace vox
