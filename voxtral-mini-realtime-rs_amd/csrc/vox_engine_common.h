// vox_engine_common.h -- pieces shared by the persistent decode engines (vox_engine.hip: single stream; vox_engine_b16.hip: <= 16 rows):
// the packet-stream layout (both engines consume the SAME per-CU weight stream), buffer / DPP / LDS helpers, bounded waits, the LDS-DMA loader wave
// and the granule sweep.  Included inside `namespace vox { namespace {` of each translation unit.
#pragma once

typedef unsigned long long u64;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int ED = 3072, ENH = 32, ENKV = 8, EHD = 128, EQD = ENH * EHD, EKD = ENKV * EHD, EF = 9216;
constexpr int NCU = 256, NCONS = 12, NWAVES = NCONS + 2, NTHR = 64 * NWAVES;      // 14 waves: 3-4 per SIMD (one wave per SIMD issues a VALU instruction only every ~8 cycles)
// ---- layer operators: 16-row x 64-column MFMA steps (v_mfma_i32_16x16x64_i8) ----
// One STEP RECORD = what the 64 lanes of a wave feed one MFMA: lane (n = lane & 15, g = lane >> 4) holds 8 bytes of nibbles = half g & 1 of Q4 block
// 2 T + (g >> 1) of tile row n (its elements 8 h .. 8 h + 7 and 16 + 8 h .. 16 + 8 h + 7), then the 2 x 16 f16 block scales [block parity][n]: 576 bytes.
// A tile of 8 rows (the k|v rows of q|k|v, the last tile of w1|w3) stores lanes n < 8 only: 288 bytes.
// A packet holds, for each of the 12 consumer waves, the records of ITS K-steps (the K range is split over the waves; the row tiles are shared):
//   q|k|v (24 rows x 48 K-steps: wave w owns steps 4 w .. 4 w + 3)   packets 0, 1 = tile 0 steps {0, 1}, {2, 3}; packet 2 = the 8-row tile, 4 steps    3 x 13824 B
//   wo    (384 rows x 2 steps: wave w owns tiles 2 w, 2 w + 1)       packet i = tile 2 w + i, both steps                                          2 x 13824 B
//   w1|w3 (72 rows x 48 steps: wave w owns steps 4 w ..)             packets 2 i, 2 i + 1 = tile i < 4; packet 8 = the 8-row tile                9 x 13824 B
//   w2    (96 rows x 18 steps: wave w owns steps 3 (w % 6) .. of tiles 3 (w / 6) ..)   packet i = tile 3 (w / 6) + i                        3 x 20736 B
// lm_head: one ROW per wave per packet (VALU path): [64] x 16 B nibbles of block `lane` | [64] x 8 B: half (lane & 1) of block 64 + lane / 2 | [64] x 2 B
// scales | [32] x 2 B scales of the split blocks = 1728 B per pass, 20736 B per packet.
constexpr int REC = 576, REC_H = 288, REC_SC = 512, REC_H_SC = 256;
constexpr int PASS_A = 1728;
constexpr int PA_Q1 = 1024, PA_S0 = 1536, PA_S1 = 1664;
constexpr int PK_A = NCONS * PASS_A, PK_M = NCONS * 2 * REC;    // 20736 / 13824 bytes, stored back to back (no padding)
constexpr int LINES_A = (PK_A + 1023) / 1024, LINES_M = (PK_M + 1023) / 1024;      // LDS-DMA instructions per packet (the last one partial: 16 / 32 lanes)
static_assert(LINES_A == 21 && LINES_M == 14, "wait_vmcnt() enumerates the in-flight line counts 14 / 21 / 28 / 35 / 42");
constexpr int SLOT_BYTES = PK_A, NSLOT = 6;
constexpr int QKV_PK = 3, WO_PK = 2, W13_PK = 9, W2_PK = 3, PK_LAYER = QKV_PK + WO_PK + W13_PK + W2_PK, PK_LAYER_M = QKV_PK + WO_PK + W13_PK;      // the first 14 packets of a layer are 13824 bytes
constexpr int LAYER_BYTES = PK_LAYER_M * PK_M + W2_PK * PK_A;   // 255744 bytes per CU per layer = exactly the Q4 bytes
constexpr int OFF_QKV = 0, OFF_WO = QKV_PK * PK_M, OFF_W13 = OFF_WO + WO_PK * PK_M, OFF_W2 = OFF_W13 + W13_PK * PK_M;   // byte offsets inside a layer
static_assert(LAYER_BYTES == 255744, "layer bytes");
constexpr int SC_MAX = 1024;                          // attention scores in LDS: cache rows per KV head (max_seq) <= 1024
constexpr int OWN = ED / NCU;                         // 12 rows of the residual stream per CU
constexpr int NPW = ENH, NP2 = ENKV;                  // partial planes of wo (one per head) / w2 (one per XCD group: the whole 1152-column slice in one CU)
constexpr int NPWB = ENKV;                            // batched engine: wo planes, one per XCD group (EOP_WOB)
constexpr size_t WOB_LAYER_BYTES = (size_t)WO_PK * PK_M;      // per CU and layer
constexpr u64 TIMEOUT_TICKS = 2000000;                // s_memrealtime ticks (100 MHz): 20 ms

enum { EOP_QKV = 0, EOP_WO = 1, EOP_W13 = 2, EOP_W2 = 3, EOP_LM = 4, EOP_WOB = 5 };
// EOP_WOB: wo for the BATCHED engine, K split by XCD group instead of by head (its own small stream, [layer][packet 2][CU][13824] -- the single-stream engine keeps EOP_WO,
// whose per-head split needs no attention edge at one row): CU (g, j) multiplies rows [96 j, +96) (6 tiles) by the 512 columns of the group's four heads (8 K-steps) ->
// 8 partial planes like w2 instead of 32.  Same packet geometry as EOP_WO (2 packets x 12 waves x 2 records): wave w owns tile w % 6, K-steps 4 (w / 6) .. + 3.
enum { ERR_RING = 1, ERR_STAGE = 2, ERR_SWEEP = 3, ERR_CBAR = 4, ERR_SLOT = 5 };

__host__ __device__ inline int lm_rows_per_cu(int vocab) { return vocab / NCU; }
__host__ __device__ inline int lm_passes(int vocab) { return lm_rows_per_cu(vocab); }      // one row per pass
__host__ __device__ inline int lm_packets(int vocab) { return (lm_passes(vocab) + NCONS - 1) / NCONS; }
__host__ __device__ inline size_t cu_stream_bytes(int n_layers, int vocab) { return (size_t)n_layers * LAYER_BYTES + (size_t)lm_packets(vocab) * PK_A + 1024; }      // + 1 KiB: the stream is read in whole 16-byte lanes only, the pad keeps the allocation comfortable

// ---- record addressing (shared by the pack kernel and -- implicitly, through the same formulas -- the consumer waves) ----
__host__ __device__ inline int op_packets(int op) { return op == EOP_QKV ? QKV_PK : (op == EOP_WO || op == EOP_WOB) ? WO_PK : op == EOP_W13 ? W13_PK : W2_PK; }
__host__ __device__ inline int op_pk_bytes(int op) { return op == EOP_W2 ? PK_A : PK_M; }
__host__ __device__ inline bool pk_half_tile(int op, int pk) { return (op == EOP_QKV && pk == 2) || (op == EOP_W13 && pk == 8); }
__host__ __device__ inline int pk_steps(int op, int pk) { return op == EOP_W2 ? 3 : pk_half_tile(op, pk) ? 4 : 2; }      // records per wave in the packet
// record s of wave w in packet pk: the 16-row tile and the K-step (64 columns) of the operator's K range on this CU
__host__ __device__ inline void rec_src(int op, int pk, int w, int s, int* tile, int* T) {
    if (op == EOP_QKV) { if (pk < 2) { *tile = 0; *T = 4 * w + 2 * pk + s; } else { *tile = 1; *T = 4 * w + s; } }
    else if (op == EOP_W13) { if (pk < 8) { *tile = pk >> 1; *T = 4 * w + 2 * (pk & 1) + s; } else { *tile = 4; *T = 4 * w + s; } }
    else if (op == EOP_W2) { *tile = 3 * (w / 6) + pk; *T = 3 * (w % 6) + s; }
    else if (op == EOP_WOB) { *tile = w % 6; *T = 4 * (w / 6) + 2 * pk + s; }
    else { *tile = 2 * w + pk; *T = s; }
}
__host__ __device__ inline int tile_row(int op, int b, int tile, int n) {      // weight-matrix row of tile row n on CU b
    const int g = b & 7, j = b >> 3, h = 4 * g + (j >> 3), s = j & 7;
    if (op == EOP_QKV) return tile == 0 ? 128 * h + 16 * s + n : n < 4 ? EQD + 128 * g + 4 * j + n : EQD + EKD + 128 * g + 4 * j + (n - 4);
    if (op == EOP_W13) return 2 * (1152 * g + 36 * j) + 16 * tile + n;      // interleaved gate / up rows: SwiGLU output i = rows 2 i, 2 i + 1
    if (op == EOP_W2 || op == EOP_WOB) return 96 * j + 16 * tile + n;
    return 384 * s + 16 * tile + n;
}
__host__ __device__ inline int step_blk0(int op, int b, int T) {      // first Q4 block (of two) of K-step T
    const int g = b & 7, j = b >> 3, h = 4 * g + (j >> 3);
    return op == EOP_W2 ? 36 * g + 2 * T : op == EOP_WO ? 4 * h + 2 * T : op == EOP_WOB ? 16 * g + 2 * T : 2 * T;
}


// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
#define ENG_CFENCE() asm volatile("" ::: "memory")
typedef const __attribute__((address_space(1))) float* gcf_p;      // pointers that come out of the device-resident layer table: the compiler cannot infer
typedef __attribute__((address_space(1))) float* gf_p;             // their address space, and a FLAT load also counts on lgkmcnt (an LDS wait would wait for it)
typedef float fv4 __attribute__((ext_vector_type(4)));
typedef float fv2 __attribute__((ext_vector_type(2)));
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
// XCD-local form (producer and reader behind the same L2, verified at start-up): nt -- misses this CU's L1 but is served by the L2 at hit latency even while the
// granule is still stale, where an sc1 poll of a clean line is a round trip to memory (tools/micro/edge_pingpong.hip: 0.31 us per hop with plain stores)
__device__ __forceinline__ u64 ld_gran_l(srd_t srd, unsigned idx) { const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(srd, (int)(idx * 8u), 0, 2); return ((u64)v.y << 32) | (u64)v.x; }
__device__ __forceinline__ float ld_gf(srd_t srd, unsigned idx) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(srd, (int)(idx * 4u), 0, 0)); }
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, RLX, WG)); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, RLX, WG); }

template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
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

// wait until *word >= target (LDS word, monotonic).  Bounded; a dead workgroup never waits.  The clock (s_memrealtime: a round trip of its own, far
// longer than the LDS poll) is read once per 256 polls only -- read on every poll it WAS the latency of every flag hand-off.
template <class Ctl>
__device__ __forceinline__ bool wait_ge(unsigned* word, unsigned target, Ctl* c, unsigned* err, unsigned code) {
    if (lds_ld(word) >= target) { ENG_CFENCE(); return true; }
    if (lds_ld(&c->dead)) return false;
    u64 t0 = 0; unsigned n = 0;
    for (;;) {
        __builtin_amdgcn_s_sleep(1);
        if (lds_ld(word) >= target) break;
        if ((++n & 255u) != 0) continue;
        if (lds_ld(&c->dead)) return false;
        const u64 t = wall_clock64();
        if (t0 == 0) t0 = t;
        else if (t - t0 > TIMEOUT_TICKS) {
            lds_st(&c->dead, 1u);
            unsigned ev = code | ((unsigned)blockIdx.x << 8);
            asm volatile("" : "+s"(ev));      // built here, on the cold path: hoisted out of the item loop it costs a VGPR (or a scratch slot) everywhere
            __hip_atomic_store(err, ev, RLX, AG);
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
// four consecutive 1 KiB lines with ONE M0 write: the instruction offset (0 / 1024 / 2048 / 3072) is added to the global address AND to the LDS address (probed in
// tools/micro/engine_bench).  The loader wave is a lone wave issuing ~one instruction per 8 cycles: at 11 scalar instructions per line (M0 save / set / restore,
// address updates, loop) a layer's 259 lines cost ~6 us of pure issue time -- this form needs 10 per four lines.
__device__ __forceinline__ void dma_lines4(unsigned voff, unsigned lds_dst_, u64 gsrc_) {
    unsigned keep;
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_);
    const u64 gsrc = ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(gsrc_ >> 32)) << 32) | (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)gsrc_);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gsrc) : "memory");      // (leaving M0 unsaved / unrestored was measured: no faster)
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

template <class Ctl, int NS>
struct Loader {
    Ctl* c; unsigned* err; unsigned ring_lds; unsigned voff;
    int nfl = 0, s0 = 0, l0 = 0, s1 = 0, l1 = 0;    // packets issued, not yet published: (s0, l0) oldest, (s1, l1) newer -- plain scalars (an indexed array would live in scratch = VMEM)
    unsigned P = 0;                                   // next packet index
    int depth = 3;                                    // packets in flight (measurement knob: flags 1024 -> 2, 2048 -> 1)
    bool thin; u64 pace = 0, t_last = 0, pause_ticks = 0;              // pace: minimum s_memrealtime ticks between two packet issues (0: none)
    __device__ __forceinline__ void publish_slot(int slot) { lds_st(&c->ring_ready[slot], lds_ld(&c->ring_ready[slot]) + 1u); }   // only this wave writes ring_ready
    __device__ __forceinline__ void flush() {
        if (nfl == 2) { wait_vmcnt(l1); publish_slot(s0); s0 = s1; l0 = l1; nfl = 1; }
        if (nfl == 1) { wait_vmcnt(0); publish_slot(s0); nfl = 0; }
    }
    template <int bytes>      // 13824 or 20736: the line loop is straight-line code (a lone wave issues ~one instruction per 8 cycles: every scalar instruction of this loop is streaming time)
    __device__ __forceinline__ void issue(u64 gsrc, int lane, bool nodma) {
        constexpr int full = bytes >> 10, tail = (bytes & 1023) >> 4, lines = full + (tail ? 1 : 0);      // tail: lanes of the last, partial LDS-DMA instruction
        const int slot = (int)(P % NS); const unsigned k = P / NS;
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
#pragma unroll
            for (int i = 0; i + 4 <= full; i += 4) dma_lines4(voff, dst + (unsigned)i * 1024u, gsrc + (u64)i * 1024u);
#pragma unroll
            for (int i = full & ~3; i < full; i++) dma_line(voff, dst + (unsigned)i * 1024u, gsrc + (u64)i * 1024u);
            if (lane < tail) dma_line(voff, dst + (unsigned)full * 1024u, gsrc + (u64)full * 1024u);      // EXEC-masked: only `tail` lanes write
        }
        P++;
        if (nfl == 2) { wait_vmcnt(l1 + lines); publish_slot(s0); s0 = s1; l0 = l1; s1 = slot; l1 = lines; }      // three in flight: retire the oldest
        else if (nfl == 1) {
            if (depth <= 2) { wait_vmcnt(lines); publish_slot(s0); s0 = slot; l0 = lines; }      // depth 2: retire the older one right away
            else { s1 = slot; l1 = lines; nfl = 2; }
        } else { s0 = slot; l0 = lines; nfl = 1; if (depth <= 1) { wait_vmcnt(0); publish_slot(s0); nfl = 0; } }
    }
};

// ------------------------------------------------------------------------------------------------
// COMM wave
// ------------------------------------------------------------------------------------------------
template <class Ctl>
__device__ __forceinline__ bool sweep_bail(u64& t0, unsigned tag, Ctl* c, unsigned* err) {      // t0: low 8 bits = failed polls since the last clock read
    __builtin_amdgcn_s_sleep(2);
    if (((++t0) & 31u) != 0) return false;        // the clock is a long round trip: one read per 32 failed polls
    if (lds_ld(&c->dead)) return true;
    const u64 t = wall_clock64() << 8;
    if (t0 < 256) t0 = t;
    else if (t - (t0 & ~(u64)255) > (TIMEOUT_TICKS << 8)) {
        lds_st(&c->dead, 1u);
        unsigned ev = (unsigned)ERR_SWEEP | ((unsigned)blockIdx.x << 8) | (tag << 16);
        asm volatile("" : "+s"(ev));
        __hip_atomic_store(err, ev, RLX, AG);
        return true;
    }
    return false;
}
// Sweep N granules per lane until every tag matches (values in v).  Bounded.  With `do_probe` the wave first polls ONE granule per lane (`probe()`:
// one granule of every producer, or of every n-th) instead of the whole set: 256 CUs polling 8 KB each would put TB/s of coherent reads next to the
// weight stream (MI355X_MICROARCH.md polling-cost).
template <int N, class IdxF, class ProbeF, class Ctl>
__device__ __forceinline__ bool sweep(const u64* base_, unsigned bytes, unsigned tag, IdxF idx, ProbeF probe, bool do_probe, float (&v)[N], Ctl* c, unsigned* err, bool local = false) {
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
            const u64 gq = local ? ld_gran_l(base, (unsigned)idx(u)) : ld_gran(base, (unsigned)idx(u));
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
