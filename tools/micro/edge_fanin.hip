// Microbenchmark (not product code), round 3: the decode engine's dependency edges IN ISOLATION (no weight stream, no consumer waves).
// A "round" = every workgroup publishes its granules, then sweeps the granules it depends on until every tag matches; rounds are dependent, so
// time / rounds = the cost of one edge.  Edges (vox_engine.hip):
//   A   XCD-local fan-in: the 32 workgroups of a group (b & 7) publish 36 granules each, every one sweeps the group's 1152 (18 per lane)
//   G   XCD-local: 12 granules per workgroup, sweep 384 (6 per lane)
//   H   all-gather across the chip: 12 granules per workgroup (write-through), every workgroup sweeps all 3072 (48 per lane) [+ probe first]
//   PW  partial planes: workgroup (h, s) publishes 384 granules of plane h; owner b sweeps 32 x 12
// Variants: load flavour (sc1 / nt / s_load probe), probe-before-sweep, s_sleep between polls, idle spinning waves beside the sweeping wave.
// Every spin is bounded (3 ms) -- the program cannot hang the box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t srd_t;
__device__ __forceinline__ srd_t make_srd(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
template <int AUX>
__device__ __forceinline__ unsigned ld_tag(srd_t srd, unsigned idx) { const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(srd, (int)(idx * 8u), 0, AUX); return v.y; }
template <int AUX>
__device__ __forceinline__ void st_gran(srd_t srd, unsigned idx, unsigned tag) { v2u_t x; x.x = idx; x.y = tag; __builtin_amdgcn_raw_buffer_store_b64(x, srd, (int)(idx * 8u), 0, AUX); }

struct Args { u64* buf; unsigned bytes; int rounds; unsigned base; u64* ticks; unsigned* fail; int sleep; int spinners; int probe; };

// EDGE: 0 A, 1 G, 2 H, 3 PW.   LAUX: aux bits of the sweep loads (16 = sc1, 2 = nt, 17 = sc0 sc1).  SAUX: aux of the stores (0 plain, 16 sc1).
template <int EDGE, int LAUX, int SAUX>
__global__ __launch_bounds__(512) void fanin(const Args a) {
    extern __shared__ unsigned char pad_lds[];
    __shared__ unsigned go;
    const int b = blockIdx.x, g = b & 7, j = b >> 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) go = 0;
    __syncthreads();
    const srd_t srd = make_srd(a.buf, a.bytes);
    if (wave > 0) {      // optional idle waves: spin on an LDS word like the engine's consumer waves do while an edge resolves
        if (wave <= a.spinners) { unsigned n = 0; while (__hip_atomic_load(&go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && ++n < (1u << 24)) __builtin_amdgcn_s_sleep(1); }
        return;
    }
    bool dead = false;
    const u64 t0 = wall_clock64();
    for (int r = 0; r < a.rounds && !dead; r++) {
        const unsigned tag = a.base + (unsigned)r + 1u;
        constexpr int N = EDGE == 0 ? 18 : EDGE == 1 ? 6 : EDGE == 2 ? 48 : 6;
        // ---- publish ----
        if (EDGE == 0) { if (lane < 36) st_gran<SAUX>(srd, (unsigned)(1152 * g + 36 * j + lane), tag); }
        else if (EDGE == 1) { if (lane < 12) st_gran<SAUX>(srd, (unsigned)(384 * g + 12 * j + lane), tag); }
        else if (EDGE == 2) { if (lane < 12) st_gran<SAUX>(srd, (unsigned)(12 * b + lane), tag); }
        else { const int h = 4 * g + (j >> 3), s = j & 7; for (int u = 0; u < 6; u++) st_gran<SAUX>(srd, (unsigned)(h * 3072 + 384 * s + lane + 64 * u), tag); }
        // ---- sweep ----
        auto idx = [&](int u) -> unsigned {
            if (EDGE == 0) return (unsigned)(1152 * g + lane + 64 * u);
            if (EDGE == 1) return (unsigned)(384 * g + lane + 64 * u);
            if (EDGE == 2) return (unsigned)(lane + 64 * u);
            const int i = lane + 64 * u, hh = i / 12, rr = i - hh * 12; return (unsigned)(hh * 3072 + 12 * b + rr);
        };
        const u64 ts = wall_clock64(); unsigned n = 0;
        if (a.probe) {      // one granule per lane: the last one of each producer (A, G), one row of every 4th producer (H), one per plane (PW)
            const unsigned pi = EDGE == 0 ? (unsigned)(1152 * g + 36 * (lane & 31) + 35) : EDGE == 1 ? (unsigned)(384 * g + 12 * (lane & 31) + 11) : EDGE == 2 ? 48u * (unsigned)lane : (unsigned)((lane & 31) * 3072 + 12 * b);
            while (!__all((int)(ld_tag<LAUX>(srd, pi) - tag) >= 0)) {      // >=: a fast neighbour may already have published the next round
                if (a.sleep) __builtin_amdgcn_s_sleep(2);
                if ((++n & 63u) == 0 && wall_clock64() - ts > 300000ull) { dead = true; break; }
            }
        }
        while (!dead) {
            bool ok = true;
            unsigned t[N];
#pragma unroll
            for (int u = 0; u < N; u++) t[u] = ld_tag<LAUX>(srd, idx(u));
#pragma unroll
            for (int u = 0; u < N; u++) ok &= (int)(t[u] - tag) >= 0;
            if (__all(ok)) break;
            if (a.sleep) __builtin_amdgcn_s_sleep(2);
            if ((++n & 63u) == 0 && wall_clock64() - ts > 300000ull) { dead = true; break; }
        }
    }
    if (dead && lane == 0) __hip_atomic_store(a.fail, 1u + (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) { a.ticks[b] = wall_clock64() - t0; __hip_atomic_store(&go, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
}

static unsigned g_base = 1;
template <int EDGE, int LAUX, int SAUX>
static void run(Args a, const char* name, int probe, int sleep, int spinners) {
    static bool attr = false;
    if (!attr) { CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fanin<EDGE, LAUX, SAUX>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); attr = true; }
    a.base = g_base; g_base += (unsigned)a.rounds + 8u; a.probe = probe; a.sleep = sleep; a.spinners = spinners;
    CHK(hipMemset(a.fail, 0, 4));
    fanin<EDGE, LAUX, SAUX><<<256, 512, 100 * 1024>>>(a);
    CHK(hipDeviceSynchronize());
    unsigned f; CHK(hipMemcpy(&f, a.fail, 4, hipMemcpyDeviceToHost));
    std::vector<u64> t(256); CHK(hipMemcpy(t.data(), a.ticks, 256 * 8, hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    printf("  %-44s probe %d sleep %d spinners %d : ", name, probe, sleep, spinners);
    if (f) printf("FAIL (workgroup %u timed out)\n", f - 1); else printf("%6.2f us per edge (median), %6.2f max\n", (double)t[128] / a.rounds / 100.0, (double)t[255] / a.rounds / 100.0);
}

int main() {
    Args a{};
    a.bytes = 32 * 3072 * 8; a.rounds = 200;
    CHK(hipMalloc(&a.buf, a.bytes)); CHK(hipMemset(a.buf, 0, a.bytes));
    CHK(hipMalloc(&a.ticks, 256 * 8)); CHK(hipMalloc(&a.fail, 4));
    for (int spinners : {0, 7}) {
        run<0, 16, 0>(a, "A  (36 x 32 -> 1152) plain store, sc1 loads", 0, 1, spinners);
        run<0, 16, 0>(a, "A  plain store, sc1 loads", 1, 1, spinners);
        run<0, 16, 0>(a, "A  plain store, sc1 loads", 0, 0, spinners);
        run<0, 2, 0>(a, "A  plain store, nt loads", 0, 1, spinners);
        run<0, 16, 16>(a, "A  sc1 store, sc1 loads", 0, 1, spinners);
        run<1, 16, 0>(a, "G  (12 x 32 -> 384) plain store, sc1 loads", 0, 1, spinners);
        run<1, 16, 16>(a, "G  sc1 store, sc1 loads", 0, 1, spinners);
        run<2, 16, 16>(a, "H  (12 x 256 -> 3072) sc1 store, sc1 loads", 0, 1, spinners);
        run<2, 16, 16>(a, "H  sc1 store, sc1 loads", 1, 1, spinners);
        run<2, 17, 16>(a, "H  sc1 store, sc0 sc1 loads", 1, 1, spinners);
        run<2, 16, 16>(a, "H  sc1 store, sc1 loads", 1, 0, spinners);
        run<3, 16, 16>(a, "PW (384 x 256 -> 32 x 12) sc1 store, sc1 loads", 0, 1, spinners);
        run<3, 16, 16>(a, "PW sc1 store, sc1 loads", 1, 1, spinners);
    }
    return 0;
}
