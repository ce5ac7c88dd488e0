// Microbenchmark (not product code), round 3: what ONE producer -> consumer hop between two CUs costs on gfx950, per cache-policy flavour of the
// store and of the polling load, for a pair of workgroups on the SAME XCD (blockIdx b and b ^ 8) and on DIFFERENT XCDs (b and b ^ 1).
// Ping-pong of an 8-byte {value, tag} granule: A stores, B polls until it sees the tag and stores its own granule, A polls -- one iteration = two hops.
// Question behind it (decode engine, vox_engine.hip): the XCD-local edges (q|k|v -> attention, SwiGLU -> w2) poll with sc1 loads, which L2 may serve
// from MEMORY (~1-2 us) while the granule is not yet dirty in L2; a load flavour that is served by the XCD's L2 would cut those edges to a fraction.
// Every spin is bounded; a flavour whose polls never see the data (e.g. served by the CU's own L1) reports FAIL instead of hanging.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;

template <int ST>
__device__ __forceinline__ void st_gran(u64* p, u64 v) {
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ u64 ld_gran(const u64* p) {
    u64 v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LD == 3) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LD == 4) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(u64)p), hi = __builtin_amdgcn_readfirstlane((unsigned)((u64)p >> 32));
        const u64 sp = ((u64)hi << 32) | lo; u64 sv;
        asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(sp) : "memory");
        v = sv;
    } else asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 now() { return wall_clock64(); }      // 100 MHz

template <int ST, int LD>
__global__ __launch_bounds__(64) void pingpong(u64* slots, int iters, int delta, int active_mod, unsigned base, u64* ticks, unsigned* fail, unsigned* xcc) {
    extern __shared__ unsigned char pad_lds[];      // 100 KB: one workgroup per CU
    const int b = blockIdx.x, partner = b ^ delta;
    const bool is_a = (b & delta) == 0;
    const int a_id = is_a ? b : partner, pair = (a_id / (2 * delta)) * delta + (a_id % delta);
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    if (pair % active_mod != 0 || threadIdx.x != 0) return;
    u64* mine = slots + (size_t)b * 16; const u64* theirs = slots + (size_t)partner * 16;
    bool dead = false;
    const u64 t0 = now();
    for (int i = 0; i < iters && !dead; i++) {
        const u64 v = ((u64)(base + (unsigned)i + 1u) << 32) | (u64)(unsigned)i;
        if (is_a) st_gran<ST>(mine, v);
        const u64 ts = now(); unsigned n = 0;
        while ((ld_gran<LD>(theirs) >> 32) != (v >> 32)) {
            __builtin_amdgcn_s_sleep(1);
            if ((++n & 63u) == 0 && now() - ts > 300000ull) { dead = true; __hip_atomic_store(fail, 1u + (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // 3 ms
        }
        if (!is_a) st_gran<ST>(mine, v);
    }
    if (is_a) ticks[pair] = now() - t0;
}

static const char* ST_NAME[] = {"plain", "sc0", "sc1", "sc0 sc1", "nt"};
static const char* LD_NAME[] = {"sc0", "sc1", "sc0 sc1", "nt", "s_load glc", "plain"};

template <int ST, int LD>
static void run(u64* slots, u64* ticks, unsigned* fail, unsigned* xcc, unsigned& base, int iters) {
    static bool attr = false;
    if (!attr) { CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong<ST, LD>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)); attr = true; }
    printf("  store %-8s load %-10s", ST_NAME[ST], LD_NAME[LD]);
    for (int delta : {8, 1}) for (int active_mod : {17, 1}) {
        CHK(hipMemset(fail, 0, 4)); CHK(hipMemset(ticks, 0, 128 * 8));
        pingpong<ST, LD><<<256, 64, 100 * 1024>>>(slots, iters, delta, active_mod, base, ticks, fail, xcc);
        CHK(hipDeviceSynchronize());
        base += (unsigned)iters + 8u;
        unsigned f; CHK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        std::vector<u64> t(128); CHK(hipMemcpy(t.data(), ticks, 128 * 8, hipMemcpyDeviceToHost));
        std::vector<double> us; for (int p = 0; p < 128; p++) if (p % active_mod == 0) us.push_back((double)t[p] / iters / 100.0 / 2.0);      // per HOP
        std::sort(us.begin(), us.end());
        if (f) printf(" | %s/%-3d   FAIL   ", delta == 8 ? "same" : "diff", (127 / active_mod) + 1);
        else printf(" | %s/%-3d %5.2f %5.2f", delta == 8 ? "same" : "diff", (127 / active_mod) + 1, us[us.size() / 2], us.back());
    }
    printf("\n");
}

int main() {
    u64* slots; u64* ticks; unsigned* fail; unsigned* xcc;
    CHK(hipMalloc(&slots, 256 * 128)); CHK(hipMemset(slots, 0, 256 * 128));
    CHK(hipMalloc(&ticks, 128 * 8)); CHK(hipMalloc(&fail, 4)); CHK(hipMalloc(&xcc, 1024));
    unsigned base = 1; const int iters = 200;
    printf("us per HOP (median, max over pairs); pairs on the same XCD (b, b ^ 8) / different XCDs (b, b ^ 1); 8 or 128 pairs active at once\n");
    run<0, 1>(slots, ticks, fail, xcc, base, iters);
    {
        std::vector<unsigned> x(256); CHK(hipMemcpy(x.data(), xcc, 1024, hipMemcpyDeviceToHost));
        int same = 0, diff = 0; for (int b = 0; b < 256; b++) { same += x[b] == x[b ^ 8]; diff += x[b] != x[b ^ 1]; }
        printf("  placement: %d / 256 workgroups share the XCD with b ^ 8, %d / 256 differ from b ^ 1; xcc of blocks 0..9:", same, diff);
        for (int b = 0; b < 10; b++) printf(" %u", x[b]);
        printf("\n");
    }
    run<0, 0>(slots, ticks, fail, xcc, base, iters); run<0, 2>(slots, ticks, fail, xcc, base, iters); run<0, 3>(slots, ticks, fail, xcc, base, iters);
    run<0, 4>(slots, ticks, fail, xcc, base, iters); run<0, 5>(slots, ticks, fail, xcc, base, iters);
    run<1, 0>(slots, ticks, fail, xcc, base, iters); run<1, 1>(slots, ticks, fail, xcc, base, iters); run<1, 4>(slots, ticks, fail, xcc, base, iters);
    run<2, 0>(slots, ticks, fail, xcc, base, iters); run<2, 1>(slots, ticks, fail, xcc, base, iters); run<2, 2>(slots, ticks, fail, xcc, base, iters); run<2, 4>(slots, ticks, fail, xcc, base, iters);
    run<3, 1>(slots, ticks, fail, xcc, base, iters); run<3, 2>(slots, ticks, fail, xcc, base, iters);
    run<4, 0>(slots, ticks, fail, xcc, base, iters); run<4, 1>(slots, ticks, fail, xcc, base, iters); run<4, 3>(slots, ticks, fail, xcc, base, iters);
    return 0;
}
