// Microbenchmark (not product code), round 2.  Three questions behind the single-stream decode step (130 dependent launches):
//  (1) what does ONE dependent kernel cost in a replayed hipGraph when it does what every decode GEMV must do before any math --
//      read the 12 KB activation vector the previous kernel just wrote (another XCD's L2), stage it in LDS, write a few outputs --
//      for the launch geometries we can choose (768 x 256 thr, 256 x 1024 thr, 256 x 256, 32 x 256);
//  (2) the XCD-hierarchical grid barrier of MI355X_MICROARCH.md ("barrier-xcd": per-group counter, group leader -> top counter ->
//      per-group generation word, relaxed sc1 polls + s_sleep, one release before the arrive and one acquire after the release) against
//      the single-counter barrier round 1 measured (tools/micro/gridbar_bench.hip);
//  (3) issue rates of the VALU instructions a Q4 dot product can be built from (cycles per wave-instruction, one wave per SIMD and four).
// Every spin is bounded (a timeout sets a flag and the kernel exits): this program cannot hang the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- (1) dependent chain --------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(T) void touch_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
    extern __shared__ float xs[];
    float s = 0.f;
    for (int i = threadIdx.x * 4; i < n; i += T * 4) { const float4 v = *reinterpret_cast<const float4*>(in + i); *reinterpret_cast<float4*>(xs + i) = v; s += v.x; }
    __syncthreads();
    // every wave produces a couple of outputs from the staged vector (like one GEMV row group), all of out[] is covered by the grid
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = T / 64, gw = blockIdx.x * nw + wave, total = gridDim.x * nw;
    for (int o = gw; o < n; o += total) { const float v = xs[(o * 7 + lane) % n] + s * 1e-9f; if (lane == 0) out[o] = v * 0.5f; }
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 12345) p[0] = 1.f; }

template <int T>
static float chain_us(hipStream_t s, int grid, float* a, float* b, int n, int launches, int reps) {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < launches; i++) { if (grid > 0) touch_kernel<T><<<grid, T, n * 4, s>>>(i & 1 ? b : a, i & 1 ? a : b, n); else empty_kernel<<<768, 256, 0, s>>>(nullptr); }
    CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, s)); for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    return ms * 1e3f / (reps * launches);
}

// ---- (2) barriers ---------------------------------------------------------------------------------------------------------
struct BarState { unsigned grp_cnt[8][32]; unsigned top[32]; unsigned grp_gen[8][32]; unsigned flat[32]; unsigned fail[32]; };   // 128 B apart
#define RLX __ATOMIC_RELAXED
#define AG __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void spin_until_ge(unsigned* w, unsigned target, unsigned* fail) {
    unsigned spins = 0;
    while (__hip_atomic_load(w, RLX, AG) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 21)) { __hip_atomic_store(fail, 1u, RLX, AG); break; } }
}
// XCD-hierarchical: group = blockIdx & 7 (== the XCD a block lands on in practice; correctness does not depend on it)
__device__ __forceinline__ void barrier_xcd(BarState* b, unsigned epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned grp = blockIdx.x & 7, n_grp = (gridDim.x + 7 - grp) >> 3, n_groups = gridDim.x < 8 ? gridDim.x : 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(&b->grp_cnt[grp][0], 1u, RLX, AG);
        if (old == epoch * n_grp - 1) {                                     // group leader = its last arriver
            const unsigned o2 = __hip_atomic_fetch_add(&b->top[0], 1u, RLX, AG);
            if (o2 == epoch * n_groups - 1) { for (unsigned g = 0; g < n_groups; g++) __hip_atomic_store(&b->grp_gen[g][0], epoch, RLX, AG); }
        }
        spin_until_ge(&b->grp_gen[grp][0], epoch, &b->fail[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ void barrier_flat(BarState* b, unsigned epoch) {   // the single-counter form (round 1), release / acquire fences as above
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&b->flat[0], 1u, RLX, AG);
        spin_until_ge(&b->flat[0], epoch * gridDim.x, &b->fail[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <int KIND>
__global__ __launch_bounds__(256) void bar_kernel(BarState* b, int iters, float* vec, int vec_n, float* sink) {
    float acc = 0.f; unsigned ep = 0;
    for (int it = 0; it < iters; it++) {
        if (__hip_atomic_load(&b->fail[0], RLX, AG)) break;      // a timeout anywhere ends the run quickly
        if (vec) {          // 16 producer workgroups write the vector, all workgroups read it after the barrier (a GEMV -> GEMV seam)
            if (blockIdx.x < 16) for (int i = threadIdx.x + blockIdx.x * 256; i < vec_n; i += 16 * 256) vec[i] = (float)(it + i);
        }
        if (KIND == 0) barrier_xcd(b, ++ep); else barrier_flat(b, ++ep);
        if (vec) { for (int i = threadIdx.x; i < vec_n; i += 256) { const float v = vec[i]; acc += v; if (v != (float)(it + i)) b->fail[1] = 1u; } }
        if (vec) { if (KIND == 0) barrier_xcd(b, ++ep); else barrier_flat(b, ++ep); }   // readers done before the next write (two barriers per iteration, both counted)
    }
    if (sink && acc == 123.456f) sink[0] = acc;
}

// ---- (3) VALU issue rates ---------------------------------------------------------------------------------------------------
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned* out, unsigned long long* cyc, int iters) {
    unsigned a[8], b = threadIdx.x * 2654435761u + 12345u, c = 0x43434343u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x + i * 7919u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 1) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(b));
                else if (OP == 2) asm volatile("v_dot8_i32_i4 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 3) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 4) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 5) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 6) asm volatile("v_perm_b32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (OP == 7) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP>
__global__ __launch_bounds__(256) void rate_pk_kernel(float* out, unsigned long long* cyc, int iters) {   // v_pk_fma_f32 needs 64-bit operands
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (f2){(float)threadIdx.x + i, (float)i};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    hipStream_t s; CHK(hipStreamCreate(&s));
    const int n = 3072;
    float *a, *b; CHK(hipMalloc(&a, n * 4)); CHK(hipMalloc(&b, n * 4)); CHK(hipMemset(a, 0, n * 4)); CHK(hipMemset(b, 0, n * 4));
    printf("== (1) dependent kernel chain in a replayed hipGraph (130 launches, 12 KB vector handed from kernel to kernel), us per kernel\n");
    printf("empty 768x256                 : %.2f\n", chain_us<256>(s, 0, a, b, n, 130, 30));
    printf("touch  768 WG x 256 thr       : %.2f\n", chain_us<256>(s, 768, a, b, n, 130, 30));
    printf("touch  256 WG x 256 thr       : %.2f\n", chain_us<256>(s, 256, a, b, n, 130, 30));
    printf("touch  256 WG x 1024 thr      : %.2f\n", chain_us<1024>(s, 256, a, b, n, 130, 30));
    printf("touch  512 WG x 512 thr       : %.2f\n", chain_us<512>(s, 512, a, b, n, 130, 30));
    printf("touch   32 WG x 256 thr       : %.2f\n", chain_us<256>(s, 32, a, b, n, 130, 30));
    printf("touch    8 WG x 256 thr       : %.2f\n", chain_us<256>(s, 8, a, b, n, 130, 30));

    printf("== (2) grid barriers, persistent kernel, 256-thread workgroups, us per barrier\n");
    BarState* bs; CHK(hipMalloc(&bs, sizeof(BarState))); float* sink; CHK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 1000;
    for (int kind = 0; kind < 2; kind++)
        for (int nwg : {256, 512, 768}) {
            for (int with_vec = 0; with_vec < 2; with_vec++) {
                if (kind == 1 && with_vec) continue;
                float ms = 0.f;
                for (int pass = 0; pass < 2; pass++) {       // pass 0 = warm-up
                    CHK(hipMemsetAsync(bs, 0, sizeof(BarState), s));
                    CHK(hipEventRecord(e0, s));
                    if (kind == 0) bar_kernel<0><<<nwg, 256, 0, s>>>(bs, pass ? iters : 10, with_vec ? a : nullptr, n, sink);
                    else bar_kernel<1><<<nwg, 256, 0, s>>>(bs, pass ? iters : 10, nullptr, n, sink);
                    CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                }
                BarState h; CHK(hipMemcpy(&h, bs, sizeof h, hipMemcpyDeviceToHost));
                const int nbar = with_vec ? 2 * iters : iters;
                printf("%-5s %4d WGs %-26s: %.2f us per barrier%s%s\n", kind == 0 ? "xcd" : "flat", nwg, with_vec ? "+ 12 KB vector hand-off" : "barrier only", ms * 1e3 / nbar,
                       h.fail[0] ? "  (TIMEOUT!)" : "", h.fail[1] ? "  (STALE DATA!)" : "");
            }
        }

    printf("== (3) VALU issue rates: cycles per wave-instruction (s_memtime), 256 workgroups; 1 wave / SIMD (256 thr) and 4 waves / SIMD (4 x 256 thr per CU)\n");
    unsigned* out; unsigned long long* cyc; CHK(hipMalloc(&out, 1024 * 256 * 4)); CHK(hipMalloc(&cyc, 1024 * 4 * 8));
    const char* names[8] = {"v_fma_f32", "v_cvt_f32_ubyte1", "v_dot8_i32_i4", "v_dot4_i32_i8", "v_dot2c_f32_bf16", "v_and_or_b32", "v_perm_b32", "v_xor_b32"};
    const int rit = 2000;
    for (int op = 0; op < 9; op++) {
        for (int occ = 0; occ < 2; occ++) {
            const int grid = occ ? 1024 : 256;
            for (int pass = 0; pass < 2; pass++) {
                switch (op) {
                case 0: rate_kernel<0><<<grid, 256, 0, s>>>(out, cyc, rit); break; case 1: rate_kernel<1><<<grid, 256, 0, s>>>(out, cyc, rit); break;
                case 2: rate_kernel<2><<<grid, 256, 0, s>>>(out, cyc, rit); break; case 3: rate_kernel<3><<<grid, 256, 0, s>>>(out, cyc, rit); break;
                case 4: rate_kernel<4><<<grid, 256, 0, s>>>(out, cyc, rit); break; case 5: rate_kernel<5><<<grid, 256, 0, s>>>(out, cyc, rit); break;
                case 6: rate_kernel<6><<<grid, 256, 0, s>>>(out, cyc, rit); break; case 7: rate_kernel<7><<<grid, 256, 0, s>>>(out, cyc, rit); break;
                default: rate_pk_kernel<0><<<grid, 256, 0, s>>>((float*)out, cyc, rit); break;
                }
                CHK(hipStreamSynchronize(s));
            }
            std::vector<unsigned long long> h(grid * 4); CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            double sum = 0; for (auto v : h) sum += (double)v;
            printf("%-18s %s: %.2f cycles per instruction per wave\n", op < 8 ? names[op] : "v_pk_fma_f32", occ ? "4 waves/SIMD" : "1 wave/SIMD ", sum / h.size() / (rit * 32.0));
        }
    }
    return 0;
}
