// Microbenchmark (not product code), round 2: can the weight stream of decode operator i+1 be hidden behind operator i?
// A chain of OPS dependent "GEMV-like" operators (256 workgroups x 768 threads; every workgroup reads the whole 12 KB activation vector
// the previous operator wrote, multiplies it with its own slice of a weight matrix it streams from HBM once, writes 12 outputs):
//   mode 0  one stream, the kernel boundary is the dependency (what the product's decode graph does);
//   mode 1  two streams, operators alternate between them; a kernel loads its whole weight slice into registers FIRST, then waits for the
//           previous operator's arrival counter (256 arrivals), then reads x with sc1 loads; outputs are write-through (sc1) stores, drained,
//           then the workgroup arrives.  Operator i+1's launch, ramp and weight stream overlap operator i.
//   mode 2  as mode 1 on ONE stream (no overlap possible: prices the flag protocol alone).
// Every spin is bounded (timeout -> fail flag, kernel continues): this program cannot hang the box.
//   hipcc --offload-arch=gfx950 -O3 -o overlap_chain overlap_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED
#define AG __HIP_MEMORY_SCOPE_AGENT
constexpr int NWG = 256, NT = 768, XN = 3072, OUT_PER_WG = XN / NWG;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// next_w / pf_pages: "translation prefetch" -- one wave on each of 8 workgroups (one per XCD in practice) touches one 16-B word in every
// pf_stride bytes of the NEXT operator's weight matrix, so that operator's first loads find their page translations (and DRAM pages) warm.
template <int NL, bool FLAG>
__global__ __launch_bounds__(NT) void op_kernel(const uint4* __restrict__ w, const float* xin, float* xout, unsigned* cnt_prev, unsigned* cnt_mine,
                                                unsigned* fail, const uint4* next_w, int pf_pages, long pf_stride16) {
    __shared__ __attribute__((aligned(16))) float xs[XN];
    __shared__ float red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 wr[NL > 0 ? NL : 1];
    if (blockIdx.x >= NWG) {
        // L2 prefetch workgroup p = blockIdx - NWG (same blockIdx % 8 -> same XCD as compute workgroup p of the NEXT operator): plain loads of exactly
        // the slice that workgroup will read, PFL of its NL loads per thread; nothing is kept
        if (!next_w) return;
        const int pb = blockIdx.x - NWG;
        const uint4* np = next_w + (size_t)pb * NT * NL + tid;
        unsigned acc_ = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) if (i < pf_pages) acc_ ^= np[(size_t)i * NT].x;
        if (acc_ == 0x12345u) xout[0] = 1.f;       // (weights are 0x35 bytes: never true)
        return;
    }
    const uint4* wp = w + (size_t)blockIdx.x * NT * NL + tid;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        if (pf_stride16 < 0) wr[i] = wp[(size_t)i * NT];       // consumer with plain loads
        else { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(wp + (size_t)i * NT)); wr[i] = make_uint4(t.x, t.y, t.z, t.w); }
    }
    const unsigned pf_sink = 0;
    float4 v;
    if (FLAG) {
        if (cnt_prev) {
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(cnt_prev, RLX, AG) < (unsigned)NWG) { __builtin_amdgcn_s_sleep(1); if (++spins > (1u << 12)) { __hip_atomic_store(fail, 1u, RLX, AG); break; } }
            }
            __syncthreads();
        }
        // sc1 loads (producer stored sc1): no acquire fence needed
        const unsigned long long* xi = reinterpret_cast<const unsigned long long*>(xin);
        const unsigned long long a = __hip_atomic_load(xi + 2 * tid, RLX, AG), b = __hip_atomic_load(xi + 2 * tid + 1, RLX, AG);
        v = make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
    } else {
        v = reinterpret_cast<const float4*>(xin)[tid];
    }
    reinterpret_cast<float4*>(xs)[tid] = v;
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int b = (tid * 5 + i * 64) % (XN - 4);
        acc += (float)(wr[i].x & 0xFFu) * xs[b] + (float)(wr[i].y & 0xFFu) * xs[b + 1] + (float)(wr[i].z & 0xFFu) * xs[b + 2] + (float)(wr[i].w & 0xFFu) * xs[b + 3];
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid < OUT_PER_WG / 2) {               // 6 threads, 2 outputs each
        float s = 0.f;
        for (int k = 0; k < NT / 64; k++) s += red[k];
        const int j = blockIdx.x * OUT_PER_WG + 2 * tid;
        if (pf_sink == 0x12345u) s += 1.f;    // keeps the prefetch loads alive (the weights are 0x35.. bytes)
        const float o0 = 0.5f * xs[j] + 0.25f * (s * 1e-4f - floorf(s * 1e-4f)), o1 = 0.5f * xs[j + 1] + 0.125f * (s * 3e-4f - floorf(s * 3e-4f));
        if (FLAG) __hip_atomic_store(reinterpret_cast<unsigned long long*>(xout + j), ((unsigned long long)__float_as_uint(o1) << 32) | __float_as_uint(o0), RLX, AG);
        else { xout[j] = o0; xout[j + 1] = o1; }
    }
    if (FLAG && wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(cnt_mine, 1u, RLX, AG);
    }
}

struct Cfg { int mode, ops, reps; bool graph; int n_w; int pf_loads; bool plain; };

template <int NL>
static float run(const Cfg& c, const uint4* w, size_t w_stride, int n_w, float* xa, float* xb, unsigned* cnt, unsigned* fail, const std::vector<float>& x0, std::vector<float>* result) {
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t fork, join; CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    auto enqueue = [&]() {
        if (c.mode != 0) CHK(hipMemsetAsync(cnt, 0, (size_t)c.ops * 128, s1));
        if (c.mode == 1) { CHK(hipEventRecord(fork, s1)); CHK(hipStreamWaitEvent(s2, fork, 0)); }
        for (int i = 0; i < c.ops; i++) {
            hipStream_t s = (c.mode == 1 && (i & 1)) ? s2 : s1;
            const int nw = c.n_w ? c.n_w : n_w;
            const uint4* wi = w + (size_t)(i % nw) * w_stride;
            const uint4* wn = c.pf_loads ? w + (size_t)((i + 1) % nw) * w_stride : nullptr;
            const int grid = c.pf_loads ? 2 * NWG : NWG;
            const float* in = i & 1 ? xb : xa; float* out = i & 1 ? xa : xb;
            if (c.mode == 0) op_kernel<NL, false><<<grid, NT, 0, s>>>(wi, in, out, nullptr, nullptr, fail, wn, c.pf_loads, c.plain ? -1 : 0);
            else op_kernel<NL, true><<<grid, NT, 0, s>>>(wi, in, out, i ? cnt + (size_t)(i - 1) * 32 : nullptr, cnt + (size_t)i * 32, fail, wn, c.pf_loads, c.plain ? -1 : 0);
        }
        if (c.mode == 1) { CHK(hipEventRecord(join, s2)); CHK(hipStreamWaitEvent(s1, join, 0)); }
    };
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    if (c.graph) {
        CHK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
        enqueue();
        CHK(hipStreamEndCapture(s1, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    auto go = [&]() { if (c.graph) CHK(hipGraphLaunch(ge, s1)); else enqueue(); };
    CHK(hipMemcpy(xa, x0.data(), XN * 4, hipMemcpyHostToDevice)); CHK(hipMemset(xb, 0, XN * 4));
    go(); CHK(hipStreamSynchronize(s1));
    if (result) { result->resize(XN); CHK(hipMemcpy(result->data(), c.ops & 1 ? xb : xa, XN * 4, hipMemcpyDeviceToHost)); }
    unsigned f = 0; CHK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    if (f) { CHK(hipMemset(fail, 0, 4)); if (ge) { CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); } CHK(hipStreamDestroy(s1)); CHK(hipStreamDestroy(s2)); return -1.f; }   // spins timed out: not concurrent
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, s1)); for (int r = 0; r < c.reps; r++) go(); CHK(hipEventRecord(e1, s1)); CHK(hipStreamSynchronize(s1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ge) { CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g)); } CHK(hipStreamDestroy(s1)); CHK(hipStreamDestroy(s2));
    return ms * 1e3f / (c.reps * c.ops);
}

template <int NL>
static void sweep(const uint4* w, size_t w_stride, int n_w, float* xa, float* xb, unsigned* cnt, unsigned* fail, const std::vector<float>& x0) {
    const int ops = 130, reps = 20;
    std::vector<float> r0, r1, r2, r3;
    const float t0 = run<NL>({0, ops, reps, true, 0, 0, false}, w, w_stride, n_w, xa, xb, cnt, fail, x0, &r0);
    const float tpl = run<NL>({0, ops, reps, true, 0, 0, true}, w, w_stride, n_w, xa, xb, cnt, fail, x0, &r1);
    const float th = run<NL>({0, ops, reps, true, 1, 0, false}, w, w_stride, n_w, xa, xb, cnt, fail, x0, nullptr);
    const float thp = run<NL>({0, ops, reps, true, 1, 0, true}, w, w_stride, n_w, xa, xb, cnt, fail, x0, nullptr);
    const float tf = run<NL>({0, ops, reps, true, 0, NL, false}, w, w_stride, n_w, xa, xb, cnt, fail, x0, &r2);
    const float tfp = run<NL>({0, ops, reps, true, 0, NL, true}, w, w_stride, n_w, xa, xb, cnt, fail, x0, &r3);
    const float thalf = run<NL>({0, ops, reps, true, 0, (NL + 1) / 2, true}, w, w_stride, n_w, xa, xb, cnt, fail, x0, nullptr);
    const bool same = !memcmp(r0.data(), r1.data(), XN * 4) && !memcmp(r0.data(), r2.data(), XN * 4) && !memcmp(r0.data(), r3.data(), XN * 4);
    const double mb = (double)NWG * NT * NL * 16 / 1e6;
    printf("weights %5.1f MB per op [%.2f us at 5.5 TB/s]: 26 matrices cycled: nt loads %.2f us/op, plain loads %.2f | ONE matrix: nt %.2f, plain %.2f | cycled + NEXT matrix prefetched into L2 "
           "by 256 extra workgroups: consumer nt %.2f, consumer plain %.2f, first half only (plain) %.2f   (identical: %s)\n", mb, mb / 5.5, t0, tpl, th, thp, tf, tfp, thalf, same ? "yes" : "NO");
    fflush(stdout);
}

int main() {
    const int n_w = 26;                                    // distinct weight matrices cycled (HBM-cold like the 26 layers)
    const size_t w_stride = (size_t)NWG * NT * 10;         // uint4 per matrix (sized for the largest NL)
    uint4* w; CHK(hipMalloc(&w, w_stride * n_w * sizeof(uint4))); CHK(hipMemset(w, 0x35, w_stride * n_w * sizeof(uint4)));
    float *xa, *xb; CHK(hipMalloc(&xa, XN * 4)); CHK(hipMalloc(&xb, XN * 4));
    unsigned *cnt, *fail; CHK(hipMalloc(&cnt, 130 * 128)); CHK(hipMalloc(&fail, 128)); CHK(hipMemset(fail, 0, 128));
    std::vector<float> x0(XN); for (int i = 0; i < XN; i++) x0[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f;
    sweep<0>(w, w_stride, n_w, xa, xb, cnt, fail, x0);
    sweep<2>(w, w_stride, n_w, xa, xb, cnt, fail, x0);
    sweep<4>(w, w_stride, n_w, xa, xb, cnt, fail, x0);
    sweep<6>(w, w_stride, n_w, xa, xb, cnt, fail, x0);
    sweep<10>(w, w_stride, n_w, xa, xb, cnt, fail, x0);
    return 0;
}
