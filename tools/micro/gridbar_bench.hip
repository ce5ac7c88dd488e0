// Microbenchmark (not product code): cost of a device-wide barrier between co-resident workgroups on MI355X, to decide whether a
// persistent decode-layer kernel (5 phases per layer separated by grid barriers) could beat 5 graph-launched kernels.
// Variants: (a) atomic counter barrier with device-scope fences; (b) the same plus a 12 KB "activation" vector written by a few
// workgroups before and read by all workgroups after each barrier (what a phase boundary really needs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* cnt, unsigned target, unsigned* fail) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // release: my workgroup's writes visible device-wide
        atomicAdd(cnt, 1u);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *fail = 1; break; }   // bounded: never hang the box
        }
        __threadfence();                                   // acquire
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(256) void bar_kernel(unsigned* cnt, unsigned* fail, int iters, float* vec, int vec_n, float* sink) {
    const unsigned nwg = gridDim.x;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        if (vec) {   // 16 producer workgroups write the vector (like a GEMV's 3072 outputs spread over workgroups)
            if (blockIdx.x < 16) for (int i = threadIdx.x + blockIdx.x * 256; i < vec_n; i += 16 * 256) vec[i] = (float)(it + i);
        }
        grid_barrier(cnt, (unsigned)(it + 1) * nwg, fail);
        if (vec) { for (int i = threadIdx.x; i < vec_n; i += 256) acc += __builtin_nontemporal_load(vec + i); }
    }
    if (sink && acc == 123.456f) sink[0] = acc;
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 12345) p[0] = 1.f; }

int main() {
    unsigned *cnt, *fail; float *vec, *sink;
    CHK(hipMalloc(&cnt, 4)); CHK(hipMalloc(&fail, 4)); CHK(hipMalloc(&vec, 3072 * 4)); CHK(hipMalloc(&sink, 4));
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int nwg : {256, 512, 768}) {
        for (int with_vec = 0; with_vec < 2; with_vec++) {
            CHK(hipMemsetAsync(cnt, 0, 4, s)); CHK(hipMemsetAsync(fail, 0, 4, s));
            bar_kernel<<<nwg, 256, 0, s>>>(cnt, fail, 10, with_vec ? vec : nullptr, 3072, sink);      // warm-up
            CHK(hipMemsetAsync(cnt, 0, 4, s));
            CHK(hipEventRecord(e0, s));
            bar_kernel<<<nwg, 256, 0, s>>>(cnt, fail, iters, with_vec ? vec : nullptr, 3072, sink);
            CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned f; CHK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
            printf("grid barrier: %d WGs x 256 thr, %s: %.2f us per barrier%s\n", nwg, with_vec ? "with 12 KB vector handoff" : "barrier only", ms * 1e3 / iters, f ? "  (TIMEOUT!)" : "");
        }
    }
    // reference: back-to-back empty kernels in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 130; i++) empty_kernel<<<768, 256, 0, s>>>(nullptr);
    CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
    CHK(hipEventRecord(e0, s)); for (int r = 0; r < 20; r++) CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph of 130 empty kernels (768 WGs): %.2f us per kernel\n", ms * 1e3 / (20 * 130));
    return 0;
}
