// Microbenchmark (not product code), round 2: cost of combining a 32-way K split with order-independent (fixed-point int64) device-scope atomics.
// 256 workgroups x 256 threads; workgroup (h = b / 8, slice = b % 8) adds 384 values (rows slice*384 ..) into acc[3072] -- every address receives
// 32 adds per launch, from 32 different workgroups -- against the same launch writing parts[h][row] with plain stores.  In a replayed graph,
// preceded by a dependent "touch" kernel so the launch is not back-to-back with itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int MODE>   // 0 plain stores to parts, 1 u64 atomics (agent scope), 2 f32 atomics
__global__ __launch_bounds__(256) void k(const float* in, float* parts, unsigned long long* acc, float* facc) {
    const int tid = threadIdx.x, h = blockIdx.x / 8, slice = blockIdx.x % 8, c = tid & 15;
    const float v0 = in[(blockIdx.x * 7 + tid) % 3072];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int row = slice * 384 + (tid >> 4) + 16 * i;
        const float v = v0 * (float)(i + 1) + (float)h * 1e-3f;
        if (c == 0) {
            if (MODE == 0) parts[(size_t)h * 3072 + row] = v;
            else if (MODE == 1) __hip_atomic_fetch_add(acc + row, (unsigned long long)(long long)(v * 4294967296.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(facc + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ void consume(const unsigned long long* acc, const float* parts, float* out, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3072) return;
    float s = 0.f;
    if (mode == 0) { for (int h = 0; h < 32; h++) s += parts[(size_t)h * 3072 + i]; }
    else s = (float)((double)(long long)acc[i] * (1.0 / 4294967296.0));
    out[i] = s * 1e-3f + 0.5f;
}
template <int MODE>
static float run(float* a, float* parts, unsigned long long* acc, float* facc, int reps) {
    hipStream_t s; CHK(hipStreamCreate(&s));
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; i++) {
        k<MODE><<<256, 256, 0, s>>>(a, parts, acc, facc);
        consume<<<12, 256, 0, s>>>(acc, parts, a, MODE);
    }
    CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, s)); for (int r = 0; r < reps; r++) CHK(hipGraphLaunch(ge, s)); CHK(hipEventRecord(e1, s)); CHK(hipStreamSynchronize(s));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * 100);
}
int main() {
    float *a, *parts, *facc; unsigned long long* acc;
    CHK(hipMalloc(&a, 3072 * 4)); CHK(hipMalloc(&parts, 32 * 3072 * 4)); CHK(hipMalloc(&acc, 3072 * 8)); CHK(hipMalloc(&facc, 3072 * 4));
    CHK(hipMemset(a, 0, 3072 * 4)); CHK(hipMemset(acc, 0, 3072 * 8)); CHK(hipMemset(facc, 0, 3072 * 4));
    const float t0 = run<0>(a, parts, acc, facc, 20), t1 = run<1>(a, parts, acc, facc, 20), t2 = run<2>(a, parts, acc, facc, 20);
    printf("producer + 12-workgroup consumer pair, us per pair: plain stores to parts[32][3072] %.2f | int64 fixed-point atomics (32 adds per address) %.2f | f32 atomics %.2f\n", t0, t1, t2);
    return 0;
}
