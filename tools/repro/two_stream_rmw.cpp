// Stand-alone check (no library code): two host threads, two HIP streams, each a private buffer and the sequence
//   write (every element = f(iteration, index))  ->  in-place read-modify-write (grid-stride, 8-byte pairs, like rope_kernel)  ->  count elements != expected.
// Run one thread at a time, then both at once.  Any mismatch is a platform (runtime / cache-coherence) matter, not a kernel race: the kernels share nothing.
//   hipcc --offload-arch=gfx950 -O2 -o two_stream_rmw tools/repro/two_stream_rmw.cpp -lpthread && ./two_stream_rmw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ unsigned f(int it, long i) { return (unsigned)i * 2654435761u + (unsigned)it * 40503u; }
__global__ __launch_bounds__(256) void write_k(unsigned* buf, long n, int it) { for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) buf[i] = f(it, i); }
__global__ void rmw_k(unsigned* __restrict__ buf, long pairs, const unsigned* __restrict__ cs) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        const unsigned c = cs[2 * (i & 1023)], s = cs[2 * (i & 1023) + 1]; unsigned* p = buf + 2 * i; const unsigned a = p[0], b = p[1]; p[0] = a * c - b * s; p[1] = a * s + b * c; }
}
__global__ __launch_bounds__(256) void check_k(const unsigned* buf, long pairs, const unsigned* cs, int it, unsigned* bad) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (long)gridDim.x * 256) {
        const unsigned c = cs[2 * (i & 1023)], s = cs[2 * (i & 1023) + 1]; const unsigned a = f(it, 2 * i), b = f(it, 2 * i + 1);
        if (buf[2 * i] != a * c - b * s || buf[2 * i + 1] != a * s + b * c) atomicAdd(bad, 1u); }
}
struct Side { hipStream_t s; unsigned* buf; unsigned* cs; unsigned* bad; long n; };
static unsigned run(Side& sd, int iters, int base) {
    for (int it = 0; it < iters; it++) {
        write_k<<<dim3(1024), dim3(256), 0, sd.s>>>(sd.buf, sd.n, base + it);
        rmw_k<<<dim3(2048), dim3(256), 0, sd.s>>>(sd.buf, sd.n / 2, sd.cs);
        check_k<<<dim3(1024), dim3(256), 0, sd.s>>>(sd.buf, sd.n / 2, sd.cs, base + it, sd.bad);
    }
    CK(hipStreamSynchronize(sd.s)); unsigned h = 0; CK(hipMemcpy(&h, sd.bad, 4, hipMemcpyDeviceToHost)); CK(hipMemset(sd.bad, 0, 4)); return h;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400; const long n = 484L * 6144;
    Side sd[2];
    for (auto& x : sd) { CK(hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking)); x.n = n; CK(hipMalloc(&x.buf, n * 4)); CK(hipMalloc(&x.cs, 2048 * 4)); CK(hipMalloc(&x.bad, 4)); CK(hipMemset(x.bad, 0, 4));
        std::vector<unsigned> h(2048); for (int i = 0; i < 2048; i++) h[i] = 1664525u * (unsigned)i + 1013904223u; CK(hipMemcpy(x.cs, h.data(), 2048 * 4, hipMemcpyHostToDevice)); }
    printf("one at a time: %u %u mismatching pairs\n", run(sd[0], iters, 0), run(sd[1], iters, 1000));
    for (int rep = 0; rep < 3; rep++) { unsigned r[2]; std::thread t0([&] { r[0] = run(sd[0], iters, 2000); }), t1([&] { r[1] = run(sd[1], iters, 3000); }); t0.join(); t1.join();
        printf("both at once:  %u %u mismatching pairs (of %ld x %d)\n", r[0], r[1], n / 2, iters); }
    return 0;
}
