// engine_bench -- stand-alone check + timing of the persistent decode-step engine (csrc/vox_engine.hip) against the per-operator kernels
// (csrc/vox_kernels.hip: the path the product used before, and keeps for other geometries).  Not product code; links the two object files.
//   engine_bench [n_layers=26] [pos=100] [reps=40] [tl_layer=-1] [flags=0] [pace_ticks=0] [rates=0] [loader_pace=-1] [time_flagged=0]
// Synthetic Q4 weights (random nibbles, f16 scales ~ N(0, 0.02) weights), random residual stream / KV cache.  Checks, in one run:
//   * LDS-DMA reaches LDS addresses >= 64 KiB (the ring needs it);
//   * v_cvt_pk_f32_fp8 of bytes 0..15 is q * 2^-9 (the consumer's nibble conversion);
//   * engine logits / argmax / new KV-cache rows vs the per-operator kernels; with n_layers == 1 also every intermediate edge (q|k|v, h1, act, h2);
//   * run-to-run bit-identical logits; time per step; optional per-phase timeline of one layer.
// Every spin in the engine is bounded: this program cannot hang the box.
#include "../../voxtral-mini-realtime-rs_amd/csrc/vox_kernels.h"
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace vox;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int D = 3072, NH = 32, NKV = 8, HD = 128, QD = 4096, KD = 1024, F = 9216, V = 131072;

__device__ __host__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void fill_u32(unsigned* p, size_t n, unsigned seed) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = hash32((unsigned)i * 2654435761u + seed); }
__global__ void fill_scale(uint16_t* p, size_t n, unsigned seed, float base) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float u = (hash32((unsigned)i * 40503u + seed) & 0xFFFF) / 65536.0f; p[i] = __half_as_ushort(__float2half(base * (0.5f + u))); }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float mean, float amp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float u = (hash32((unsigned)i * 69069u + seed) & 0xFFFFFF) / 16777216.0f; p[i] = mean + amp * (2.0f * u - 1.0f); }
}
template <class T> static T* dalloc(size_t n) { T* p; CHK(hipMalloc((void**)&p, n * sizeof(T))); return p; }
static Q4W make_q4(int N, int K, unsigned seed, float scale_base) {
    Q4W w{}; w.N = N; w.K = K; w.nb = K / 32; w.fmt = WFMT_Q4_0;
    const size_t nblk = (size_t)N * w.nb;
    uint4* qs = dalloc<uint4>(nblk); uint16_t* sc = dalloc<uint16_t>(nblk);
    fill_u32<<<(unsigned)((nblk * 4 + 255) / 256), 256>>>((unsigned*)qs, nblk * 4, seed);
    fill_scale<<<(unsigned)((nblk + 255) / 256), 256>>>(sc, nblk, seed ^ 0x9e3779b9u, scale_base);
    w.qs = qs; w.sc = sc; return w;
}
static float* make_f32(size_t n, unsigned seed, float mean, float amp) { float* p = dalloc<float>(n); fill_f32<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, mean, amp); return p; }

// ---- probes ----
__global__ __launch_bounds__(64) void dma_probe_kernel(const unsigned* src, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned offs[4] = {1024u, 60u * 1024, 100u * 1024, 150u * 1024};
    for (int t = 0; t < 4; t++) {
        unsigned keep; const unsigned voff = lane * 16;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)lds + offs[t]));
        const unsigned long long g = (unsigned long long)(src + 256 * t);
        const unsigned long long gs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(g >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)g);
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(dst), "s"(gs) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = 0; t < 4; t++) for (int i = lane; i < 256; i += 64) out[256 * t + i] = reinterpret_cast<unsigned*>(lds + offs[t])[i];
}
// the four-line form of the loader: one M0 write, instruction offsets 0 / 1024 / 2048 / 3072 -- do the offsets move the LDS address too?
__global__ __launch_bounds__(64) void dma4_probe_kernel(const unsigned* src, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    unsigned keep; const unsigned voff = lane * 16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)lds + 70u * 1024));
    const unsigned long long g = (unsigned long long)src;
    const unsigned long long gs = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(g >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)g);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:1024 nt\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:2048 nt\n\tglobal_load_lds_dwordx4 %1, %3 offset:3072 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(dst), "s"(gs) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = reinterpret_cast<unsigned*>(lds + 70 * 1024)[i];
}
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void fp8_probe_kernel(float* out) {
    const unsigned q = threadIdx.x & 15;
    const f2v a = __builtin_amdgcn_cvt_pk_f32_fp8((int)(q | ((15 - q) << 8) | (q << 16) | ((15 - q) << 24)), false);
    const f2v b = __builtin_amdgcn_cvt_pk_f32_fp8((int)(q | ((15 - q) << 8) | (q << 16) | ((15 - q) << 24)), true);
    if (threadIdx.x < 16) { out[4 * q] = a.x; out[4 * q + 1] = a.y; out[4 * q + 2] = b.x; out[4 * q + 3] = b.y; }
}

// ---- VALU issue rates of the consumer's instruction mix (one workgroup per CU; 1 or 2 waves per SIMD): ns per wave-instruction
template <int OP>
__global__ void rate_kernel(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b0 = 1.5f, b1 = 2.5f; unsigned w = threadIdx.x * 0x01010101u;
    typedef float f2r __attribute__((ext_vector_type(2)));
    f2r p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0}, q0 = {b0, b1};
    for (int i = 0; i < iters; i++) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q0));
        if (OP == 2) asm volatile("v_cvt_pk_f32_fp8_e32 %0, %4\n v_cvt_pk_f32_fp8_e32 %1, %4\n v_cvt_pk_f32_fp8_e32 %2, %4\n v_cvt_pk_f32_fp8_e32 %3, %4\n v_cvt_pk_f32_fp8_e32 %0, %4\n v_cvt_pk_f32_fp8_e32 %1, %4\n v_cvt_pk_f32_fp8_e32 %2, %4\n v_cvt_pk_f32_fp8_e32 %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w));
        if (OP == 3) asm volatile("v_cvt_pk_f32_fp8_sdwa %0, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %1, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %2, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %3, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %0, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %1, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %2, %4 src0_sel:WORD_1\n v_cvt_pk_f32_fp8_sdwa %3, %4 src0_sel:WORD_1" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(w));
        if (OP == 4) asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
        if (OP == 5) asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));
        if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(p0), "+v"(p1) : "v"(q0));   // two dependent chains (the consumer's block_dot)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.x + p2.x + p3.x + p0.y + p1.y;
}
static void run_rates() {
    float* out = dalloc<float>(256 * 1024);
    const char* names[7] = {"v_fma_f32 (4 chains)", "v_pk_fma_f32 (4 chains)", "v_cvt_pk_f32_fp8 e32", "v_cvt_pk_f32_fp8 sdwa", "v_cvt_f32_ubyteN", "v_and_b32", "v_pk_fma_f32 (2 chains)"};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int op = 0; op < 7; op++) for (int wv = 4; wv <= 8; wv += 4) {
        auto launch = [&]() {
            switch (op) { case 0: rate_kernel<0><<<256, 64 * wv>>>(out, iters); break; case 1: rate_kernel<1><<<256, 64 * wv>>>(out, iters); break; case 2: rate_kernel<2><<<256, 64 * wv>>>(out, iters); break;
                          case 3: rate_kernel<3><<<256, 64 * wv>>>(out, iters); break; case 4: rate_kernel<4><<<256, 64 * wv>>>(out, iters); break; case 5: rate_kernel<5><<<256, 64 * wv>>>(out, iters); break; default: rate_kernel<6><<<256, 64 * wv>>>(out, iters); }
        };
        launch(); CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("rate %-26s %d waves/SIMD: %.2f ns per wave-instruction (%.1f cycles at 2.4 GHz)\n", names[op], wv / 4, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4);
    }
}

struct Layer { Q4W wqkv, wo, w13, w2; float *attn_norm, *ffn_norm, *ada; };

static double maxabs(const std::vector<float>& a) { double m = 0; for (float v : a) m = std::max(m, (double)std::fabs(v)); return m; }
static double maxdiff(const std::vector<float>& a, const std::vector<float>& b) { double m = 0; for (size_t i = 0; i < a.size(); i++) { const double d = std::fabs((double)a[i] - b[i]); if (!(d <= m)) m = d; } return m; }
static std::vector<float> d2h(const float* p, size_t n) { std::vector<float> v(n); CHK(hipMemcpy(v.data(), p, n * 4, hipMemcpyDeviceToHost)); return v; }
static std::vector<float> granules(const unsigned long long* p, size_t n, unsigned* tag_min, unsigned* tag_max) {
    std::vector<unsigned long long> g(n); CHK(hipMemcpy(g.data(), p, n * 8, hipMemcpyDeviceToHost));
    std::vector<float> v(n); *tag_min = 0xffffffffu; *tag_max = 0;
    for (size_t i = 0; i < n; i++) { const unsigned lo = (unsigned)g[i], hi = (unsigned)(g[i] >> 32); memcpy(&v[i], &lo, 4); *tag_min = std::min(*tag_min, hi); *tag_max = std::max(*tag_max, hi); }
    return v;
}
static void report(const char* what, const std::vector<float>& ref, const std::vector<float>& got) {
    const double m = maxabs(ref), d = maxdiff(ref, got);
    printf("  %-34s max|ref| %.4g  max|diff| %.3g  rel %.3g  %s\n", what, m, d, d / (m > 0 ? m : 1), d <= 2e-4 * m ? "ok" : "MISMATCH");
}

int main(int argc, char** argv) {
    const int n_layers = argc > 1 ? atoi(argv[1]) : 26, pos = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 40, tl_layer = argc > 4 ? atoi(argv[4]) : -1, flags = argc > 5 ? atoi(argv[5]) : 0, pace = argc > 6 ? atoi(argv[6]) : 0;
    const int lpace = argc > 8 ? atoi(argv[8]) : -1;      // explicit loader pace (10-ns ticks between packet issues) next to an all-gather delay
    const int max_seq = pos < 248 ? 256 : ((pos + 8 + 63) / 64) * 64, window = 8192;
    if (max_seq > 1024) { printf("pos %d: the engine holds at most 1024 cache rows\n", pos); return 1; }
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("flags %d pace %d; ", flags, pace); printf("device %s, %d CUs; n_layers %d pos %d reps %d; engine LDS %d bytes, stream %.1f MB\n", prop.name, prop.multiProcessorCount, n_layers, pos, reps, eng_lds_bytes(),
           eng_stream_bytes(n_layers, V) / 1e6);
    hipStream_t s; CHK(hipStreamCreate(&s));

    // ---- probes
    {
        unsigned* src = dalloc<unsigned>(1024); unsigned* out = dalloc<unsigned>(1024);
        fill_u32<<<4, 256>>>(src, 1024, 77u);
        CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        dma_probe_kernel<<<1, 64, 160 * 1024>>>(src, out);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned> a(1024), b(1024); CHK(hipMemcpy(a.data(), src, 4096, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), out, 4096, hipMemcpyDeviceToHost));
        for (int t = 0; t < 4; t++) { int bad = 0; for (int i = 0; i < 256; i++) bad += a[256 * t + i] != b[256 * t + i]; printf("LDS-DMA probe, LDS offset %6d KiB: %s (%d / 256 words differ)\n", t == 0 ? 1 : t == 1 ? 60 : t == 2 ? 100 : 150, bad ? "FAIL" : "ok", bad); }
        {
            CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma4_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            CHK(hipMemset(out, 0, 4096)); dma4_probe_kernel<<<1, 64, 160 * 1024>>>(src, out); CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(b.data(), out, 4096, hipMemcpyDeviceToHost)); int bad4 = 0; for (int i = 0; i < 1024; i++) bad4 += a[i] != b[i];
            printf("LDS-DMA probe, four lines with instruction offsets 0..3072 behind one M0 write: %s (%d / 1024 words differ)\n", bad4 ? "FAIL" : "ok", bad4);
        }
        float* fo = dalloc<float>(64); fp8_probe_kernel<<<1, 64>>>(fo); CHK(hipDeviceSynchronize());
        auto f = d2h(fo, 64); int bad = 0;
        for (int q = 0; q < 16; q++) bad += f[4 * q] != q / 512.0f || f[4 * q + 1] != (15 - q) / 512.0f || f[4 * q + 2] != q / 512.0f || f[4 * q + 3] != (15 - q) / 512.0f;
        printf("fp8 (e4m3) nibble conversion q * 2^-9: %s (cvt(1) = %g)\n", bad ? "FAIL" : "ok", f[4]);
    }

    if (argc > 7 && atoi(argv[7])) run_rates();
    // ---- synthetic model
    std::vector<Layer> L(n_layers);
    const float sb = 0.004f;
    for (int l = 0; l < n_layers; l++) {
        L[l].wqkv = make_q4(QD + 2 * KD, D, 1000u + 16 * l, sb); L[l].wo = make_q4(D, QD, 1001u + 16 * l, sb);
        L[l].w13 = make_q4(2 * F, D, 1002u + 16 * l, sb); L[l].w2 = make_q4(D, F, 1003u + 16 * l, sb);
        L[l].attn_norm = make_f32(D, 2000u + l, 1.0f, 0.2f); L[l].ffn_norm = make_f32(D, 3000u + l, 1.0f, 0.2f); L[l].ada = make_f32(D, 4000u + l, 1.0f, 0.1f);
    }
    Q4W tok = make_q4(V, D, 5000u, sb);
    float* final_norm = make_f32(D, 5001u, 1.0f, 0.2f);
    float* h_in = make_f32(D, 5002u, 0.0f, 1.5f);
    const size_t lf = (size_t)NKV * max_seq * HD;
    float* kc_ref = make_f32((size_t)n_layers * lf, 6001u, 0.0f, 1.0f); float* vc_ref = make_f32((size_t)n_layers * lf, 6002u, 0.0f, 1.0f);
    float* kc_eng = dalloc<float>((size_t)n_layers * lf); float* vc_eng = dalloc<float>((size_t)n_layers * lf);
    CHK(hipMemcpy(kc_eng, kc_ref, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(vc_eng, vc_ref, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice));
    const int max_pos = 4096;
    std::vector<float> hc((size_t)max_pos * 64), hs((size_t)max_pos * 64);
    for (int p_ = 0; p_ < max_pos; p_++) for (int i = 0; i < 64; i++) { const double th = std::pow(1.0e6, -2.0 * i / 128.0); hc[(size_t)p_ * 64 + i] = (float)std::cos(p_ * th); hs[(size_t)p_ * 64 + i] = (float)std::sin(p_ * th); }
    float* rope_c = dalloc<float>(hc.size()); float* rope_s = dalloc<float>(hs.size());
    CHK(hipMemcpy(rope_c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(rope_s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    int* d_pos = dalloc<int>(1); CHK(hipMemcpy(d_pos, &pos, 4, hipMemcpyHostToDevice));
    CHK(hipDeviceSynchronize());

    // ---- reference: the per-operator kernels (five launches per layer)
    float* h = dalloc<float>(D); float* q = dalloc<float>(QD); float* att = dalloc<float>(QD); float* act = dalloc<float>(F); float* logits_ref = dalloc<float>(V);
    float* h1_keep = dalloc<float>(D);
    float* pv = dalloc<float>(4096); int* pi = dalloc<int>(4096);
    const float eps = 1e-5f;
    auto ref_step = [&](bool keep) {
        CHK(hipMemcpyAsync(h, h_in, D * 4, hipMemcpyDeviceToDevice, s));
        for (int l = 0; l < n_layers; l++) {
            float* kl = kc_ref + (size_t)l * lf; float* vl = vc_ref + (size_t)l * lf;
            GemvParams p{}; p.w = L[l].wqkv; p.x = h; p.x_stride = D; p.out = q; p.out_stride = QD; p.gamma = L[l].attn_norm; p.eps = eps;
            p.pos_ptr = d_pos; p.pos_off = 0; p.rope_cos = rope_c; p.rope_sin = rope_s; p.hd = HD; p.n_q = QD; p.n_k = KD; p.kcache = kl; p.vcache = vl; p.cache_head_stride = max_seq * HD;
            CHK(launch_q4_gemv(p, 1, PRO_RMS, EPI_ROPE_KV, q4_gemv_default_R(p.w.N, p.w.K, EPI_ROPE_KV), s));
            AttnParams ap{}; ap.q = q; ap.k = kl; ap.v = vl; ap.kv_row_stride = HD; ap.kv_head_stride = max_seq * HD; ap.out = att; ap.n_heads = NH; ap.n_kv_heads = NKV; ap.offset = 0; ap.window = window; ap.pos_ptr = d_pos; ap.M = 1;
            CHK(launch_attn_decode(ap, HD, max_seq, s));
            GemvParams o{}; o.w = L[l].wo; o.x = att; o.x_stride = QD; o.out = h; o.out_stride = D; o.resid = h; o.resid_stride = D;
            CHK(launch_q4_gemv(o, 1, PRO_NONE, EPI_RESID, q4_gemv_default_R(o.w.N, o.w.K, EPI_RESID), s));
            if (keep && l == 0) CHK(hipMemcpyAsync(h1_keep, h, D * 4, hipMemcpyDeviceToDevice, s));
            GemvParams f{}; f.w = L[l].w13; f.x = h; f.x_stride = D; f.out = act; f.out_stride = F; f.gamma = L[l].ffn_norm; f.mul = L[l].ada; f.eps = eps;
            CHK(launch_q4_gemv(f, 1, PRO_RMS_MUL, EPI_SWIGLU, q4_gemv_default_R(f.w.N, f.w.K, EPI_SWIGLU), s));
            GemvParams d{}; d.w = L[l].w2; d.x = act; d.x_stride = F; d.out = h; d.out_stride = D; d.resid = h; d.resid_stride = D;
            CHK(launch_q4_gemv(d, 1, PRO_NONE, EPI_RESID, q4_gemv_default_R(d.w.N, d.w.K, EPI_RESID), s));
        }
        GemvParams p{}; p.w = tok; p.x = h; p.x_stride = D; p.out = logits_ref; p.out_stride = V; p.gamma = final_norm; p.eps = eps; p.part_val = pv; p.part_idx = pi;
        CHK(launch_q4_gemv(p, 1, PRO_RMS, EPI_ARGMAX, q4_gemv_default_R(V, D, EPI_ARGMAX), s));
    };
    ref_step(true);
    CHK(hipStreamSynchronize(s));
    auto ref_logits = d2h(logits_ref, V), ref_h = d2h(h, D), ref_q = d2h(q, QD), ref_act = d2h(act, F), ref_h1 = d2h(h1_keep, D);
    int ref_arg = (int)(std::max_element(ref_logits.begin(), ref_logits.end()) - ref_logits.begin());
    {   // time the reference chain (eager launches; the product replays them from a hipGraph: a few % faster)
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        for (int i = 0; i < 3; i++) ref_step(false);
        CHK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) ref_step(false); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); printf("per-operator kernels (eager, 5 launches per layer): %.1f us per step\n", ms * 1000 / reps);
    }

    // ---- engine
    const size_t sbytes = eng_stream_bytes(n_layers, V);
    unsigned char* stream = dalloc<unsigned char>(sbytes); CHK(hipMemset(stream, 0, sbytes));
    for (int l = 0; l < n_layers; l++) {
        CHK(launch_eng_pack(L[l].wqkv, 0, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].wo, 1, l, n_layers, stream, V, s));
        CHK(launch_eng_pack(L[l].w13, 2, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].w2, 3, l, n_layers, stream, V, s));
    }
    CHK(launch_eng_pack(tok, 4, 0, n_layers, stream, V, s));
    unsigned char* state = dalloc<unsigned char>(eng_state_bytes()); CHK(hipMemset(state, 0, eng_state_bytes()));
    std::vector<EngLayerTab> tab(n_layers);
    for (int l = 0; l < n_layers; l++) tab[l] = EngLayerTab{L[l].attn_norm, L[l].ffn_norm, L[l].ada, kc_eng + (size_t)l * lf, vc_eng + (size_t)l * lf};
    EngLayerTab* d_tab = dalloc<EngLayerTab>(n_layers); CHK(hipMemcpy(d_tab, tab.data(), n_layers * sizeof(EngLayerTab), hipMemcpyHostToDevice));
    float* logits_eng = dalloc<float>(V); float* pv2 = dalloc<float>(256); int* pi2 = dalloc<int>(256);
    unsigned long long* tlbuf = dalloc<unsigned long long>(256 * 32); CHK(hipMemset(tlbuf, 0, 256 * 32 * 8));
    EngParams ep{}; ep.stream = stream; ep.cu_stride = sbytes / 256; ep.layers = d_tab; ep.n_layers = n_layers; ep.h_in = h_in; ep.final_norm = final_norm; ep.pos_ptr = d_pos; ep.pos_off = 0;
    ep.rope_cos = rope_c; ep.rope_sin = rope_s; ep.max_seq = max_seq; ep.window = window; ep.eps = eps; eng_state_carve(state, &ep);
    ep.part_val = pv2; ep.part_idx = pi2; ep.logits_out = logits_eng; ep.vocab = V; ep.tl = nullptr; ep.tl_layer = -1; ep.flags = flags; ep.pace_ticks = lpace >= 0 ? lpace : (flags & 512) ? 0 : pace; ep.ag_delay_ticks = (flags & 512) ? pace : 0;
    CHK(hipStreamSynchronize(s));
    auto check_err = [&](const char* when) { unsigned e; CHK(hipMemcpy(&e, ep.err, 4, hipMemcpyDeviceToHost)); if (e) printf("ENGINE ERROR after %s: code %u, workgroup %u, tag bits %u\n", when, e & 0xff, (e >> 8) & 0xff, e >> 16); return e; };
    CHK(launch_decode_engine(ep, s));
    hipError_t se = hipStreamSynchronize(s);
    if (se != hipSuccess) { printf("engine launch failed: %s\n", hipGetErrorString(se)); return 1; }
    check_err("first launch");
    auto eng_logits = d2h(logits_eng, V);
    std::vector<float> epv = d2h(pv2, 256); std::vector<int> epi(256); CHK(hipMemcpy(epi.data(), pi2, 1024, hipMemcpyDeviceToHost));
    int eng_arg = epi[0]; float eng_best = epv[0];
    for (int i = 1; i < 256; i++) if (epv[i] > eng_best || (epv[i] == eng_best && epi[i] < eng_arg)) { eng_best = epv[i]; eng_arg = epi[i]; }
    printf("engine vs per-operator kernels:\n");
    report("logits [131072]", ref_logits, eng_logits);
    printf("  argmax ref %d (%.6f)  engine %d (%.6f)  %s\n", ref_arg, ref_logits[ref_arg], eng_arg, eng_best, ref_arg == eng_arg ? "ok" : "DIFFERENT");
    {
        unsigned t0, t1;
        // the owners publish their rows ALREADY multiplied by the consumer's norm weight (the final norm * 512 behind the last layer): compare like with like
        auto g_h0 = granules(ep.H0, D, &t0, &t1); printf("  H0 tags %u..%u\n", t0, t1);
        { auto fn = d2h(final_norm, D); std::vector<float> exp_h(D); for (int i = 0; i < D; i++) exp_h[i] = ref_h[i] * fn[i] * 512.0f; report("layer-stack output h * final norm * 512", exp_h, g_h0); }
        std::vector<float> kr = d2h(kc_ref, (size_t)n_layers * lf), ke = d2h(kc_eng, (size_t)n_layers * lf), vr = d2h(vc_ref, (size_t)n_layers * lf), ve = d2h(vc_eng, (size_t)n_layers * lf);
        report("K cache (all rows, new row at pos)", kr, ke); report("V cache", vr, ve);
        if (n_layers == 1) {
            auto g = granules(ep.G, QD + 2 * KD, &t0, &t1); printf("  G tags %u..%u\n", t0, t1);
            std::vector<float> gq(g.begin(), g.begin() + QD); report("q (RoPE applied)", ref_q, gq);
            auto g_h1 = granules(ep.H1, D, &t0, &t1); printf("  H1 tags %u..%u\n", t0, t1);
            { auto fw = d2h(L[0].ffn_norm, D), ad = d2h(L[0].ada, D); std::vector<float> e1(D); for (int i = 0; i < D; i++) e1[i] = ref_h1[i] * (fw[i] * ad[i]); report("(h + wo(attention)) * ffn_norm * Ada", e1, g_h1); }
            auto g_a = granules(ep.A, F, &t0, &t1); printf("  A tags %u..%u\n", t0, t1); report("SwiGLU activations", ref_act, g_a);
        }
    }
    // ---- determinism + timing
    CHK(launch_decode_engine(ep, s)); CHK(hipStreamSynchronize(s));
    auto again = d2h(logits_eng, V);
    printf("  run-to-run: %s\n", memcmp(again.data(), eng_logits.data(), (size_t)V * 4) == 0 ? "bit-identical" : "DIFFERENT");
    check_err("second launch");
    EngParams e3_keep{}; int* d_pos2_keep = nullptr;
    {   // ---- flags 65536: the step input formed at launch start (argmax over the previous launch's partials + embedding + audio row) against the three-launch
        // form (engine, argmax_embed_kernel, engine): same token, same position bookkeeping, BIT-identical logits of the following step
        float* audio = make_f32((size_t)(max_seq + 8) * D, 7001u, 0.0f, 1.0f);
        int* d_tok = dalloc<int>(max_seq + 8); CHK(hipMemset(d_tok, 0, (max_seq + 8) * 4));
        int* d_pos2 = dalloc<int>(1); CHK(hipMemcpy(d_pos2, &pos, 4, hipMemcpyHostToDevice));
        float* h2 = dalloc<float>(D); float* pvk = dalloc<float>(256); int* pik = dalloc<int>(256);
        CHK(hipMemcpy(pvk, pv2, 1024, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(pik, pi2, 1024, hipMemcpyDeviceToDevice));      // partials of the step just run
        CHK(launch_argmax_embed(pv2, pi2, 256, d_tok, d_pos2, tok, audio, D, h2, s));
        EngParams e2 = ep; e2.h_in = h2; e2.pos_ptr = d_pos2; e2.logits_out = logits_eng;
        CHK(launch_decode_engine(e2, s)); CHK(hipStreamSynchronize(s));
        auto la = d2h(logits_eng, V); int tok_a, pos_a; CHK(hipMemcpy(&tok_a, d_tok + pos + 1, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&pos_a, d_pos2, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(pv2, pvk, 1024, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(pi2, pik, 1024, hipMemcpyDeviceToDevice));
        CHK(hipMemset(d_tok, 0, (max_seq + 8) * 4)); CHK(hipMemcpy(d_pos2, &pos, 4, hipMemcpyHostToDevice)); CHK(hipMemset(logits_eng, 0, (size_t)V * 4));
        EngParams e3 = ep; e3.h_in = nullptr; e3.pos_ptr = d_pos2; e3.pos_rw = d_pos2; e3.tokens = d_tok; e3.tok_qs = tok.qs; e3.tok_sc = tok.sc; e3.tok_nb = tok.nb; e3.audio = audio;
        e3.logits_out = logits_eng; e3.flags = ep.flags | 65536;
        CHK(launch_decode_engine(e3, s)); CHK(hipStreamSynchronize(s));
        auto lb = d2h(logits_eng, V); int tok_b, pos_b; CHK(hipMemcpy(&tok_b, d_tok + pos + 1, 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&pos_b, d_pos2, 4, hipMemcpyDeviceToHost));
        printf("  argmax at launch start: token %d vs %d, position %d vs %d, next step's logits %s (max|diff| %.3g)  %s\n", tok_b, tok_a, pos_b, pos_a,
               memcmp(la.data(), lb.data(), (size_t)V * 4) == 0 ? "bit-identical" : "DIFFERENT", maxdiff(la, lb), tok_a == tok_b && pos_a == pos_b && pos_b == pos + 1 && memcmp(la.data(), lb.data(), (size_t)V * 4) == 0 ? "ok" : "MISMATCH");
        check_err("argmax-at-start launches");
        if (argc > 9 && atoi(argv[9])) {      // time the flagged launch (position advances: a few hundred launches stay inside the cache)
            e3.logits_out = nullptr; CHK(hipMemcpy(d_pos2, &pos, 4, hipMemcpyHostToDevice));
            hipEvent_t a0, a1; CHK(hipEventCreate(&a0)); CHK(hipEventCreate(&a1));
            const int nrep = std::min(reps, max_seq - pos - 4);
            CHK(hipEventRecord(a0, s)); for (int i = 0; i < nrep; i++) CHK(launch_decode_engine(e3, s)); CHK(hipEventRecord(a1, s)); CHK(hipEventSynchronize(a1));
            float ms2; CHK(hipEventElapsedTime(&ms2, a0, a1)); printf("  flagged launches: %.1f us per step (%d launches, positions %d..)\n", ms2 * 1000 / nrep, nrep, pos + 1);
            check_err("flagged timing loop");
        }
        e3_keep = e3; e3_keep.logits_out = nullptr; d_pos2_keep = d_pos2;
        // the plain launches below run at `pos` again: the rows written at pos + 1 are never read by them
    }
    ep.logits_out = nullptr;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CHK(launch_decode_engine(ep, s));
    CHK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) CHK(launch_decode_engine(ep, s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n_layers * 65470464.0 + 226492416.0;
    printf("engine: %.1f us per step (%d layers + lm_head) = %.2f TB/s of Q4 bytes, %.3f of 8 TB/s\n", ms * 1000 / reps, n_layers, bytes / (ms * 1e-3 / reps) / 1e12, bytes / (ms * 1e-3 / reps) / 8e12);
    check_err("timing loop");
    if (tl_layer >= 0) {
        ep.tl = tlbuf; ep.tl_layer = tl_layer;
        if (argc > 9 && atoi(argv[9]) == 2) {      // timeline of a launch that forms its own input (flags 65536)
            CHK(hipMemcpy(d_pos2_keep, &pos, 4, hipMemcpyHostToDevice)); e3_keep.tl = tlbuf; e3_keep.tl_layer = tl_layer;
            CHK(launch_decode_engine(e3_keep, s)); CHK(hipStreamSynchronize(s)); printf("(timeline of a flags-65536 launch)\n");
        } else {
        CHK(launch_decode_engine(ep, s)); CHK(hipStreamSynchronize(s));
        }
        std::vector<unsigned long long> tb(256 * 32); CHK(hipMemcpy(tb.data(), tlbuf, tb.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int b = 0; b < 256; b++) if (tb[b * 32 + 19]) t0 = std::min(t0, tb[b * 32 + 19]);
        const char* names[32] = {"cons: x(q|k|v) staged", "cons: q|k|v done", "cons: q,k,v gathered", "cons: attention done", "cons: wo done", "cons: x(w13) staged", "cons: w1|w3 done", "cons: x(w2) staged",
                                 "comm: h gathered+staged", "comm: q|k|v gathered", "comm: wo partials summed, h1 out", "comm: h1 gathered+staged", "comm: act gathered", "comm: w2 partials summed, h2 out",
                                 "cons: w2 done", "cons: lm_head done", "load: layer's first packet issued", "load: layer's last packet issued", "load: stream done", "kernel start",
                                 "comm(h1 gather): start", "comm(h1 gather): probe ok", "comm(h1 gather): 52 granules in", "cons(attn): P.V partials written",
                                 "cons(w13): x in registers", "cons(attn): barrier 2 passed", "cons(attn): scores written", "cons(attn): barrier 1 passed", "cons: LAST wave published q|k|v", "cons: LAST wave published wo", "cons: LAST wave published w1|w3", "cons: LAST wave published w2"};
        const int order[32] = {19, 16, 8, 0, 1, 28, 9, 2, 26, 27, 23, 25, 3, 4, 29, 10, 20, 21, 22, 11, 5, 24, 6, 30, 12, 7, 14, 31, 13, 17, 18, 15};
        printf("timeline of layer %d (us since the first workgroup started; min / median / max over the 256 CUs):\n", tl_layer);
        for (int oi = 0; oi < 32; oi++) {
            const int e = order[oi]; std::vector<double> v;
            for (int b = 0; b < 256; b++) if (tb[b * 32 + e]) v.push_back((double)(tb[b * 32 + e] - t0) / 100.0);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  %-36s %9.2f %9.2f %9.2f\n", names[e], v.front(), v[v.size() / 2], v.back());
        }
        for (int e : {1, 6, 14}) {      // where do the tails come from?  slowest CUs and per-XCD medians of three compute-phase ends
            std::vector<std::pair<double, int>> v; for (int b = 0; b < 256; b++) if (tb[b * 32 + e]) v.push_back({(double)(tb[b * 32 + e] - t0) / 100.0, b});
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  tail of '%s': slowest", names[e]); for (int i = 0; i < 8; i++) printf(" cu%d(x%d,j%d)=%.2f", v[v.size() - 1 - i].second, v[v.size() - 1 - i].second & 7, v[v.size() - 1 - i].second >> 3, v[v.size() - 1 - i].first);
            printf("\n    per-XCD median:");
            for (int x = 0; x < 8; x++) { std::vector<double> m; for (auto& pr : v) if ((pr.second & 7) == x) m.push_back(pr.first); printf(" %.2f", m[m.size() / 2]); }
            printf("\n");
        }
        {   // duration of w1|w3 (x staged -> done) by slice index j: median over the 8 groups
            printf("  w1|w3 duration by j (median over groups):");
            for (int j = 0; j < 32; j++) { std::vector<double> m; for (int x = 0; x < 8; x++) { const int b = 8 * j + x; if (tb[b * 32 + 6] && tb[b * 32 + 5]) m.push_back((double)(tb[b * 32 + 6] - tb[b * 32 + 5]) / 100.0); } std::sort(m.begin(), m.end()); if (!m.empty()) printf(" %.1f", m[m.size() / 2]); }
            printf("\n  q|k|v duration by j:");
            for (int j = 0; j < 32; j++) { std::vector<double> m; for (int x = 0; x < 8; x++) { const int b = 8 * j + x; if (tb[b * 32 + 1] && tb[b * 32 + 0]) m.push_back((double)(tb[b * 32 + 1] - tb[b * 32 + 0]) / 100.0); } std::sort(m.begin(), m.end()); if (!m.empty()) printf(" %.1f", m[m.size() / 2]); }
            printf("\n");
            unsigned e1; CHK(hipMemcpy(&e1, ep.err + 1, 4, hipMemcpyDeviceToHost)); if (e1) printf("  note: XCD group check failed somewhere (workgroup %u on xcc %u): fast edges off\n", (e1 >> 8) & 0xff, e1 >> 16);
        }
        check_err("timeline launch");
    }
    return 0;
}
