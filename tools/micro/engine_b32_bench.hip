// engine_b32_bench -- the TWO-GROUP launch of the batched decode-layer engine (csrc/vox_engine_b16.hip, launch_decode_engine_b16x2: 2 x 16 sequences, phase by phase) and
// its cache-slice indirection (EngBParams::kv_row) against the one-group launch, which engine_b16_bench checks against a CPU restatement.  Not product code; links the
// library's kernel objects (python voxtral-mini-realtime-rs_amd/build.py first).
//   engine_b32_bench [n_layers=2] [pos=100] [reps=20] [flags=2241] [tl_layer=-1]
// Three runs on the same synthetic weights / inputs / caches (32 sequences, positions pos - 3 i):
//   R1  two one-group launches, sequence i in cache slice i                      (the checked form)
//   R2  two one-group launches with kv_row = a permutation of the 32 slices     -> outputs and new cache rows bit-identical to R1
//   R3  ONE two-group launch, same permutation                                   -> bit-identical to R1; run-to-run bit-identical; time per launch against 2 x R1's
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

constexpr int D = 3072, NKV = 8, HD = 128, QD = 4096, KD = 1024, F = 9216, V = 131072, BM = 16, NSEQ = 32;

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
template <class T> static std::vector<T> d2h(const T* p, size_t n) { std::vector<T> v(n); CHK(hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost)); return v; }
struct Layer { Q4W wqkv, wo, w13, w2; float *attn_norm, *ffn_norm, *ada; };

int main(int argc, char** argv) {
    const int n_layers = argc > 1 ? atoi(argv[1]) : 2, pos0 = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 20, flags = argc > 4 ? atoi(argv[4]) : 2241, tl_layer = argc > 5 ? atoi(argv[5]) : -1;
    const int max_seq = pos0 < 248 ? 256 : ((pos0 + 8 + 63) / 64) * 64, window = 8192;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; n_layers %d pos %d reps %d flags %d; engine LDS %d (one group) / %d (two groups) bytes\n", prop.name, prop.multiProcessorCount, n_layers, pos0, reps, flags, engb_lds_bytes(), engb_lds_bytes2());
    { int occ = -1; hipError_t oe = engb_occupancy(&occ); printf("occupancy query (two-group kernel): %s, %d workgroup(s) per CU\n", hipGetErrorString(oe), occ); }
    hipStream_t s; CHK(hipStreamCreate(&s));
    std::vector<Layer> L(n_layers);
    const float sb = 0.004f;
    for (int l = 0; l < n_layers; l++) {
        L[l].wqkv = make_q4(QD + 2 * KD, D, 1000u + 16 * l, sb); L[l].wo = make_q4(D, QD, 1001u + 16 * l, sb);
        L[l].w13 = make_q4(2 * F, D, 1002u + 16 * l, sb); L[l].w2 = make_q4(D, F, 1003u + 16 * l, sb);
        L[l].attn_norm = make_f32(D, 2000u + l, 1.0f, 0.2f); L[l].ffn_norm = make_f32(D, 3000u + l, 1.0f, 0.2f); L[l].ada = make_f32(D, 4000u + l, 1.0f, 0.1f);
    }
    float* final_norm = make_f32(D, 5001u, 1.0f, 0.2f);
    float* h_in = make_f32((size_t)NSEQ * D, 5002u, 0.0f, 1.5f);
    const size_t seq_stride = (size_t)NKV * max_seq * HD, lf = (size_t)(NSEQ + 1) * seq_stride, slab = (size_t)n_layers * lf;      // [layer][slice 33][kv head][max_seq][hd]
    // content C_i of sequence i; slab_ref: slice i = C_i; slab_perm: slice perm[i] = C_i (slice 32: spare)
    std::vector<int> perm(NSEQ); for (int i = 0; i < NSEQ; i++) perm[i] = (i * 13 + 5) % 33;      // 13 and 33 coprime: injective into 0..32
    float* k_ref0 = make_f32(slab, 6001u, 0.0f, 1.0f); float* v_ref0 = make_f32(slab, 6002u, 0.0f, 1.0f);
    float* k_perm0 = dalloc<float>(slab); float* v_perm0 = dalloc<float>(slab);
    CHK(hipMemset(k_perm0, 0, slab * 4)); CHK(hipMemset(v_perm0, 0, slab * 4));
    for (int l = 0; l < n_layers; l++) for (int i = 0; i < NSEQ; i++) {
        CHK(hipMemcpy(k_perm0 + (size_t)l * lf + (size_t)perm[i] * seq_stride, k_ref0 + (size_t)l * lf + (size_t)i * seq_stride, seq_stride * 4, hipMemcpyDeviceToDevice));
        CHK(hipMemcpy(v_perm0 + (size_t)l * lf + (size_t)perm[i] * seq_stride, v_ref0 + (size_t)l * lf + (size_t)i * seq_stride, seq_stride * 4, hipMemcpyDeviceToDevice));
    }
    float* kc = dalloc<float>(slab); float* vc = dalloc<float>(slab);
    const int max_pos = 4096;
    std::vector<float> hc((size_t)max_pos * 64), hs((size_t)max_pos * 64);
    for (int p_ = 0; p_ < max_pos; p_++) for (int i = 0; i < 64; i++) { const double th = std::pow(1.0e6, -2.0 * i / 128.0); hc[(size_t)p_ * 64 + i] = (float)std::cos(p_ * th); hs[(size_t)p_ * 64 + i] = (float)std::sin(p_ * th); }
    float* rope_c = dalloc<float>(hc.size()); float* rope_s = dalloc<float>(hs.size());
    CHK(hipMemcpy(rope_c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(rope_s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> pos(NSEQ); for (int i = 0; i < NSEQ; i++) pos[i] = std::max(pos0 - 3 * i, 1);
    int* d_pos = dalloc<int>(NSEQ); CHK(hipMemcpy(d_pos, pos.data(), NSEQ * 4, hipMemcpyHostToDevice));
    int* d_perm = dalloc<int>(NSEQ); CHK(hipMemcpy(d_perm, perm.data(), NSEQ * 4, hipMemcpyHostToDevice));
    const float eps = 1e-5f;

    const size_t sbytes = eng_stream_bytes(n_layers, V);
    unsigned char* stream = dalloc<unsigned char>(sbytes); CHK(hipMemset(stream, 0, sbytes));
    for (int l = 0; l < n_layers; l++) {
        CHK(launch_eng_pack(L[l].wqkv, 0, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].wo, 1, l, n_layers, stream, V, s));
        CHK(launch_eng_pack(L[l].w13, 2, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].w2, 3, l, n_layers, stream, V, s));
    }
    unsigned char* stream_wo = dalloc<unsigned char>(engb_wo_stream_bytes(n_layers)); CHK(hipMemset(stream_wo, 0, engb_wo_stream_bytes(n_layers)));
    for (int l = 0; l < n_layers; l++) CHK(launch_eng_pack(L[l].wo, 5, l, n_layers, stream_wo, V, s));
    unsigned char* state[2];
    for (int q = 0; q < 2; q++) { state[q] = dalloc<unsigned char>(engb_state_bytes()); CHK(engb_state_init(state[q], s)); }
    CHK(hipStreamSynchronize(s));
    // layer tables: [0] slab base (kv_row form, and group A of R1), [1] group B of R1 (slices 16..)
    EngLayerTab* d_tab[2];
    for (int q = 0; q < 2; q++) {
        std::vector<EngLayerTab> tab(n_layers);
        for (int l = 0; l < n_layers; l++) tab[l] = EngLayerTab{L[l].attn_norm, L[l].ffn_norm, L[l].ada, kc + (size_t)l * lf + (size_t)q * BM * seq_stride, vc + (size_t)l * lf + (size_t)q * BM * seq_stride};
        d_tab[q] = dalloc<EngLayerTab>(n_layers); CHK(hipMemcpy(d_tab[q], tab.data(), n_layers * sizeof(EngLayerTab), hipMemcpyHostToDevice));
    }
    const size_t xf_u16 = (size_t)2 * (D / 128) * 256 * 8;
    uint16_t* xf_out[2]; float* ssq_out[2];
    for (int q = 0; q < 2; q++) { xf_out[q] = dalloc<uint16_t>(xf_u16); ssq_out[q] = dalloc<float>(256 * BM); }
    auto params = [&](int q, bool kvr) {
        EngBParams ep{}; ep.stream = stream; ep.stream_wo = stream_wo; ep.layers = kvr ? d_tab[0] : d_tab[q]; ep.n_layers = n_layers; ep.kv_seq_stride = (long)seq_stride;
        ep.h_in = h_in + (size_t)q * BM * D; ep.h_stride = D; ep.n_rows = BM; ep.final_norm = final_norm; ep.pos = d_pos + q * BM; ep.kv_row = kvr ? d_perm + q * BM : nullptr;
        ep.rope_cos = rope_c; ep.rope_sin = rope_s; ep.max_seq = max_seq; ep.window = window; ep.eps = eps;
        engb_state_carve(state[q], &ep); ep.xf_out = xf_out[q]; ep.ssq_out = ssq_out[q]; ep.tl = nullptr; ep.tl_layer = -1; ep.flags = flags;
        return ep;
    };
    auto check_err = [&](const char* when) { int bad = 0; for (int q = 0; q < 2; q++) { EngBParams ep{}; engb_state_carve(state[q], &ep); unsigned e[2]; CHK(hipMemcpy(e, ep.err, 8, hipMemcpyDeviceToHost));
        if (e[0]) { printf("ENGINE ERROR after %s (state %d): code %u, workgroup %u, tag bits %u\n", when, q, e[0] & 0xff, (e[0] >> 8) & 0xff, e[0] >> 16); bad = 1; }
        if (e[1]) printf("  note (%s): XCD group check failed (workgroup %u on xcc %u): plain-store edges off\n", when, (e[1] >> 8) & 0xff, e[1] >> 16); } return bad; };
    auto reset_caches = [&](bool permuted) { CHK(hipMemcpy(kc, permuted ? k_perm0 : k_ref0, slab * 4, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(vc, permuted ? v_perm0 : v_ref0, slab * 4, hipMemcpyDeviceToDevice)); };
    struct Out { std::vector<uint16_t> xf[2]; std::vector<float> ss[2]; std::vector<float> k, v; };
    auto grab = [&](bool permuted) {      // outputs + the caches in sequence order
        Out o; for (int q = 0; q < 2; q++) { o.xf[q] = d2h(xf_out[q], xf_u16); o.ss[q] = d2h(ssq_out[q], 256 * BM); }
        auto kk = d2h(kc, slab), vv = d2h(vc, slab); o.k.resize((size_t)n_layers * NSEQ * seq_stride); o.v.resize(o.k.size());
        for (int l = 0; l < n_layers; l++) for (int i = 0; i < NSEQ; i++) {
            const size_t src = (size_t)l * lf + (size_t)(permuted ? perm[i] : i) * seq_stride, dst = ((size_t)l * NSEQ + i) * seq_stride;
            memcpy(&o.k[dst], &kk[src], seq_stride * 4); memcpy(&o.v[dst], &vv[src], seq_stride * 4);
        }
        return o;
    };
    auto same = [&](const char* what, const Out& a, const Out& b) {
        bool ok = true;
        for (int q = 0; q < 2; q++) {
            const bool x = memcmp(a.xf[q].data(), b.xf[q].data(), xf_u16 * 2) == 0, y = memcmp(a.ss[q].data(), b.ss[q].data(), a.ss[q].size() * 4) == 0;
            size_t nd = 0; for (size_t i = 0; i < xf_u16; i++) nd += a.xf[q][i] != b.xf[q][i];
            printf("  %-52s group %c: XF planes %s (%zu of %zu words differ), sums of squares %s\n", what, 'A' + q, x ? "bit-identical" : "DIFFERENT", nd, xf_u16, y ? "bit-identical" : "DIFFERENT"); ok &= x && y;
        }
        size_t dk = 0, dv = 0; for (size_t i = 0; i < a.k.size(); i++) { dk += memcmp(&a.k[i], &b.k[i], 4) != 0; dv += memcmp(&a.v[i], &b.v[i], 4) != 0; }
        printf("  %-52s caches: %zu K words, %zu V words differ\n", what, dk, dv); ok &= dk == 0 && dv == 0;
        return ok;
    };
    bool all_ok = true;
    // ---- R1: one-group launches, identity slices
    reset_caches(false);
    for (int q = 0; q < 2; q++) CHK(launch_decode_engine_b16(params(q, false), s));
    if (hipStreamSynchronize(s) != hipSuccess || check_err("R1")) return 1;
    const Out r1 = grab(false);
    { size_t touched = 0; auto k0 = d2h(k_ref0, slab), kk = d2h(kc, slab); for (size_t i = 0; i < slab; i++) touched += memcmp(&k0[i], &kk[i], 4) != 0; printf("R1: K cache words changed: %zu (expected <= %d)\n", touched, n_layers * NSEQ * KD); }
    // ---- R2: one-group launches, permuted slices
    reset_caches(true);
    for (int q = 0; q < 2; q++) CHK(launch_decode_engine_b16(params(q, true), s));
    if (hipStreamSynchronize(s) != hipSuccess || check_err("R2")) return 1;
    const Out r2 = grab(true);
    all_ok &= same("one group per launch, kv_row permutation vs R1", r1, r2);
    // ---- R3: the two-group launch
    reset_caches(true);
    for (int q = 0; q < 2; q++) { CHK(hipMemset(xf_out[q], 0, xf_u16 * 2)); CHK(hipMemset(ssq_out[q], 0, 256 * BM * 4)); }
    CHK(launch_decode_engine_b16x2(params(0, true), params(1, true), s));
    { hipError_t se = hipStreamSynchronize(s); if (se != hipSuccess) { printf("two-group launch failed: %s\n", hipGetErrorString(se)); return 1; } }
    if (check_err("R3")) return 1;
    const Out r3 = grab(true);
    all_ok &= same("TWO groups in one launch vs R1", r1, r3);
    reset_caches(true);
    CHK(launch_decode_engine_b16x2(params(0, true), params(1, true), s)); CHK(hipStreamSynchronize(s));
    if (check_err("R3 again")) return 1;
    const Out r3b = grab(true);
    all_ok &= same("two-group launch, run to run", r3, r3b);
    // ---- timing
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms1, ms2;
    for (int i = 0; i < 3; i++) for (int q = 0; q < 2; q++) CHK(launch_decode_engine_b16(params(q, true), s));
    CHK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) for (int q = 0; q < 2; q++) CHK(launch_decode_engine_b16(params(q, true), s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms1, e0, e1));
    if (check_err("timing, one group per launch")) return 1;
    for (int i = 0; i < 3; i++) CHK(launch_decode_engine_b16x2(params(0, true), params(1, true), s));
    CHK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) CHK(launch_decode_engine_b16x2(params(0, true), params(1, true), s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    CHK(hipEventElapsedTime(&ms2, e0, e1));
    if (check_err("timing, two groups per launch")) return 1;
    const int nl = std::max(n_layers, 1);
    printf("32 sequences, %d layers: two one-group launches %.1f us (%.1f us per layer and group), ONE two-group launch %.1f us (%.1f us per layer for both groups) = %.2f x\n",
           n_layers, ms1 * 1000 / reps, ms1 * 1000 / reps / nl / 2, ms2 * 1000 / reps, ms2 * 1000 / reps / nl, ms1 / ms2);
    if (tl_layer >= 0) {      // both groups' stamps of one layer, merged and ordered by the median over the 256 CUs
        unsigned long long* tlbuf = dalloc<unsigned long long>(2 * 256 * 32); CHK(hipMemset(tlbuf, 0, 2 * 256 * 32 * 8));
        EngBParams pa = params(0, true), pb = params(1, true); pa.tl = tlbuf; pa.tl_layer = tl_layer;
        CHK(launch_decode_engine_b16x2(pa, pb, s)); CHK(hipStreamSynchronize(s));
        std::vector<unsigned long long> tb(2 * 256 * 32); CHK(hipMemcpy(tb.data(), tlbuf, tb.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int b = 0; b < 256; b++) if (tb[b * 32 + 19]) t0 = std::min(t0, tb[b * 32 + 19]);
        const char* names[32] = {"cons: A(q|k|v) in registers", "cons: q|k|v published (wave 0)", "cons: q,k,v of the sequences staged", "cons: attention published (wave 0)", "cons: A(wo) in registers", "cons: wo plane stored (wave 0)",
                                 "cons: A(w13) in registers", "cons: SwiGLU published (wave 0)", "comm: h all-gather: poll start", "comm: h all-gather: flags in", "comm: q|k|v granules swept", "comm: attention flags in",
                                 "comm: wo flag written", "comm: wo planes summed", "comm: h1 published", "cons: w2 plane stored (wave 0)", "load: layer's first packet issued", "load: layer's last packet issued", "load: stream done", "kernel start",
                                 "comm: h1 all-gather: poll start", "comm: h1 all-gather: flags in", "comm: SwiGLU flags in", "comm: w2: own waves through", "comm: w2 planes summed", "comm: h2 published", "cons: A(w2) in registers", "cons(w13): GEMM done (wave 0)", "cons(w13): barrier 1 passed", "cons(w13): SwiGLU in LDS (wave 0)", "cons(w13): barrier 2 passed", ""};
        struct Row { double lo, med, hi; int q, e; };
        std::vector<Row> rows;
        for (int q = 0; q < 2; q++) for (int e = 0; e < 31; e++) {
            if (e == 19 || e == 18) continue;
            std::vector<double> v; for (int b = 0; b < 256; b++) { const unsigned long long t = tb[(size_t)q * 8192 + b * 32 + e]; if (t) v.push_back((double)(t - t0) / 100.0); }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end()); rows.push_back(Row{v.front(), v[v.size() / 2], v.back(), q, e});
        }
        std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.med < b.med; });
        printf("timeline of layer %d, two-group launch (us since the first workgroup started; min / median / max over the 256 CUs; delta = median - previous row's):\n", tl_layer);
        double prev = rows.empty() ? 0 : rows[0].med;
        for (auto& r : rows) { printf("  %c  %-40s %9.2f %9.2f %9.2f   +%.2f\n", 'A' + r.q, names[r.e], r.lo, r.med, r.hi, r.med - prev); prev = r.med; }
        check_err("timeline launch");
    }
    printf("%s\n", all_ok ? "ALL OK" : "SOME CHECKS FAILED");
    return all_ok ? 0 : 2;
}
