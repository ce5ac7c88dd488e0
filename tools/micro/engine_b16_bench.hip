// engine_b16_bench -- stand-alone check + timing of the batched (<= 16 sequences) decode-layer engine (csrc/vox_engine_b16.hip) against a plain CPU restatement of the
// decoder layer (f64 accumulation, OpenMP).  Not product code; links the library's kernel objects (python voxtral-mini-realtime-rs_amd/build.py first).
//   engine_b16_bench [n_layers=1] [pos=100] [reps=20] [tl_layer=-1] [flags=129] [n_rows=16] [check=1]
// Synthetic Q4 weights (random nibbles, f16 scales), random residual stream / KV caches, per-sequence positions pos - 3 m.  Checks (n_layers <= 4, check != 0): every edge
// buffer of the LAST layer the engine leaves in memory (q|k|v granules, attention outputs, post-attention stream, SwiGLU outputs), the new KV-cache rows of every layer, the
// layer stack's output (XF planes of h * final_norm, partial sums of squares); run-to-run bit-identical output; time per launch; optional per-phase timeline of one layer.
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

constexpr int D = 3072, NH = 32, NKV = 8, HD = 128, QD = 4096, KD = 1024, F = 9216, V = 131072, BM = 16;

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
struct HostQ4 { std::vector<uint8_t> qs; std::vector<uint16_t> sc; int N, K, nb; };
static HostQ4 to_host(const Q4W& w) { HostQ4 h; h.N = w.N; h.K = w.K; h.nb = w.nb; h.qs.resize((size_t)w.N * w.nb * 16); h.sc.resize((size_t)w.N * w.nb);
    CHK(hipMemcpy(h.qs.data(), w.qs, h.qs.size(), hipMemcpyDeviceToHost)); CHK(hipMemcpy(h.sc.data(), w.sc, h.sc.size() * 2, hipMemcpyDeviceToHost)); return h; }
static float h2f(uint16_t b) { return __half2float(__ushort_as_half(b)); }
// out[m][n] = sum_k x[m][k] * W[n][k] for M rows (f64 accumulation; W dequantised as (q - 8) * d, gguf/tensor.rs:98-109)
static void ref_linear(const HostQ4& w, const std::vector<double>& x, int M, std::vector<double>& out) {
    out.assign((size_t)M * w.N, 0.0);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < w.N; n++) {
        std::vector<float> row(w.K);
        for (int b = 0; b < w.nb; b++) {
            const uint8_t* q = &w.qs[((size_t)n * w.nb + b) * 16]; const float d = h2f(w.sc[(size_t)n * w.nb + b]);
            for (int i = 0; i < 16; i++) { row[32 * b + i] = ((int)(q[i] & 15) - 8) * d; row[32 * b + 16 + i] = ((int)(q[i] >> 4) - 8) * d; }
        }
        for (int m = 0; m < M; m++) { double a = 0; const double* xm = &x[(size_t)m * w.K]; for (int k = 0; k < w.K; k++) a += xm[k] * row[k]; out[(size_t)m * w.N + n] = a; }
    }
}
static float bf16f(uint16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
// fragment buffer [blk][plane][y * 16 + m][8 x bf16] -> value of (column, m)
static double frag_val(const std::vector<uint8_t>& buf, int col, int m) {
    const int blk = col >> 5, y = (col & 31) >> 3, i = col & 7;
    const uint16_t* hi = reinterpret_cast<const uint16_t*>(&buf[(size_t)((blk * 2 + 0) * 64 + y * 16 + m) * 16]);
    const uint16_t* lo = reinterpret_cast<const uint16_t*>(&buf[(size_t)((blk * 2 + 1) * 64 + y * 16 + m) * 16]);
    return (double)bf16f(hi[i]) + (double)bf16f(lo[i]);
}
// the launch-based path's XF layout (xf_store4 in vox_kernels.hip) -> value of (column k, row)
static double xf_val(const std::vector<uint16_t>& xf, int K, int k, int row) {
    const int q = k >> 7, j = (k >> 5) & 3, e = k & 31, half = e >> 4, g = (e & 15) >> 2, t = e & 3;
    const size_t base = ((size_t)((q * 4 + j) * 64 + g * 16 + row)) * 8 + 2 * half + ((t & 1) ? 4 : 0) + (t >> 1), plane = (size_t)(K >> 7) * 256 * 8;
    return (double)bf16f(xf[base]) + (double)bf16f(xf[plane + base]);
}
static bool report(const char* what, const std::vector<double>& ref, const std::vector<double>& got, double tol = 2e-4) {
    double m = 0, d = 0; size_t at = 0;
    for (size_t i = 0; i < ref.size(); i++) { m = std::max(m, std::fabs(ref[i])); const double e = std::fabs(ref[i] - got[i]); if (!(e <= d)) { d = e; at = i; } }
    const bool ok = d <= tol * (m > 0 ? m : 1);
    printf("  %-44s max|ref| %.4g  max|diff| %.3g  rel %.3g  (worst at %zu: ref %.6g got %.6g)  %s\n", what, m, d, d / (m > 0 ? m : 1), at, ref[at], got[at], ok ? "ok" : "MISMATCH");
    return ok;
}

int main(int argc, char** argv) {
    const int n_layers = argc > 1 ? atoi(argv[1]) : 1, pos0 = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 20, tl_layer = argc > 4 ? atoi(argv[4]) : -1,
              flags = argc > 5 ? atoi(argv[5]) : 129, n_rows = argc > 6 ? atoi(argv[6]) : 16;
    const bool check = (argc > 7 ? atoi(argv[7]) : 1) != 0 && n_layers <= 4;
    const int max_seq = pos0 < 248 ? 256 : ((pos0 + 8 + 63) / 64) * 64, window = 8192;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs; n_layers %d pos %d reps %d flags %d n_rows %d; engine LDS %d bytes, state %.1f MB\n", prop.name, prop.multiProcessorCount, n_layers, pos0, reps, flags, n_rows, engb_lds_bytes(), engb_state_bytes() / 1e6);
    { int occ = -1; hipError_t oe = engb_occupancy(&occ); printf("occupancy query: %s, %d workgroup(s) per CU\n", hipGetErrorString(oe), occ); }
    hipStream_t s; CHK(hipStreamCreate(&s));
    std::vector<Layer> L(n_layers);
    const float sb = 0.004f;
    for (int l = 0; l < n_layers; l++) {
        L[l].wqkv = make_q4(QD + 2 * KD, D, 1000u + 16 * l, sb); L[l].wo = make_q4(D, QD, 1001u + 16 * l, sb);
        L[l].w13 = make_q4(2 * F, D, 1002u + 16 * l, sb); L[l].w2 = make_q4(D, F, 1003u + 16 * l, sb);
        L[l].attn_norm = make_f32(D, 2000u + l, 1.0f, 0.2f); L[l].ffn_norm = make_f32(D, 3000u + l, 1.0f, 0.2f); L[l].ada = make_f32(D, 4000u + l, 1.0f, 0.1f);
    }
    float* final_norm = make_f32(D, 5001u, 1.0f, 0.2f);
    float* h_in = make_f32((size_t)BM * D, 5002u, 0.0f, 1.5f);
    const size_t seq_stride = (size_t)NKV * max_seq * HD, lf = (size_t)BM * seq_stride;      // [layer][sequence][kv head][max_seq][hd]
    float* kc0 = make_f32((size_t)n_layers * lf, 6001u, 0.0f, 1.0f); float* vc0 = make_f32((size_t)n_layers * lf, 6002u, 0.0f, 1.0f);
    float* kc = dalloc<float>((size_t)n_layers * lf); float* vc = dalloc<float>((size_t)n_layers * lf);
    CHK(hipMemcpy(kc, kc0, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(vc, vc0, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice));
    const int max_pos = 4096;
    std::vector<float> hc((size_t)max_pos * 64), hs((size_t)max_pos * 64);
    for (int p_ = 0; p_ < max_pos; p_++) for (int i = 0; i < 64; i++) { const double th = std::pow(1.0e6, -2.0 * i / 128.0); hc[(size_t)p_ * 64 + i] = (float)std::cos(p_ * th); hs[(size_t)p_ * 64 + i] = (float)std::sin(p_ * th); }
    float* rope_c = dalloc<float>(hc.size()); float* rope_s = dalloc<float>(hs.size());
    CHK(hipMemcpy(rope_c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(rope_s, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> pos(BM); for (int m = 0; m < BM; m++) pos[m] = std::max(pos0 - 3 * m, 1);
    int* d_pos = dalloc<int>(BM); CHK(hipMemcpy(d_pos, pos.data(), BM * 4, hipMemcpyHostToDevice));
    const float eps = 1e-5f;

    // ---- engine set-up
    const size_t sbytes = eng_stream_bytes(n_layers, V);
    unsigned char* stream = dalloc<unsigned char>(sbytes); CHK(hipMemset(stream, 0, sbytes));
    for (int l = 0; l < n_layers; l++) {
        CHK(launch_eng_pack(L[l].wqkv, 0, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].wo, 1, l, n_layers, stream, V, s));
        CHK(launch_eng_pack(L[l].w13, 2, l, n_layers, stream, V, s)); CHK(launch_eng_pack(L[l].w2, 3, l, n_layers, stream, V, s));
    }
    unsigned char* stream_wo = dalloc<unsigned char>(engb_wo_stream_bytes(n_layers)); CHK(hipMemset(stream_wo, 0, engb_wo_stream_bytes(n_layers)));
    for (int l = 0; l < n_layers; l++) CHK(launch_eng_pack(L[l].wo, 5, l, n_layers, stream_wo, V, s));      // wo in the XCD-group K split (the batched engine's own stream)
    unsigned char* state = dalloc<unsigned char>(engb_state_bytes()); CHK(engb_state_init(state, s)); CHK(hipStreamSynchronize(s));
    std::vector<EngLayerTab> tab(n_layers);
    for (int l = 0; l < n_layers; l++) tab[l] = EngLayerTab{L[l].attn_norm, L[l].ffn_norm, L[l].ada, kc + (size_t)l * lf, vc + (size_t)l * lf};
    EngLayerTab* d_tab = dalloc<EngLayerTab>(std::max(n_layers, 1)); CHK(hipMemcpy(d_tab, tab.data(), n_layers * sizeof(EngLayerTab), hipMemcpyHostToDevice));
    const size_t xf_u16 = (size_t)2 * (D / 128) * 256 * 8;
    uint16_t* xf_out = dalloc<uint16_t>(xf_u16); float* ssq_out = dalloc<float>(256 * BM);
    CHK(hipMemset(xf_out, 0, xf_u16 * 2)); CHK(hipMemset(ssq_out, 0, 256 * BM * 4));
    unsigned long long* tlbuf = dalloc<unsigned long long>(256 * 32); CHK(hipMemset(tlbuf, 0, 256 * 32 * 8));
    EngBParams ep{}; ep.stream = stream; ep.stream_wo = stream_wo; ep.layers = d_tab; ep.n_layers = n_layers; ep.kv_seq_stride = (long)seq_stride; ep.h_in = h_in; ep.h_stride = D; ep.n_rows = n_rows;
    ep.final_norm = final_norm; ep.pos = d_pos; ep.rope_cos = rope_c; ep.rope_sin = rope_s; ep.max_seq = max_seq; ep.window = window; ep.eps = eps;
    engb_state_carve(state, &ep); ep.xf_out = xf_out; ep.ssq_out = ssq_out; ep.tl = nullptr; ep.tl_layer = -1; ep.flags = flags;
    CHK(hipStreamSynchronize(s));
    auto check_err = [&](const char* when) { unsigned e[2]; CHK(hipMemcpy(e, ep.err, 8, hipMemcpyDeviceToHost)); if (e[0]) printf("ENGINE ERROR after %s: code %u, workgroup %u, tag bits %u\n", when, e[0] & 0xff, (e[0] >> 8) & 0xff, e[0] >> 16);
                                         if (e[1]) printf("  note (%s): XCD group check failed (workgroup %u on xcc %u): plain-store edges off\n", when, (e[1] >> 8) & 0xff, e[1] >> 16); return e[0]; };
    CHK(launch_decode_engine_b16(ep, s));
    hipError_t se = hipStreamSynchronize(s);
    if (se != hipSuccess) { printf("engine launch failed: %s\n", hipGetErrorString(se)); return 1; }
    if (check_err("first launch")) return 1;
    auto xf1 = d2h(xf_out, xf_u16); auto ss1 = d2h(ssq_out, 256 * BM);

    bool all_ok = true;
    if (check) {
        printf("CPU reference (f64 accumulation) ...\n");
        auto hin = d2h(h_in, (size_t)BM * D); auto k0 = d2h(kc0, (size_t)n_layers * lf), v0 = d2h(vc0, (size_t)n_layers * lf);
        std::vector<double> h((size_t)BM * D, 0.0);
        for (int m = 0; m < n_rows; m++) for (int k = 0; k < D; k++) h[(size_t)m * D + k] = hin[(size_t)m * D + k];
        std::vector<double> qkv, att((size_t)BM * QD), o, act((size_t)BM * F), gu, xn((size_t)BM * D), h1;
        std::vector<double> ref_q, ref_xo, ref_h1g, ref_act;
        auto kcE = d2h(kc, (size_t)n_layers * lf), vcE = d2h(vc, (size_t)n_layers * lf);
        for (int l = 0; l < n_layers; l++) {
            auto an = d2h(L[l].attn_norm, D), fn = d2h(L[l].ffn_norm, D), ad = d2h(L[l].ada, D);
            HostQ4 wqkv = to_host(L[l].wqkv), wo = to_host(L[l].wo), w13 = to_host(L[l].w13), w2 = to_host(L[l].w2);
            for (int m = 0; m < BM; m++) { double ss = 0; for (int k = 0; k < D; k++) ss += h[(size_t)m * D + k] * h[(size_t)m * D + k]; const double r = 1.0 / std::sqrt(ss / D + eps); for (int k = 0; k < D; k++) xn[(size_t)m * D + k] = h[(size_t)m * D + k] * r * an[k]; }
            ref_linear(wqkv, xn, BM, qkv);
            std::vector<double> newk((size_t)BM * KD), newv((size_t)BM * KD);
            for (int m = 0; m < BM; m++) {
                double* r = &qkv[(size_t)m * (QD + 2 * KD)]; const int pm = m < n_rows ? pos[m] : 0;
                for (int c = 0; c < QD + KD; c += 2) { const int pr = (c % HD) / 2; const double cs_ = hc[(size_t)pm * 64 + pr], sn = hs[(size_t)pm * 64 + pr]; const double a = r[c], b = r[c + 1]; r[c] = a * cs_ - b * sn; r[c + 1] = b * cs_ + a * sn; }
                for (int c = 0; c < KD; c++) { newk[(size_t)m * KD + c] = r[QD + c]; newv[(size_t)m * KD + c] = r[QD + KD + c]; }
            }
            ref_q = qkv;
            for (int m = 0; m < BM; m++) {
                const bool live = m < n_rows; const int pm = live ? pos[m] : 0, j_lo = std::max(0, pm - window);
                for (int hh = 0; hh < NH; hh++) {
                    const int g = hh / 4; const double* q = &qkv[(size_t)m * (QD + 2 * KD) + hh * HD];
                    std::vector<double> sc; std::vector<const float*> vr; std::vector<double> vnew(HD);
                    const float* kb = &k0[(size_t)l * lf + (size_t)std::min(m, n_rows - 1) * seq_stride + (size_t)g * max_seq * HD]; const float* vb = &v0[(size_t)l * lf + (size_t)std::min(m, n_rows - 1) * seq_stride + (size_t)g * max_seq * HD];
                    double mx = -1e300;
                    if (live) for (int jj = j_lo; jj < pm; jj++) { double a = 0; for (int d = 0; d < HD; d++) a += q[d] * kb[(size_t)jj * HD + d]; a /= std::sqrt((double)HD); sc.push_back(a); mx = std::max(mx, a); vr.push_back(vb + (size_t)jj * HD); }
                    double an_ = 0; for (int d = 0; d < HD; d++) an_ += q[d] * newk[(size_t)m * KD + g * HD + d]; an_ /= std::sqrt((double)HD); mx = std::max(mx, an_);
                    double den = 0; std::vector<double> ov(HD, 0.0);
                    for (size_t t = 0; t < sc.size(); t++) { const double pw = std::exp(sc[t] - mx); den += pw; for (int d = 0; d < HD; d++) ov[d] += pw * vr[t][d]; }
                    { const double pw = std::exp(an_ - mx); den += pw; for (int d = 0; d < HD; d++) ov[d] += pw * newv[(size_t)m * KD + g * HD + d]; }
                    for (int d = 0; d < HD; d++) att[(size_t)m * QD + hh * HD + d] = live ? ov[d] / den : 0.0;
                }
            }
            ref_xo = att;
            ref_linear(wo, att, BM, o);
            h1 = h; for (size_t i = 0; i < h1.size(); i++) h1[i] += o[i];
            ref_h1g.assign((size_t)BM * D, 0.0);
            for (int m = 0; m < BM; m++) { double ss = 0; for (int k = 0; k < D; k++) ss += h1[(size_t)m * D + k] * h1[(size_t)m * D + k]; const double r = 1.0 / std::sqrt(ss / D + eps);
                for (int k = 0; k < D; k++) { xn[(size_t)m * D + k] = h1[(size_t)m * D + k] * r * fn[k] * ad[k]; ref_h1g[(size_t)m * D + k] = h1[(size_t)m * D + k] * ((double)fn[k] * ad[k]); } }
            ref_linear(w13, xn, BM, gu);
            for (int m = 0; m < BM; m++) for (int i = 0; i < F; i++) { const double gt = gu[(size_t)m * 2 * F + 2 * i], up = gu[(size_t)m * 2 * F + 2 * i + 1]; act[(size_t)m * F + i] = gt / (1.0 + std::exp(-gt)) * up; }
            ref_act = act;
            ref_linear(w2, act, BM, o);
            h = h1; for (size_t i = 0; i < h.size(); i++) h[i] += o[i];
            // new cache rows of this layer
            std::vector<double> rk, gk, rv, gv;
            for (int m = 0; m < n_rows; m++) for (int g = 0; g < NKV; g++) for (int d = 0; d < HD; d++) {
                const size_t ci = (size_t)l * lf + (size_t)m * seq_stride + (size_t)g * max_seq * HD + (size_t)pos[m] * HD + d;
                rk.push_back(newk[(size_t)m * KD + g * HD + d]); gk.push_back(kcE[ci]); rv.push_back(newv[(size_t)m * KD + g * HD + d]); gv.push_back(vcE[ci]);
            }
            char nm[64]; snprintf(nm, sizeof nm, "layer %d: new K cache rows", l); all_ok &= report(nm, rk, gk); snprintf(nm, sizeof nm, "layer %d: new V cache rows", l); all_ok &= report(nm, rv, gv);
            size_t touched = 0; for (size_t i = 0; i < lf; i++) touched += kcE[(size_t)l * lf + i] != k0[(size_t)l * lf + i];
            printf("  layer %d: K cache words changed: %zu (expected <= %d)\n", l, touched, n_rows * KD);
        }
        printf("engine vs CPU reference (edges of the last layer, then the stack's output):\n");
        if (n_layers > 0) {
            {   // q|k|v granules
                auto g = d2h(ep.G, (size_t)BM * 6144); std::vector<double> got(g.size()); unsigned tmin = ~0u, tmax = 0;
                for (size_t i = 0; i < g.size(); i++) { const unsigned lo = (unsigned)g[i], hi = (unsigned)(g[i] >> 32); float f; memcpy(&f, &lo, 4); got[i] = f; tmin = std::min(tmin, hi); tmax = std::max(tmax, hi); }
                printf("  G tags %u..%u\n", tmin, tmax);
                all_ok &= report("q|k|v after RoPE [16][6144]", ref_q, got);
            }
            {   // attention outputs: XO [head][4 blocks]...
                auto xo = d2h(ep.XO, (size_t)NH * 8192); std::vector<double> got((size_t)BM * QD);
                for (int hh = 0; hh < NH; hh++) { std::vector<uint8_t> one(xo.begin() + (size_t)hh * 8192, xo.begin() + (size_t)(hh + 1) * 8192); for (int m = 0; m < BM; m++) for (int d = 0; d < HD; d++) got[(size_t)m * QD + hh * HD + d] = frag_val(one, d, m); }
                all_ok &= report("attention output [16][4096]", ref_xo, got);
            }
            {   // post-attention stream * ffn_norm * ada
                auto xh = d2h(ep.XH1, (size_t)96 * 2048); std::vector<double> got((size_t)BM * D);
                for (int m = 0; m < BM; m++) for (int k = 0; k < D; k++) got[(size_t)m * D + k] = frag_val(xh, k, m);
                all_ok &= report("(h + wo(att)) * ffn_norm * ada [16][3072]", ref_h1g, got);
            }
            {   // SwiGLU outputs: XA [group][36 blocks]
                auto xa = d2h(ep.XA, (size_t)8 * 73728); std::vector<double> got((size_t)BM * F);
                for (int g = 0; g < 8; g++) { std::vector<uint8_t> one(xa.begin() + (size_t)g * 73728, xa.begin() + (size_t)(g + 1) * 73728); for (int m = 0; m < BM; m++) for (int k = 0; k < 1152; k++) got[(size_t)m * F + 1152 * g + k] = frag_val(one, k, m); }
                all_ok &= report("SwiGLU activations [16][9216]", ref_act, got);
            }
        }
        {   // the stack's output
            auto fnw = d2h(final_norm, D); std::vector<double> ref((size_t)BM * D), got((size_t)BM * D), rss(BM, 0.0), gss(BM, 0.0);
            for (int m = 0; m < BM; m++) for (int k = 0; k < D; k++) { ref[(size_t)m * D + k] = h[(size_t)m * D + k] * fnw[k]; got[(size_t)m * D + k] = xf_val(xf1, D, k, m); rss[m] += h[(size_t)m * D + k] * h[(size_t)m * D + k]; }
            for (int b = 0; b < 256; b++) for (int m = 0; m < BM; m++) gss[m] += ss1[(size_t)b * BM + m];
            all_ok &= report("layer-stack output h * final_norm (XF)", ref, got);
            all_ok &= report("sums of squares per sequence", rss, gss);
        }
    }
    // ---- determinism + timing
    CHK(hipMemcpy(kc, kc0, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice)); CHK(hipMemcpy(vc, vc0, (size_t)n_layers * lf * 4, hipMemcpyDeviceToDevice));
    CHK(launch_decode_engine_b16(ep, s)); CHK(hipStreamSynchronize(s));
    auto xf2 = d2h(xf_out, xf_u16); auto ss2 = d2h(ssq_out, 256 * BM);
    const bool same = memcmp(xf1.data(), xf2.data(), xf_u16 * 2) == 0 && memcmp(ss1.data(), ss2.data(), ss1.size() * 4) == 0;
    printf("  run-to-run: %s\n", same ? "bit-identical" : "DIFFERENT"); all_ok &= same;
    if (check_err("second launch")) return 1;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) CHK(launch_decode_engine_b16(ep, s));
    CHK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) CHK(launch_decode_engine_b16(ep, s)); CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n_layers * 65470464.0;
    printf("engine b16: %.1f us per launch (%d layers, %d rows) = %.1f us per layer = %.2f TB/s of Q4 bytes, %.3f of 8 TB/s\n", ms * 1000 / reps, n_layers, n_rows, ms * 1000 / reps / std::max(n_layers, 1),
           bytes / (ms * 1e-3 / reps) / 1e12, bytes / (ms * 1e-3 / reps) / 8e12);
    check_err("timing loop");
    if (tl_layer >= 0) {
        ep.tl = tlbuf; ep.tl_layer = tl_layer;
        CHK(launch_decode_engine_b16(ep, s)); CHK(hipStreamSynchronize(s));
        std::vector<unsigned long long> tb(256 * 32); CHK(hipMemcpy(tb.data(), tlbuf, tb.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int b = 0; b < 256; b++) if (tb[b * 32 + 19]) t0 = std::min(t0, tb[b * 32 + 19]);
        const char* names[32] = {"cons: A(q|k|v) in registers", "cons: q|k|v published (wave 0)", "cons: q,k,v of the sequences staged", "cons: attention published (wave 0)", "cons: A(wo) in registers", "cons: wo plane stored (wave 0)",
                                 "cons: A(w13) in registers", "cons: SwiGLU published (wave 0)", "comm: h all-gather: poll start", "comm: h all-gather: flags in", "comm: q|k|v granules swept", "comm: attention flags in",
                                 "comm: wo flag written", "comm: wo plane flags in", "comm: h1 published", "cons: w2 plane stored (wave 0)", "load: layer's first packet issued", "load: layer's last packet issued", "load: stream done", "kernel start",
                                 "comm: h1 all-gather: poll start", "comm: h1 all-gather: flags in", "comm: SwiGLU flags in", "comm: w2 flag written", "comm: w2 plane flags in", "comm: h2 published", "cons: A(w2) in registers", "cons(w13): GEMM done (wave 0)", "cons(w13): barrier 1 passed", "cons(w13): SwiGLU in LDS (wave 0)", "cons(w13): barrier 2 passed", ""};
        const int order[31] = {19, 16, 8, 9, 0, 1, 10, 2, 3, 11, 4, 5, 12, 13, 14, 20, 21, 6, 27, 28, 29, 30, 7, 22, 26, 15, 23, 24, 25, 17, 18};
        printf("timeline of layer %d (us since the first workgroup started; min / median / max over the 256 CUs):\n", tl_layer);
        for (int oi = 0; oi < 31; oi++) {
            const int e = order[oi]; std::vector<double> v;
            for (int b = 0; b < 256; b++) if (tb[b * 32 + e]) v.push_back((double)(tb[b * 32 + e] - t0) / 100.0);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  %-40s %9.2f %9.2f %9.2f\n", names[e], v.front(), v[v.size() / 2], v.back());
        }
        check_err("timeline launch");
    }
    printf("%s\n", all_ok ? "ALL OK" : "SOME CHECKS FAILED");
    return all_ok ? 0 : 2;
}
