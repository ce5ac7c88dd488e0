// Stand-alone reproducer (no library code).  Victim: the arithmetic of rope_kernel -- an in-place rotation of (x0, x1) pairs by table values, which hipcc turns into
// v_pk_mul_f32 / v_pk_fma_f32 on gfx950 -- on SMALL INTEGER data (every product and sum exact in f32: the result does not depend on contraction).  Aggressors on a second
// stream: (1) an MFMA loop, (2) a plain VALU loop, (3) a copy.  The victim's output is checked on the device after every launch.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fp32_corun tools/repro/pk_fp32_corun.cpp -lpthread && ./pk_fp32_corun
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float val(int it, long i) { return (float)((int)(((unsigned)i * 2654435761u + (unsigned)it * 40503u) >> 20 & 63) - 32); }
__global__ __launch_bounds__(256) void write_k(float* buf, long n, int it) { for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) buf[i] = val(it, i); }
__global__ void rot_k(float* __restrict__ buf, int M, int stride, int n_rot, int hd, const float* __restrict__ cos_t, const float* __restrict__ sin_t) {      // == rope_kernel
    const int half_cols = n_rot >> 1; const long total = (long)M * half_cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / half_cols), pc = (int)(i % half_cols); const int col = pc * 2, j = (col % hd) >> 1;
        const size_t ti = (size_t)m * (hd >> 1) + j;
        const float c = cos_t[ti], sn = sin_t[ti];
        float* p = buf + (size_t)m * stride + col;
        const float xr = p[0], xi = p[1];
        p[0] = xr * c - xi * sn; p[1] = xr * sn + xi * c;
    }
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same arithmetic with the instruction sequence pinned: V = 0 the compiler's own sequence (in-place v_pk_mul_f32, then v_pk_fma_f32 reading its result as src2);
// V = 1..4: s_nop (2^(V-1) - 1 .. ) wait states between the two; V = 5: the product into a fresh register pair (not in place); V = 6: scalar v_mul / v_fma
template <int V>
__global__ void rot_asm_k(float* buf, int M, int stride, int n_rot, int hd, const float* cos_t, const float* sin_t) {
    const int half_cols = n_rot >> 1; const long total = (long)M * half_cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / half_cols), pc = (int)(i % half_cols); const int col = pc * 2, j = (col % hd) >> 1;
        const size_t ti = (size_t)m * (hd >> 1) + j;
        f32x2 c2, t, x, o0, o1; c2.x = cos_t[ti]; c2.y = 0.f; t.x = sin_t[ti]; t.y = 0.f;
        float* p = buf + (size_t)m * stride + col;
        x = *reinterpret_cast<f32x2*>(p);
        if (V == 6) { const float a = t.x * x.y, b = t.x * x.x; asm volatile("" ::: "memory"); o0.x = __builtin_fmaf(c2.x, x.x, -a); o0.y = __builtin_fmaf(c2.x, x.y, b); }
        else if (V == 5) { f32x2 u;
            asm volatile("v_pk_mul_f32 %[u], %[t], %[x] op_sel:[0,1] op_sel_hi:[0,0]\n\tv_pk_fma_f32 %[o0], %[c], %[x], %[u] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 %[o1], %[c], %[x], %[u] op_sel_hi:[0,1,1]"
                         : [u] "=&v"(u), [o0] "=&v"(o0), [o1] "=&v"(o1) : [t] "v"(t), [x] "v"(x), [c] "v"(c2)); o0.y = o1.y; }
        else {
#define SEQ(NOP_) asm volatile("v_pk_mul_f32 %[t], %[t], %[x] op_sel:[0,1] op_sel_hi:[0,0]\n\t" NOP_ "v_pk_fma_f32 %[o0], %[c], %[x], %[t] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\tv_pk_fma_f32 %[o1], %[c], %[x], %[t] op_sel_hi:[0,1,1]" \
                         : [t] "+v"(t), [o0] "=&v"(o0), [o1] "=&v"(o1) : [x] "v"(x), [c] "v"(c2))
            if (V == 0) SEQ(""); else if (V == 1) SEQ("s_nop 0\n\t"); else if (V == 2) SEQ("s_nop 1\n\t"); else if (V == 3) SEQ("s_nop 3\n\t"); else SEQ("s_nop 7\n\t");
#undef SEQ
            o0.y = o1.y;
        }
        *reinterpret_cast<f32x2*>(p) = o0;
    }
}
// which encodings?  E = 0: v_pk_mul_f32 straight lanes; 1: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] (lo result from src1.hi, hi result from src1.lo);
// 2: v_pk_fma_f32 straight lanes; 3: v_pk_fma_f32 op_sel_hi:[0,1,1] (src0.lo broadcast); 4: v_pk_fma_f32 with neg_lo / neg_hi on src2; 5: v_pk_add_f32 straight
template <int E>
__global__ void enc_k(float* buf, int M, int stride, int n_rot, int hd, const float* cos_t, const float* sin_t) {
    const int half_cols = n_rot >> 1; const long total = (long)M * half_cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / half_cols), pc = (int)(i % half_cols); const int col = pc * 2, j = (col % hd) >> 1;
        const size_t ti = (size_t)m * (hd >> 1) + j;
        const float c = cos_t[ti], sn = sin_t[ti];
        float* p = buf + (size_t)m * stride + col;
        const f32x2 x = *reinterpret_cast<f32x2*>(p);
        f32x2 o;
        if (E == 0) { f32x2 s2 = {sn, sn}, u; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(s2), "v"(x)); /* u = (sn xr, sn xi) */ o.x = __builtin_fmaf(c, x.x, -u.y); o.y = __builtin_fmaf(c, x.y, u.x); }
        else if (E == 1) { f32x2 s2 = {sn, 0.f}, u; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(u) : "v"(s2), "v"(x)); /* u = (sn xi, sn xr) */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else if (E == 2) { f32x2 c2 = {c, c}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(o) : "v"(c2), "v"(x), "v"(u)); }
        else if (E == 3) { f32x2 c2 = {c, 0.f}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(o) : "v"(c2), "v"(x), "v"(u)); }
        else if (E == 4) { f32x2 c2 = {c, c}, u = {sn * x.y, -sn * x.x}; asm volatile("" : "+v"(u)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(o) : "v"(c2), "v"(x), "v"(u)); }
        else if (E == 5) { f32x2 a = {c * x.x, c * x.y}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u), "+v"(a)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(u)); }
        else if (E == 6) { f32x2 s2 = {sn, 0.f}, u; asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(u) : "v"(s2), "v"(x)); /* src0 = x swapped: u = (xi sn, xr sn) */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else if (E == 7) { f32x2 a = {c * x.x, c * x.y}, u = {sn * x.x, -sn * x.y}; asm volatile("" : "+v"(u), "+v"(a)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(o) : "v"(a), "v"(u)); /* src1 swapped */ }
        else if (E == 8) { f32x2 c2 = {c, c}, u = {sn * x.x, -sn * x.y}; asm volatile("" : "+v"(u)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(o) : "v"(c2), "v"(x), "v"(u)); /* src2 swapped */ }
        else if (E == 9) { f32x2 xs = {x.y, x.x}, c2 = {c, c}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u), "+v"(xs)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(o) : "v"(xs), "v"(c2), "v"(u)); /* src0 swapped */ }
        else if (E == 10) { f32x2 xs = {x.y, x.x}, c2 = {c, c}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u), "+v"(xs)); asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(o) : "v"(xs), "v"(c2), "v"(u)); /* src1 swapped */ }
        else if (E == 12) { f32x2 s2 = {sn, 7.f}, xs = {x.y, x.x}, u; asm volatile("" : "+v"(xs)); asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[1,0]" : "=v"(u) : "v"(s2), "v"(xs)); /* src1 = s2, lo broadcast: u = (xi sn, xr sn) */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else if (E == 13) { f32x2 c2 = {c, 7.f}, u = {-sn * x.y, sn * x.x}; asm volatile("" : "+v"(u)); asm volatile("v_pk_fma_f32 %0, %2, %1, %3 op_sel_hi:[1,0,1]" : "=v"(o) : "v"(c2), "v"(x), "v"(u)); /* src1 = c2, lo broadcast */ }
        else if (E == 14) { f32x2 s2 = {sn, 7.f}, xs = {x.y, x.x}, u; asm volatile("" : "+v"(xs)); asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(u) : "v"(s2), "v"(xs)); /* src0 = s2, lo broadcast */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else if (E == 15) { f32x2 sx = {sn * x.x, 7.f}, c2 = {c, c}, xs = {-x.y / (x.x != 0.f ? x.x : 1.f), 1.f}; (void)xs; f32x2 u0 = {0.f, 0.f}; (void)u0;
                            /* src2 lo broadcast: o = (c x.x + t, c x.y + t) with t = src2.lo; only o.y is the rotation's (t = sn xr); o.x is recomputed scalar */
                            f32x2 r; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(c2), "v"(x), "v"(sx)); o.y = r.y; o.x = __builtin_fmaf(c, x.x, -(sn * x.y)); }
        else if (E == 16 || E == 17 || E == 18) { f32x2 a = {x.x, x.y}, b = {x.x, x.y}, sw; asm volatile("" : "+v"(a), "+v"(b));
            if (E == 16) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(sw) : "v"(a), "v"(b));       /* (a.lo, b.hi): straight */
            else if (E == 17) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(sw) : "v"(a), "v"(b));  /* (a.hi, b.hi): src0 crosses */
            else asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(sw) : "v"(a), "v"(b));               /* (a.lo, b.lo): src1 crosses */
            const float xr = sw.x == x.x ? x.x : sw.x, xi = E == 16 ? sw.y : x.y;      /* E 17: sw = (xi, xi); E 18: sw = (xr, xr) */
            const float XR = E == 17 ? x.x : xr, XI = E == 17 ? sw.x : (E == 18 ? x.y : xi), chk = E == 17 ? sw.y - x.y : (E == 18 ? sw.y - x.x : 0.f);      /* chk != 0: the moved lane was wrong */
            o.x = __builtin_fmaf(c, XR, -(sn * XI)) + chk * 1000.f; o.y = __builtin_fmaf(c, XI, sn * XR); }
        else if (E == 19) { f32x2 a = {123.f, x.y}, b = {x.x, 456.f}, sw; asm volatile("" : "+v"(a), "+v"(b)); asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(sw) : "v"(a), "v"(b)); /* DISTINCT registers: sw = (a.hi, b.lo) = (xi, xr) */
                            o.x = __builtin_fmaf(c, sw.y, -(sn * sw.x)); o.y = __builtin_fmaf(c, sw.x, sn * sw.y); }
        else if (E == 20) { f32x2 s2 = {7.f, sn}, xs = {x.y, x.x}, u; asm volatile("" : "+v"(xs)); asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(u) : "v"(s2), "v"(xs)); /* src1 = s2, HI broadcast: u = (xi sn, xr sn) */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else if (E == 21) { f32x2 s2 = {7.f, sn}, xs = {x.y, x.x}, u; asm volatile("" : "+v"(xs)); asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(u) : "v"(s2), "v"(xs)); /* src0 = s2, HI broadcast */ o.x = __builtin_fmaf(c, x.x, -u.x); o.y = __builtin_fmaf(c, x.y, u.y); }
        else { f32x2 sw; asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(sw) : "v"(x)); /* sw = (x.hi, x.lo): the only v_pk_mov_b32 form hipcc emits in the library */
               o.x = __builtin_fmaf(c, sw.y, -(sn * sw.x)); o.y = __builtin_fmaf(c, sw.x, sn * sw.y); }
        *reinterpret_cast<f32x2*>(p) = o;
    }
}
__global__ __launch_bounds__(256) void check_k(const float* buf, int M, int stride, int n_rot, int hd, const float* cos_t, const float* sin_t, int it, unsigned* bad) {
    const int half_cols = n_rot >> 1; const long total = (long)M * half_cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / half_cols), pc = (int)(i % half_cols); const int col = pc * 2, j = (col % hd) >> 1; const size_t ti = (size_t)m * (hd >> 1) + j, o = (size_t)m * stride + col;
        const float c = cos_t[ti], sn = sin_t[ti], xr = val(it, (long)o), xi = val(it, (long)o + 1);
        const float e0 = (float)((int)xr * (int)c - (int)xi * (int)sn), e1 = (float)((int)xr * (int)sn + (int)xi * (int)c);      // integer arithmetic: not the instructions under test
        if (buf[o] != e0 || buf[o + 1] != e1) { const unsigned k = atomicAdd(bad, 1u); if (k < 4) { float* r = (float*)(bad + 4 + 8 * k); r[0] = buf[o]; r[1] = buf[o + 1]; r[2] = e0; r[3] = e1; r[4] = xr; r[5] = xi; r[6] = c; r[7] = sn; } }
    }
}
__global__ __launch_bounds__(256, 2) void mfma_k(float* out, int iters) {      // aggressor 1: dependent MFMA chains, 4 waves per workgroup
    bf16x8 a, b; for (int i = 0; i < 8; i++) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    f32x4 acc[8]; for (auto& x : acc) x = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    float s = 0.f; for (auto& x : acc) s += x[0] + x[1] + x[2] + x[3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void valu_k(float* out, int iters) {      // aggressor 2: plain FMAs
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int it = 0; it < iters; it++) { x0 = fmaf(x0, 1.0001f, x1); x1 = fmaf(x1, 0.9999f, x2); x2 = fmaf(x2, 1.0001f, x3); x3 = fmaf(x3, 0.9999f, x0); }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    const int M = 484, QD = 2048, stride = 3 * QD, n_rot = 2 * QD, hd = 64; const long n = (long)M * stride;
    hipStream_t sv, sa; CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    float *buf, *ct, *st, *junk, *c1, *c2; unsigned* bad;
    CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&ct, (size_t)M * 32 * 4)); CK(hipMalloc(&st, (size_t)M * 32 * 4)); CK(hipMalloc(&junk, (size_t)4096 * 256 * 4)); CK(hipMalloc(&bad, 4096)); CK(hipMemset(bad, 0, 4096));
    CK(hipMalloc(&c1, n * 4)); CK(hipMalloc(&c2, n * 4));
    { std::vector<float> hc((size_t)M * 32), hs((size_t)M * 32); for (size_t i = 0; i < hc.size(); i++) { hc[i] = (float)((int)(i * 7 % 9) - 4); hs[i] = (float)((int)(i * 5 % 7) - 3); }
      CK(hipMemcpy(ct, hc.data(), hc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(st, hs.data(), hs.size() * 4, hipMemcpyHostToDevice)); }
    const long total = (long)M * (n_rot / 2); int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    int variant = -1;
    auto victim = [&](int base) {
        for (int it = 0; it < iters; it++) {
            write_k<<<dim3(1024), dim3(256), 0, sv>>>(buf, n, base + it);
            switch (variant) {
            case 0: rot_asm_k<0><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 1: rot_asm_k<1><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 2: rot_asm_k<2><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 3: rot_asm_k<3><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 4: rot_asm_k<4><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 5: rot_asm_k<5><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 6: rot_asm_k<6><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 10: enc_k<0><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 11: enc_k<1><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 12: enc_k<2><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 13: enc_k<3><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 14: enc_k<4><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 15: enc_k<5><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 16: enc_k<6><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 17: enc_k<7><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 18: enc_k<8><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 19: enc_k<9><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 20: enc_k<10><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 21: enc_k<11><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 22: enc_k<12><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 23: enc_k<13><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 24: enc_k<14><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 25: enc_k<15><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 26: enc_k<16><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 27: enc_k<17><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 28: enc_k<18><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 29: enc_k<19><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 30: enc_k<20><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            case 31: enc_k<21><<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            default: rot_k<<<dim3(blocks), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st); break;
            }
            check_k<<<dim3(1024), dim3(256), 0, sv>>>(buf, M, stride, n_rot, hd, ct, st, base + it, bad);
        }
        CK(hipStreamSynchronize(sv)); unsigned h[4 + 32]; CK(hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemset(bad, 0, 4096));
        if (h[0]) { const float* r = (const float*)(h + 4); printf("      e.g. got (%g, %g) expected (%g, %g) from x (%g, %g), c %g, s %g\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]); }
        return h[0];
    };
    printf("victim alone: %u wrong pairs (of %ld x %d)\n", victim(0), total, iters);
    {   // the pinned sequences next to the MFMA loop
        const char* vn[7] = {"in-place v_pk_mul_f32 -> v_pk_fma_f32, back to back", "... s_nop 0 between", "... s_nop 1 between", "... s_nop 3 between", "... s_nop 7 between", "product into a fresh register pair", "scalar v_mul_f32 / v_fma_f32"};
        for (variant = 0; variant < 7; variant++) {
            const unsigned w0 = victim(50000 + 100 * variant);
            std::atomic<bool> stop{false};
            std::thread t([&] { while (!stop.load()) { for (int k = 0; k < 20; k++) mfma_k<<<dim3(2048), dim3(256), 0, sa>>>(junk, 2000); CK(hipStreamSynchronize(sa)); } });
            const unsigned w = victim(60000 + 100 * variant);
            stop.store(true); t.join();
            printf("[%s] alone: %u wrong pairs; next to the MFMA loop: %u wrong pairs\n", vn[variant], w0, w);
        }
        const char* en[22] = {"v_pk_mul_f32, straight lanes", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,0] (src1 swapped)", "v_pk_fma_f32, straight lanes", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_fma_f32 neg_lo / neg_hi on src2", "v_pk_add_f32, straight lanes",
                              "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,0] (src0 swapped)", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 swapped)", "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 swapped)", "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] (src0 swapped)", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (src1 swapped)", "v_pk_mov_b32 op_sel:[1,0] (lo <- src0.hi, hi <- src1.lo)",
                              "v_pk_mul_f32 op_sel_hi:[1,0] (src1.lo broadcast)", "v_pk_fma_f32 op_sel_hi:[1,0,1] (src1.lo broadcast)", "v_pk_mul_f32 op_sel_hi:[0,1] (src0.lo broadcast)", "v_pk_fma_f32 op_sel_hi:[1,1,0] (src2.lo broadcast)",
                              "v_pk_mov_b32 op_sel:[0,1] (straight)", "v_pk_mov_b32 op_sel:[1,1] (lo <- src0.hi)", "v_pk_mov_b32 op_sel:[0,0] (hi <- src1.lo)",
                              "v_pk_mov_b32 op_sel:[1,0], DISTINCT source registers", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,1] (src1.hi broadcast)", "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,1] (src0.hi broadcast)"};
        for (variant = (argc > 2 ? atoi(argv[2]) : 10); variant < 32; variant++) {
            const unsigned w0 = victim(70000 + 100 * variant);
            std::atomic<bool> stop{false};
            std::thread t([&] { while (!stop.load()) { for (int k = 0; k < 20; k++) mfma_k<<<dim3(2048), dim3(256), 0, sa>>>(junk, 2000); CK(hipStreamSynchronize(sa)); } });
            const unsigned w = victim(80000 + 100 * variant);
            stop.store(true); t.join();
            printf("[one packed instruction: %s] alone: %u wrong pairs; next to the MFMA loop: %u wrong pairs\n", en[variant - 10], w0, w);
        }
        variant = -1;
    }
    const char* names[4] = {"MFMA loop", "VALU loop", "device copy", "MFMA loop, 1 workgroup per CU"};
    for (int ag = 0; ag < 4; ag++) {
        std::atomic<bool> stop{false};
        std::thread t([&] { while (!stop.load()) { for (int k = 0; k < 20; k++) { if (ag == 0) mfma_k<<<dim3(2048), dim3(256), 0, sa>>>(junk, 2000); else if (ag == 1) valu_k<<<dim3(4096), dim3(256), 0, sa>>>(junk, 20000);
                                                        else if (ag == 2) CK(hipMemcpyAsync(c2, c1, n * 4, hipMemcpyDeviceToDevice, sa)); else mfma_k<<<dim3(256), dim3(256), 0, sa>>>(junk, 20000); } CK(hipStreamSynchronize(sa)); } });
        const unsigned w = victim(1000 * (ag + 1));
        stop.store(true); t.join();
        printf("victim next to a %s on another stream: %u wrong pairs\n", names[ag], w);
    }
    return 0;
}
