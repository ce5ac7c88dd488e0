/*
 * vox_oracle.c -- CPU ORACLE (test infrastructure; see vox_oracle.h header note).
 *
 * Plain C11 restatement of the reference's arithmetic for the hot path
 * mel -> encoder -> adapter -> decoder (Q4_0 weights dequantised on the fly,
 * f32 everywhere else -- the definition the reference's own tests use,
 * gguf/tests.rs:170-185).  All f32 products are "mul then add" (Rust never
 * contracts to FMA), so the build uses -ffp-contract=off.
 *
 * Summation order: sequential over k, as gguf/tests.rs:172-185
 * (`reference_matmul`).  Burn/wgpu's own GPU reduction order is third-party and
 * unpinned, so this is the reference's *test* oracle order.
 */
#define _GNU_SOURCE
#include "vox_oracle.h"

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512];
const char* orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); } while (0)

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static double now_ms(void) {
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ---------------------------------------------------------------- half */
/* `half` crate semantics (IEEE binary16, round-to-nearest-even). */
static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { /* subnormal */
            int e = -1; do { man <<= 1; e++; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}
static uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u; x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          /* overflow -> inf (after rounding) */
    if (x < 0x33000001u) return (uint16_t)sign;                        /* underflow -> 0 */
    int e = (int)(x >> 23) - 127; uint32_t m = (x & 0x7fffffu) | 0x800000u;
    int shift; uint32_t half;
    if (e < -14) { shift = 13 + (-14 - e); half = 0; }                 /* subnormal */
    else { shift = 13; half = (uint32_t)(e + 15) << 10; }
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (e >= -14) r &= 0x3ffu;
    uint32_t out = half + r;                                           /* carry into exponent is correct */
    if (rem > mid || (rem == mid && (out & 1))) out++;
    return (uint16_t)(sign | out);
}
uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

/* ---------------------------------------------------------------- audio */

/* audio/io.rs:59-68 */
void orc_peak_normalize(float* s, size_t n, float target_peak) {
    float max_amp = 0.0f;
    for (size_t i = 0; i < n; i++) { float a = fabsf(s[i]); if (a > max_amp) max_amp = a; }
    if (max_amp < 1e-10f) return;
    float scale = target_peak / max_amp;
    for (size_t i = 0; i < n; i++) s[i] *= scale;
}

/* audio/pad.rs:32-46 */
void orc_pad_cfg_voxtral(orc_pad_cfg* c) {
    c->sample_rate = 16000; c->n_left_pad_tokens = 76; c->frame_rate = 12.5f; c->extra_right_pad_tokens = 17;
}
/* audio/pad.rs:55-57 */
size_t orc_pad_samples_per_token(const orc_pad_cfg* c) { return (size_t)((float)c->sample_rate / c->frame_rate); }
/* audio/pad.rs:60-62 */
size_t orc_pad_left_samples(const orc_pad_cfg* c) { return c->n_left_pad_tokens * orc_pad_samples_per_token(c); }
/* audio/pad.rs:68-74 */
size_t orc_pad_right_samples(const orc_pad_cfg* c, size_t total) {
    size_t spt = orc_pad_samples_per_token(c), rem = total % spt;
    size_t align = rem == 0 ? 0 : spt - rem;
    return align + c->extra_right_pad_tokens * spt;
}
/* audio/pad.rs:89-103 */
size_t orc_pad_len(const orc_pad_cfg* c, size_t n) {
    size_t left = orc_pad_left_samples(c);
    return left + n + orc_pad_right_samples(c, n + left);
}
void orc_pad_audio(const orc_pad_cfg* c, const float* in, size_t n, float* out) {
    size_t left = orc_pad_left_samples(c), total = orc_pad_len(c, n);
    memset(out, 0, total * sizeof(float));
    memcpy(out + left, in, n * sizeof(float));
}

/* audio/chunk.rs:164-166 */
int orc_needs_chunking(size_t n, const orc_chunk_cfg* c) { return n > (size_t)c->max_mel_frames * c->hop_length; }
/* audio/chunk.rs:120-161 (ChunkIterator) */
size_t orc_chunk_plan(size_t n, const orc_chunk_cfg* c, orc_chunk* out, size_t cap) {
    size_t max_s = (size_t)c->max_mel_frames * c->hop_length;
    size_t step = (size_t)(c->max_mel_frames - c->overlap_frames) * c->hop_length;
    size_t pos = 0, idx = 0;
    while (pos < n) {
        size_t end = pos + max_s; if (end > n) end = n;
        if (out && idx < cap) { out[idx].start_sample = pos; out[idx].end_sample = end; out[idx].index = idx; out[idx].is_last = end >= n; }
        pos += step; idx++;
        if (step == 0) break;
    }
    return idx;
}

#define N_FFT 400
#define HOP 160
#define N_MELS 128
#define N_FREQ 201
#define PI_F 3.14159265358979323846f /* std::f32::consts::PI */

/* audio/mel.rs:345-349 */
void orc_hann_window(int length, float* out) {
    for (int i = 0; i < length; i++)
        out[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)length));
}
/* audio/mel.rs:260-271 */
float orc_hz_to_mel(float f) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    if (f < MIN_LOG_HZ) return f / F_SP;
    return MIN_LOG_MEL + logf(f / MIN_LOG_HZ) / LOGSTEP;
}
/* audio/mel.rs:274-285 */
float orc_mel_to_hz(float m) {
    const float F_SP = 200.0f / 3.0f, MIN_LOG_HZ = 1000.0f, MIN_LOG_MEL = MIN_LOG_HZ / F_SP, LOGSTEP = 0.06875174f;
    if (m < MIN_LOG_MEL) return m * F_SP;
    return MIN_LOG_HZ * expf((m - MIN_LOG_MEL) * LOGSTEP);
}
/* audio/mel.rs:288-339 (sample_rate 16000, n_fft 400, 128 mels, fmin 0, fmax 8000) */
void orc_mel_filterbank(float* fb) {
    float mel_min = orc_hz_to_mel(0.0f), mel_max = orc_hz_to_mel(8000.0f);
    float hz[N_MELS + 2], freqs[N_FREQ];
    for (int i = 0; i <= N_MELS + 1; i++) {
        float m = mel_min + (mel_max - mel_min) * (float)i / (float)(N_MELS + 1);
        hz[i] = orc_mel_to_hz(m);
    }
    for (int j = 0; j < N_FREQ; j++) freqs[j] = (float)j * 16000.0f / 400.0f;
    memset(fb, 0, sizeof(float) * N_MELS * N_FREQ);
    for (int i = 0; i < N_MELS; i++) {
        float lo = hz[i], ce = hz[i + 1], up = hz[i + 2];
        for (int j = 0; j < N_FREQ; j++) {
            float fr = freqs[j];
            if (fr >= lo && fr <= ce && ce > lo) fb[i * N_FREQ + j] = (fr - lo) / (ce - lo);
            else if (fr > ce && fr <= up && up > ce) fb[i * N_FREQ + j] = (up - fr) / (up - ce);
        }
        float bw = hz[i + 2] - hz[i];
        if (bw > 0.0f) { float en = 2.0f / bw; for (int j = 0; j < N_FREQ; j++) fb[i * N_FREQ + j] *= en; }
    }
}
/* audio/mel.rs:175-182 */
size_t orc_mel_num_frames(size_t n) { return (n + 2 * (N_FFT / 2) - N_FFT) / HOP; }

/* audio/mel.rs:185-244 (stft) + :107-117 (power) + :247-257 (filterbank).
 * The 400-point transform is rustfft in the reference (f32, algorithm unpinned);
 * here it is a direct DFT accumulated in f64 and rounded once to f32. */
void orc_mel_compute(const float* samples, size_t n, float* out) {
    const size_t pad = N_FFT / 2, plen = n + 2 * pad;
    float* padded = (float*)malloc(plen * sizeof(float));
    size_t w = 0;
    for (size_t i = pad; i >= 1; i--) {                     /* mel.rs:196-199 */
        size_t lim = n > 0 ? n - 1 : 0, idx = i < lim ? i : lim;
        padded[w++] = idx < n ? samples[idx] : 0.0f;
    }
    memcpy(padded + w, samples, n * sizeof(float)); w += n;
    for (size_t i = 0; i < pad; i++) {                      /* mel.rs:202-205 */
        size_t a = n >= 2 ? n - 2 : 0, idx = a >= i ? a - i : 0;
        padded[w++] = idx < n ? samples[idx] : 0.0f;
    }
    float window[N_FFT]; orc_hann_window(N_FFT, window);
    float* fb = (float*)malloc(sizeof(float) * N_MELS * N_FREQ); orc_mel_filterbank(fb);
    double ct[N_FFT], st[N_FFT];
    for (int m = 0; m < N_FFT; m++) { ct[m] = cos(2.0 * M_PI * m / N_FFT); st[m] = sin(2.0 * M_PI * m / N_FFT); }
    const size_t n_frames = (plen - N_FFT) / HOP;
#pragma omp parallel for schedule(static)
    for (long fi = 0; fi < (long)n_frames; fi++) {
        float buf[N_FFT], power[N_FREQ];
        const float* src = padded + (size_t)fi * HOP;
        for (int j = 0; j < N_FFT; j++) buf[j] = src[j] * window[j];
        for (int f = 0; f < N_FREQ; f++) {
            double re = 0.0, im = 0.0; int ph = 0;
            for (int j = 0; j < N_FFT; j++) {
                re += (double)buf[j] * ct[ph]; im -= (double)buf[j] * st[ph];
                ph += f; if (ph >= N_FFT) ph -= N_FFT;
            }
            float fre = (float)re, fim = (float)im;
            power[f] = fre * fre + fim * fim;                /* Complex::norm_sqr */
        }
        for (int m = 0; m < N_MELS; m++) {
            float acc = 0.0f; const float* row = fb + m * N_FREQ;
            for (int j = 0; j < N_FREQ; j++) acc += row[j] * power[j];
            out[(size_t)fi * N_MELS + m] = acc;
        }
    }
    free(fb); free(padded);
}
/* audio/mel.rs:128-165 */
void orc_mel_compute_log(const float* samples, size_t n, float* out) {
    orc_mel_compute(samples, n, out);
    size_t tot = orc_mel_num_frames(n) * N_MELS;
    const float min_val = 1.5f - 8.0f;
    for (size_t i = 0; i < tot; i++) {
        float v = log10f(fmaxf(out[i], 1e-10f));
        v = fmaxf(v, min_val);
        out[i] = (v + 4.0f) / 4.0f;
    }
}

/* models/time_embedding.rs:41-71 */
void orc_time_embedding(float t, int dim, float theta, float* out) {
    int half = dim / 2; float log_theta = logf(theta);
    for (int i = 0; i < half; i++) {
        float freq = expf(-log_theta * (float)i / (float)half);
        float ang = t * freq;
        out[i] = cosf(ang); out[half + i] = sinf(ang);
    }
}

/* ------------------------------------------------------------- q4 codec */

/* gguf/tests.rs:24-57 */
void orc_q4_quantize(const float* data, size_t n, uint8_t* out) {
    size_t nb = n / 32;
#pragma omp parallel for schedule(static) if (nb > 4096)
    for (long b = 0; b < (long)nb; b++) {
        const float* blk = data + (size_t)b * 32; uint8_t* o = out + (size_t)b * 18;
        float amax = 0.0f; for (int i = 0; i < 32; i++) { float a = fabsf(blk[i]); if (a > amax) amax = a; }
        float d = amax / 7.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
        uint16_t h = f32_to_f16(d); o[0] = (uint8_t)(h & 0xff); o[1] = (uint8_t)(h >> 8);
        for (int i = 0; i < 16; i++) {
            float a = blk[i] * id + 8.5f, c = blk[i + 16] * id + 8.5f;
            /* Rust `as u8`: saturating, truncating, NaN -> 0 */
            int q0 = a != a ? 0 : a <= 0.0f ? 0 : a >= 255.0f ? 255 : (int)a;
            int q1 = c != c ? 0 : c <= 0.0f ? 0 : c >= 255.0f ? 255 : (int)c;
            if (q0 > 15) q0 = 15; if (q1 > 15) q1 = 15;
            o[2 + i] = (uint8_t)(q0 | (q1 << 4));
        }
    }
}
/* gguf/tensor.rs:88-113 */
void orc_q4_dequantize(const uint8_t* raw, size_t n, float* out) {
    size_t nb = n / 32;
#pragma omp parallel for schedule(static) if (nb > 4096)
    for (long b = 0; b < (long)nb; b++) {
        const uint8_t* p = raw + (size_t)b * 18; float* o = out + (size_t)b * 32;
        float d = f16_to_f32((uint16_t)(p[0] | (p[1] << 8)));
        for (int i = 0; i < 16; i++) {
            uint8_t by = p[2 + i];
            o[i] = ((float)(by & 0x0f) - 8.0f) * d;
            o[i + 16] = ((float)((by >> 4) & 0x0f) - 8.0f) * d;
        }
    }
}
/* gguf/tests.rs:172-185 */
void orc_reference_matmul(const float* a, const float* bt, int m, int k, int n, float* out) {
#pragma omp parallel for schedule(static) if ((long)m * n * k > (1 << 20))
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            float acc = 0.0f;
            for (int l = 0; l < k; l++) acc += a[(size_t)i * k + l] * bt[(size_t)j * k + l];
            out[(size_t)i * n + j] = acc;
        }
}

/* A linear layer: weight [N][K] either raw Q4_0 blocks or dense f32; optional bias. */
typedef struct { const uint8_t* q4; const float* dense; int64_t N, K; const float* bias; } lin_t;

typedef float v8f __attribute__((vector_size(32), aligned(4)));

/* out[M][N] = x[M][K] * W^T (+bias). Per output element: sequential-k f32 sum (mul then add),
 * identical value to orc_reference_matmul on the dequantised weights; vectorised across 8
 * output rows n (each lane keeps its own sequential chain). */
static void linear_fwd(const lin_t* L, const float* x, int64_t M, float* out) {
    const int64_t N = L->N, K = L->K;
    const int64_t ntile = (N + 7) / 8;
#pragma omp parallel
    {
        float* tile = (float*)aligned_alloc(64, (size_t)K * 8 * sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < ntile; t++) {
            const int64_t n0 = t * 8, nr = (N - n0) < 8 ? (N - n0) : 8;
            if (nr < 8) memset(tile, 0, (size_t)K * 8 * sizeof(float));
            for (int r = 0; r < nr; r++) {
                if (L->q4) {
                    const uint8_t* row = L->q4 + (size_t)(n0 + r) * (K / 32) * 18;
                    for (int64_t b = 0; b < K / 32; b++) {
                        const uint8_t* p = row + b * 18; float d = f16_to_f32((uint16_t)(p[0] | (p[1] << 8)));
                        float* tk = tile + (size_t)b * 32 * 8 + r;
                        for (int i = 0; i < 16; i++) {
                            uint8_t by = p[2 + i];
                            tk[(size_t)i * 8] = ((float)(by & 0x0f) - 8.0f) * d;
                            tk[(size_t)(i + 16) * 8] = ((float)((by >> 4) & 0x0f) - 8.0f) * d;
                        }
                    }
                } else {
                    const float* row = L->dense + (size_t)(n0 + r) * K;
                    for (int64_t k = 0; k < K; k++) tile[(size_t)k * 8 + r] = row[k];
                }
            }
            int64_t m = 0;
            for (; m + 4 <= M; m += 4) {
                v8f a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
                const float *x0 = x + (size_t)m * K, *x1 = x0 + K, *x2 = x1 + K, *x3 = x2 + K;
                for (int64_t k = 0; k < K; k++) {
                    v8f w = *(const v8f*)(tile + (size_t)k * 8);
                    a0 = a0 + w * x0[k]; a1 = a1 + w * x1[k]; a2 = a2 + w * x2[k]; a3 = a3 + w * x3[k];
                }
                for (int r = 0; r < nr; r++) {
                    float b = L->bias ? L->bias[n0 + r] : 0.0f;
                    out[(size_t)(m + 0) * N + n0 + r] = L->bias ? a0[r] + b : a0[r];
                    out[(size_t)(m + 1) * N + n0 + r] = L->bias ? a1[r] + b : a1[r];
                    out[(size_t)(m + 2) * N + n0 + r] = L->bias ? a2[r] + b : a2[r];
                    out[(size_t)(m + 3) * N + n0 + r] = L->bias ? a3[r] + b : a3[r];
                }
            }
            for (; m < M; m++) {
                v8f a0 = {0}; const float* x0 = x + (size_t)m * K;
                for (int64_t k = 0; k < K; k++) a0 = a0 + *(const v8f*)(tile + (size_t)k * 8) * x0[k];
                for (int r = 0; r < nr; r++)
                    out[(size_t)m * N + n0 + r] = L->bias ? a0[r] + L->bias[n0 + r] : a0[r];
            }
        }
        free(tile);
    }
}

void orc_q4_matmul(const uint8_t* w, int64_t N, int64_t K, const float* x, int64_t BM, const float* bias, float* out) {
    lin_t L = {w, NULL, N, K, bias};
    linear_fwd(&L, x, BM, out);
}

/* ---------------------------------------------------------- gguf reader */
/* gguf/reader.rs:13-223 */
typedef struct { char* name; uint32_t ndims; uint64_t dims[4]; uint32_t dtype; uint64_t offset, nbytes; } ginfo_t;
struct orc_gguf { uint8_t* map; size_t size; uint32_t version; uint64_t n; ginfo_t* t; uint64_t data_off; };

typedef struct { const uint8_t* p; size_t pos, size; int bad; } cur_t;
static uint64_t rd(cur_t* c, int nbytes) {
    if (c->pos + nbytes > c->size) { c->bad = 1; return 0; }
    uint64_t v = 0; memcpy(&v, c->p + c->pos, nbytes); c->pos += nbytes; return v;
}
static char* rd_str(cur_t* c) {
    uint64_t len = rd(c, 8);
    if (c->bad || c->pos + len > c->size) { c->bad = 1; return NULL; }
    char* s = (char*)malloc(len + 1); memcpy(s, c->p + c->pos, len); s[len] = 0; c->pos += len; return s;
}
/* gguf/reader.rs:327-376 */
static void skip_val(cur_t* c, uint32_t ty, int depth) {
    switch (ty) {
    case 0: case 1: case 7: c->pos += 1; break;
    case 2: case 3: c->pos += 2; break;
    case 4: case 5: case 6: c->pos += 4; break;
    case 8: { char* s = rd_str(c); free(s); break; }
    case 9: { uint32_t et = (uint32_t)rd(c, 4); uint64_t cnt = rd(c, 8);
              for (uint64_t i = 0; i < cnt && !c->bad; i++) skip_val(c, et, depth + 1); break; }
    case 10: case 11: case 12: c->pos += 8; break;
    default: c->bad = 2; break;
    }
    if (c->pos > c->size) c->bad = 1;
}
/* gguf/reader.rs:37-48 */
static uint64_t dtype_bytes(uint32_t dt, uint64_t n) { return dt == 0 ? n * 4 : dt == 1 ? n * 2 : (n / 32) * 18; }

orc_gguf* orc_gguf_open(const char* path) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { FAIL("open %s: %s", path, strerror(errno)); return NULL; }
    struct stat st; fstat(fd, &st);
    uint8_t* map = (uint8_t*)mmap(NULL, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0); close(fd);
    if (map == MAP_FAILED) { FAIL("mmap %s failed", path); return NULL; }
    cur_t c = {map, 0, (size_t)st.st_size, 0};
    uint32_t magic = (uint32_t)rd(&c, 4);
    if (c.bad || magic != 0x46554747u) { FAIL("Invalid GGUF magic: 0x%08X", magic); munmap(map, st.st_size); return NULL; }
    uint32_t ver = (uint32_t)rd(&c, 4);
    if (ver != 2 && ver != 3) { FAIL("Unsupported GGUF version: %u (expected 2 or 3)", ver); munmap(map, st.st_size); return NULL; }
    uint64_t nt = rd(&c, 8), nkv = rd(&c, 8);
    for (uint64_t i = 0; i < nkv && !c.bad; i++) { char* k = rd_str(&c); free(k); uint32_t ty = (uint32_t)rd(&c, 4); skip_val(&c, ty, 0); }
    if (c.bad) { FAIL("Failed to parse GGUF metadata"); munmap(map, st.st_size); return NULL; }
    orc_gguf* g = (orc_gguf*)calloc(1, sizeof *g);
    g->map = map; g->size = st.st_size; g->version = ver; g->n = nt; g->t = (ginfo_t*)calloc(nt ? nt : 1, sizeof(ginfo_t));
    for (uint64_t i = 0; i < nt; i++) {
        ginfo_t* t = &g->t[i];
        t->name = rd_str(&c); t->ndims = (uint32_t)rd(&c, 4);
        if (c.bad || t->ndims > 4) { c.bad = 1; break; }
        uint64_t ne = 1; for (uint32_t d = 0; d < t->ndims; d++) { t->dims[d] = rd(&c, 8); ne *= t->dims[d]; }
        t->dtype = (uint32_t)rd(&c, 4); t->offset = rd(&c, 8);
        if (t->dtype > 2) { FAIL("Unsupported GGML dtype code: %u", t->dtype); c.bad = 3; break; }
        t->nbytes = dtype_bytes(t->dtype, ne);
    }
    if (c.bad) { if (c.bad != 3) FAIL("Failed to parse GGUF tensor index"); orc_gguf_close(g); return NULL; }
    g->data_off = (c.pos + 31) / 32 * 32;
    for (uint64_t i = 0; i < nt; i++)
        if (g->data_off + g->t[i].offset + g->t[i].nbytes > g->size) { FAIL("Tensor '%s' exceeds file size", g->t[i].name); orc_gguf_close(g); return NULL; }
    return g;
}
void orc_gguf_close(orc_gguf* g) {
    if (!g) return;
    for (uint64_t i = 0; i < g->n; i++) free(g->t[i].name);
    free(g->t); munmap(g->map, g->size); free(g);
}
uint32_t orc_gguf_version(const orc_gguf* g) { return g->version; }
uint64_t orc_gguf_tensor_count(const orc_gguf* g) { return g->n; }
const char* orc_gguf_tensor_name(const orc_gguf* g, uint64_t i) { return i < g->n ? g->t[i].name : NULL; }
static const ginfo_t* gfind(const orc_gguf* g, const char* name) {
    for (uint64_t i = 0; i < g->n; i++) if (!strcmp(g->t[i].name, name)) return &g->t[i];
    return NULL;
}
int orc_gguf_tensor_info(const orc_gguf* g, const char* name, uint64_t dims[4], uint32_t* nd, uint32_t* dt, uint64_t* nb) {
    const ginfo_t* t = gfind(g, name); if (!t) return -1;
    for (uint32_t d = 0; d < t->ndims; d++) dims[d] = t->dims[d];
    *nd = t->ndims; *dt = t->dtype; *nb = t->nbytes; return 0;
}
const void* orc_gguf_tensor_data(const orc_gguf* g, const char* name) {
    const ginfo_t* t = gfind(g, name); return t ? g->map + g->data_off + t->offset : NULL;
}

/* --------------------------------------------------------------- layers */

float orc_gelu(float x) { return x * 0.5f * (1.0f + erff(x / 1.41421356237309504880f)); } /* burn gelu: erf form */
float orc_silu(float x) { return x / (1.0f + expf(-x)); }                                   /* x * sigmoid(x) */

/* burn::nn::RmsNorm::forward: rms = sqrt(mean(x^2) + eps); (x / rms) * gamma */
void orc_rms_norm(const float* x, int rows, int dim, const float* gamma, float eps, float* out) {
#pragma omp parallel for schedule(static) if (rows > 8)
    for (int r = 0; r < rows; r++) {
        const float* xr = x + (size_t)r * dim; float* o = out + (size_t)r * dim;
        float ss = 0.0f; for (int i = 0; i < dim; i++) ss += xr[i] * xr[i];
        float rms = sqrtf(ss / (float)dim + eps);
        for (int i = 0; i < dim; i++) o[i] = (xr[i] / rms) * gamma[i];
    }
}

/* models/layers/rope.rs:35-64 tables; :99-141 rotation */
void orc_rope(float* x, int seq, int heads, int hd, int offset, float theta) {
    int half = hd / 2;
    for (int s = 0; s < seq; s++) {
        float pos = (float)(offset + s);
        for (int j = 0; j < half; j++) {
            float inv = 1.0f / powf(theta, (float)(2 * j) / (float)hd);
            float fr = pos * inv, c = cosf(fr), sn = sinf(fr);
            for (int h = 0; h < heads; h++) {
                float* p = x + ((size_t)s * heads + h) * hd + 2 * j;
                float xr = p[0], xi = p[1];
                p[0] = xr * c - xi * sn; p[1] = xr * sn + xi * c;
            }
        }
    }
}

/* models/layers/conv.rs:47-48 */
int orc_conv_out_len(int L) { return (L + 2 * 1 - 3) / 2 + 1; }
/* models/layers/conv.rs:78-83, one stage. Sum order: bias first?  burn conv1d adds bias after the
 * reduction; here: acc over (ci, kk) sequentially then + bias. */
void orc_conv1d_gelu(const float* in, int Cin, int L, const float* w, const float* b, int Cout, float* out) {
    int Lo = orc_conv_out_len(L);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; co++) {
        const float* wc = w + (size_t)co * Cin * 3;
        for (int t = 0; t < Lo; t++) {
            float acc = 0.0f; int base = t * 2 - 1;
            for (int ci = 0; ci < Cin; ci++) {
                const float* row = in + (size_t)ci * L;
                for (int kk = 0; kk < 3; kk++) { int p = base + kk; if (p >= 0 && p < L) acc += row[p] * wc[ci * 3 + kk]; }
            }
            out[(size_t)co * Lo + t] = orc_gelu(acc + (b ? b[co] : 0.0f));
        }
    }
}

/* gguf/model.rs:100-120 / :149-172 with masks of models/layers/masking.rs */
void orc_attention(const float* q, const float* k, const float* v, int q_len, int kv_len, int n_heads, int n_kv, int hd,
                   int offset, int causal, int window, float* out) {
    const float scale = powf((float)hd, -0.5f);  /* gguf/model.rs:65 */
    const int rep = n_heads / n_kv;              /* gguf/model.rs:177-197: q-head h uses kv-head h/rep */
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)kv_len);
#pragma omp for collapse(2) schedule(static)
        for (int h = 0; h < n_heads; h++)
            for (int i = 0; i < q_len; i++) {
                const int hk = h / rep, pos = offset + i;
                const float* qi = q + ((size_t)i * n_heads + h) * hd;
                float mx = -INFINITY;
                for (int j = 0; j < kv_len; j++) {
                    const float* kj = k + ((size_t)j * n_kv + hk) * hd;
                    float s = 0.0f; for (int d = 0; d < hd; d++) s += qi[d] * kj[d];
                    s *= scale;
                    if (causal && j > pos) s = -INFINITY;
                    if (window >= 0 && abs(pos - j) > window) s = -INFINITY;
                    sc[j] = s; if (s > mx) mx = s;
                }
                float sum = 0.0f;
                for (int j = 0; j < kv_len; j++) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                float* o = out + ((size_t)i * n_heads + h) * hd;
                for (int d = 0; d < hd; d++) o[d] = 0.0f;
                for (int j = 0; j < kv_len; j++) {
                    float p = sc[j] / sum; const float* vj = v + ((size_t)j * n_kv + hk) * hd;
                    for (int d = 0; d < hd; d++) o[d] += p * vj[d];
                }
            }
        free(sc);
    }
}

/* ---------------------------------------------------------------- model */

typedef struct { float *attn_norm, *ffn_norm; lin_t wq, wk, wv, wo, w1, w2, w3; } enc_layer_t;
typedef struct { float *attn_norm, *ffn_norm; lin_t wq, wk, wv, wo, w1, w2, w3, ada0, ada2; } dec_layer_t;
struct orc_model {
    orc_gguf* g; orc_model_cfg cfg;
    float *conv1_w, *conv1_b, *conv2_w, *conv2_b, *enc_norm, *dec_norm;
    enc_layer_t* enc; dec_layer_t* dec; lin_t ad0, ad2, tok;
    float** owned; int n_owned, cap_owned;
};
struct orc_cache { int layers, n_kv, hd, max_seq, len; float* k; float* v; }; /* [layer][max_seq][n_kv][hd] */

#define ENC "mm_streams_embeddings.embedding_module.whisper_encoder"   /* models/weights.rs:221 */
#define EMB "mm_streams_embeddings.embedding_module"
#define TOK EMB ".tok_embeddings.weight"                                /* models/weights.rs:225 */
#define ADP EMB ".audio_language_projection"                            /* models/weights.rs:227 */

static float* own(orc_model* m, float* p) {
    if (m->n_owned == m->cap_owned) { m->cap_owned = m->cap_owned ? m->cap_owned * 2 : 256; m->owned = (float**)realloc(m->owned, sizeof(float*) * m->cap_owned); }
    m->owned[m->n_owned++] = p; return p;
}
/* gguf/loader.rs:443-474 load_f32_tensor (F32 or F16 -> f32) */
static float* load_f32(orc_model* m, const char* name, int required) {
    const ginfo_t* t = gfind(m->g, name);
    if (!t) { if (required) FAIL("Tensor '%s' not found", name); return NULL; }
    if (t->dtype == 2) { FAIL("Cannot load Q4_0 tensor '%s' as f32", name); return NULL; }
    uint64_t ne = 1; for (uint32_t d = 0; d < t->ndims; d++) ne *= t->dims[d];
    float* out = (float*)malloc(ne * sizeof(float)); const uint8_t* src = m->g->map + m->g->data_off + t->offset;
    if (t->dtype == 0) memcpy(out, src, ne * 4);
    else {
        _Pragma("omp parallel for schedule(static)")
        for (int64_t i = 0; i < (int64_t)ne; i++) { uint16_t h; memcpy(&h, src + 2 * i, 2); out[i] = f16_to_f32(h); }
    }
    return own(m, out);
}
/* gguf/loader.rs:385-441 load_q4_linear(_with_optional_bias); dims reversed (:497-499) */
static int load_lin(orc_model* m, const char* name, const char* bias_name, lin_t* L) {
    const ginfo_t* t = gfind(m->g, name);
    if (!t) { FAIL("Tensor '%s' not found", name); return -1; }
    if (t->ndims != 2) { FAIL("Tensor '%s' is not 2-D", name); return -1; }
    L->K = (int64_t)t->dims[0]; L->N = (int64_t)t->dims[1]; L->q4 = NULL; L->dense = NULL; L->bias = NULL;
    if (t->dtype == 2) {
        if ((L->N * L->K) % 32) { FAIL("Q4_0 requires element count divisible by 32"); return -1; }
        L->q4 = m->g->map + m->g->data_off + t->offset;
    } else { L->dense = load_f32(m, name, 1); if (!L->dense) return -1; }
    if (bias_name && gfind(m->g, bias_name)) L->bias = load_f32(m, bias_name, 1);
    return 0;
}

orc_model* orc_model_load_gguf(const char* path) {
    orc_gguf* g = orc_gguf_open(path); if (!g) return NULL;
    orc_model* m = (orc_model*)calloc(1, sizeof *m); m->g = g;
    orc_model_cfg* c = &m->cfg; char nm[256], nb[256]; int bad = 0;
    /* models/config.rs:441-493 defaults that are not derivable from shapes */
    c->enc_head_dim = 64; c->dec_head_dim = 128; c->enc_window = 750; c->dec_window = 8192;
    c->rope_theta = 1e6f; c->norm_eps = 1e-5f; c->reshape_factor = 4;
    for (c->enc_layers = 0;; c->enc_layers++) { snprintf(nm, sizeof nm, ENC ".transformer.layers.%d.attention.wq.weight", c->enc_layers); if (!gfind(g, nm)) break; }
    for (c->dec_layers = 0;; c->dec_layers++) { snprintf(nm, sizeof nm, "layers.%d.attention.wq.weight", c->dec_layers); if (!gfind(g, nm)) break; }
    m->enc = (enc_layer_t*)calloc(c->enc_layers ? c->enc_layers : 1, sizeof(enc_layer_t));
    m->dec = (dec_layer_t*)calloc(c->dec_layers ? c->dec_layers : 1, sizeof(dec_layer_t));
    /* gguf/loader.rs:263-275 */
    m->conv1_w = load_f32(m, ENC ".conv_layers.0.conv.weight", 1); m->conv1_b = load_f32(m, ENC ".conv_layers.0.conv.bias", 1);
    m->conv2_w = load_f32(m, ENC ".conv_layers.1.conv.weight", 1); m->conv2_b = load_f32(m, ENC ".conv_layers.1.conv.bias", 1);
    if (!m->conv1_w || !m->conv1_b || !m->conv2_w || !m->conv2_b) bad = 1;
    if (!bad) { const ginfo_t* t = gfind(g, ENC ".conv_layers.0.conv.weight"); c->n_mels = (int)t->dims[1]; c->enc_dim = (int)t->dims[2]; }
    /* gguf/loader.rs:215-260 */
    for (int i = 0; i < c->enc_layers && !bad; i++) {
        enc_layer_t* L = &m->enc[i];
#define EN(s) (snprintf(nm, sizeof nm, ENC ".transformer.layers.%d." s, i), nm)
#define EB(s) (snprintf(nb, sizeof nb, ENC ".transformer.layers.%d." s, i), nb)
        L->attn_norm = load_f32(m, EN("attention_norm.weight"), 1); L->ffn_norm = load_f32(m, EN("ffn_norm.weight"), 1);
        bad |= !L->attn_norm || !L->ffn_norm;
        bad |= load_lin(m, EN("attention.wq.weight"), EB("attention.wq.bias"), &L->wq);
        bad |= load_lin(m, EN("attention.wk.weight"), NULL, &L->wk);
        bad |= load_lin(m, EN("attention.wv.weight"), EB("attention.wv.bias"), &L->wv);
        bad |= load_lin(m, EN("attention.wo.weight"), EB("attention.wo.bias"), &L->wo);
        bad |= load_lin(m, EN("feed_forward.w1.weight"), NULL, &L->w1);
        bad |= load_lin(m, EN("feed_forward.w2.weight"), EB("feed_forward.w2.bias"), &L->w2);
        bad |= load_lin(m, EN("feed_forward.w3.weight"), NULL, &L->w3);
    }
    m->enc_norm = load_f32(m, ENC ".transformer.norm.weight", 1); bad |= !m->enc_norm;
    /* gguf/loader.rs:378-383 */
    bad |= load_lin(m, ADP ".0.weight", NULL, &m->ad0); bad |= load_lin(m, ADP ".2.weight", NULL, &m->ad2);
    /* gguf/loader.rs:305-326 (tok embeddings; dequantised on use) */
    bad |= load_lin(m, TOK, NULL, &m->tok);
    /* gguf/loader.rs:329-375 */
    for (int i = 0; i < c->dec_layers && !bad; i++) {
        dec_layer_t* L = &m->dec[i];
#define DN(s) (snprintf(nm, sizeof nm, "layers.%d." s, i), nm)
        bad |= load_lin(m, DN("ada_rms_norm_t_cond.0.weight"), NULL, &L->ada0);
        bad |= load_lin(m, DN("ada_rms_norm_t_cond.2.weight"), NULL, &L->ada2);
        L->attn_norm = load_f32(m, DN("attention_norm.weight"), 1); L->ffn_norm = load_f32(m, DN("ffn_norm.weight"), 1);
        bad |= !L->attn_norm || !L->ffn_norm;
        bad |= load_lin(m, DN("attention.wq.weight"), NULL, &L->wq); bad |= load_lin(m, DN("attention.wk.weight"), NULL, &L->wk);
        bad |= load_lin(m, DN("attention.wv.weight"), NULL, &L->wv); bad |= load_lin(m, DN("attention.wo.weight"), NULL, &L->wo);
        bad |= load_lin(m, DN("feed_forward.w1.weight"), NULL, &L->w1); bad |= load_lin(m, DN("feed_forward.w2.weight"), NULL, &L->w2);
        bad |= load_lin(m, DN("feed_forward.w3.weight"), NULL, &L->w3);
    }
    m->dec_norm = load_f32(m, "norm.weight", 1); bad |= !m->dec_norm;
    if (bad || c->enc_layers == 0 || c->dec_layers == 0) { if (!bad) FAIL("GGUF has no encoder/decoder layers"); orc_model_free(m); return NULL; }
    c->enc_heads = (int)(m->enc[0].wq.N / c->enc_head_dim); c->enc_ffn = (int)m->enc[0].w1.N;
    c->dec_dim = (int)m->dec[0].wq.K; c->dec_heads = (int)(m->dec[0].wq.N / c->dec_head_dim);
    c->dec_kv_heads = (int)(m->dec[0].wk.N / c->dec_head_dim); c->dec_ffn = (int)m->dec[0].w1.N;
    c->vocab = (int)m->tok.N; c->t_cond_dim = (int)m->dec[0].ada0.N;
    if (m->ad0.K != (int64_t)c->enc_dim * c->reshape_factor || m->ad2.N != c->dec_dim || m->tok.K != c->dec_dim) {
        FAIL("inconsistent adapter / embedding shapes"); orc_model_free(m); return NULL; }
    return m;
}
void orc_model_free(orc_model* m) {
    if (!m) return;
    for (int i = 0; i < m->n_owned; i++) free(m->owned[i]);
    free(m->owned); free(m->enc); free(m->dec); orc_gguf_close(m->g); free(m);
}
void orc_model_config(const orc_model* m, orc_model_cfg* out) { *out = m->cfg; }

int orc_enc_seq_len(const orc_model* m, int T) { (void)m; return orc_conv_out_len(orc_conv_out_len(T)); }

/* models/layers/conv.rs:78-83 + gguf/model.rs:426-427 (swap_dims) */
void orc_encoder_conv(const orc_model* m, const float* mel, int T, float* out) {
    const orc_model_cfg* c = &m->cfg; int T1 = orc_conv_out_len(T), T2 = orc_conv_out_len(T1), D = c->enc_dim;
    float* a = (float*)malloc(sizeof(float) * (size_t)D * T1); float* b = (float*)malloc(sizeof(float) * (size_t)D * T2);
    orc_conv1d_gelu(mel, c->n_mels, T, m->conv1_w, m->conv1_b, D, a);
    orc_conv1d_gelu(a, D, T1, m->conv2_w, m->conv2_b, D, b);
    for (int t = 0; t < T2; t++) for (int d = 0; d < D; d++) out[(size_t)t * D + d] = b[(size_t)d * T2 + t];
    free(a); free(b);
}

/* gguf/model.rs:220-224 */
static void swiglu(const lin_t* w1, const lin_t* w2, const lin_t* w3, const float* x, int M, float* out) {
    int64_t F = w1->N; float* g = (float*)malloc(sizeof(float) * (size_t)M * F); float* u = (float*)malloc(sizeof(float) * (size_t)M * F);
    linear_fwd(w1, x, M, g); linear_fwd(w3, x, M, u);
    for (size_t i = 0; i < (size_t)M * F; i++) g[i] = orc_silu(g[i]) * u[i];
    linear_fwd(w2, g, M, out); free(g); free(u);
}

/* gguf/model.rs:287-297 (+ :77-122 attention, offset 0, causal, window 750) */
void orc_encoder_layer(const orc_model* m, int li, float* x, int S) {
    const orc_model_cfg* c = &m->cfg; const enc_layer_t* L = &m->enc[li]; int D = c->enc_dim, H = c->enc_heads, hd = c->enc_head_dim;
    size_t sd = (size_t)S * D, sh = (size_t)S * H * hd;
    float *xn = (float*)malloc(sizeof(float) * sd), *q = (float*)malloc(sizeof(float) * sh), *k = (float*)malloc(sizeof(float) * sh),
          *v = (float*)malloc(sizeof(float) * sh), *at = (float*)malloc(sizeof(float) * sh), *o = (float*)malloc(sizeof(float) * sd);
    orc_rms_norm(x, S, D, L->attn_norm, c->norm_eps, xn);
    linear_fwd(&L->wq, xn, S, q); linear_fwd(&L->wk, xn, S, k); linear_fwd(&L->wv, xn, S, v);
    orc_rope(q, S, H, hd, 0, c->rope_theta); orc_rope(k, S, H, hd, 0, c->rope_theta);
    orc_attention(q, k, v, S, S, H, H, hd, 0, 1, c->enc_window, at);
    linear_fwd(&L->wo, at, S, o);
    for (size_t i = 0; i < sd; i++) x[i] = o[i] + x[i];          /* x + residual */
    orc_rms_norm(x, S, D, L->ffn_norm, c->norm_eps, xn);
    swiglu(&L->w1, &L->w2, &L->w3, xn, S, o);
    for (size_t i = 0; i < sd; i++) x[i] = o[i] + x[i];
    free(xn); free(q); free(k); free(v); free(at); free(o);
}
void orc_encoder_final_norm(const orc_model* m, float* x, int S) {
    float* t = (float*)malloc(sizeof(float) * (size_t)S * m->cfg.enc_dim);
    orc_rms_norm(x, S, m->cfg.enc_dim, m->enc_norm, m->cfg.norm_eps, t);
    memcpy(x, t, sizeof(float) * (size_t)S * m->cfg.enc_dim); free(t);
}

/* gguf/model.rs:783-788; reshape: models/adapter.rs:108-122; adapter: gguf/model.rs:745-749 */
int orc_encode_audio(const orc_model* m, const float* mel, int T, float* out) {
    const orc_model_cfg* c = &m->cfg; int S = orc_enc_seq_len(m, T), D = c->enc_dim;
    float* x = (float*)malloc(sizeof(float) * (size_t)S * D);
    orc_encoder_conv(m, mel, T, x);
    for (int l = 0; l < c->enc_layers; l++) orc_encoder_layer(m, l, x, S);
    orc_encoder_final_norm(m, x, S);
    int S4 = S / c->reshape_factor;                 /* rows beyond S4*4 dropped; reshape is a no-op on row-major data */
    if (S4 > 0) {
        float* h = (float*)malloc(sizeof(float) * (size_t)S4 * m->ad0.N);
        linear_fwd(&m->ad0, x, S4, h);
        for (size_t i = 0; i < (size_t)S4 * m->ad0.N; i++) h[i] = orc_gelu(h[i]);
        linear_fwd(&m->ad2, h, S4, out); free(h);
    }
    free(x); return S4;
}

/* ---- sample-rate conversion (audio/resample.rs:16-52).  The reference calls rubato 1.0 (Cargo.toml:41 `rubato = "1.0"`; no Cargo.lock in the tree):
 *   Fft::<f32>::new(sr_in, sr_out, 1024, 2, 1, FixedSync::Input)  +  process_all_into_buffer(.., n_in, None)        (resample.rs:22-45)
 * rubato is a third-party crate that is NOT in /root/reference, so what follows is a restatement of its PUBLISHED algorithm (the crate's synchronous FFT
 * resampler, src/synchro.rs + src/sinc.rs + src/windows.rs as released), not of code in the tree; PARITY UNPINNED -- no rubato output exists here to pin it
 * with; the anchors are the reference's call site above and its contract tests (resample.rs:56-108: same rate = clone, length within 100 samples, duration).
 *   plan      gcd = gcd(sr_in, sr_out); fft_chunks = ceil(f32(1024) / f32(2) / f32(sr_in / gcd)); fft_in = fft_chunks * sr_in / gcd, fft_out = fft_chunks * sr_out / gcd
 *   filter    cutoff = 0.4f32 ^ (16 / fft_in) [* fft_out / fft_in when down-sampling]; taps h[x] = w[x] * sinc((x - fft_in / 2) * cutoff), x < fft_in, all in f32:
 *             w = (periodic 4-term Blackman-Harris)^2, sinc(0) = 1, fft_in / 2 the INTEGER half; normalised to unit sum, then / (2 fft_in)
 *   unit      one block of fft_in samples, zero-padded to 2 fft_in -> real FFT -> bins [0, new_len) times the filter's FFT, the rest dropped
 *             (new_len = fft_out when down-sampling, fft_in + 1 otherwise) -> UNNORMALISED inverse real FFT of length 2 fft_out (the imaginary part of bin 0 is ignored)
 *             -> first half + the previous block's second half is the block's fft_out output samples (overlap-add)
 *   all       the stream of blocks over the zero-extended input, minus the first output_delay = fft_out / 2 samples, cut to ceil(n_in * (f64(sr_out) / f64(sr_in))).
 * The transforms here are plain O(N * bins) DFT sums in double precision (rubato runs f32 FFTs: agreement to f32 rounding, not bits). */
typedef struct { long fft_in, fft_out, new_len, delay; float cutoff; } orc_rs_plan;
static orc_rs_plan orc_rs_make_plan(uint32_t sr_in, uint32_t sr_out) {
    long a = sr_in, b = sr_out; while (b) { long t = a % b; a = b; b = t; }
    orc_rs_plan p; const long min_in = sr_in / a, min_out = sr_out / a;
    const long chunks = (long)ceilf(1024.0f / 2.0f / (float)min_in);
    p.fft_in = chunks * min_in; p.fft_out = chunks * min_out;
    p.cutoff = powf(0.4f, 16.0f / (float)p.fft_in); if (p.fft_in > p.fft_out) p.cutoff = p.cutoff * (float)p.fft_out / (float)p.fft_in;
    p.new_len = p.fft_in < p.fft_out ? p.fft_in + 1 : p.fft_out; p.delay = p.fft_out / 2;
    return p;
}
static void orc_rs_taps(const orc_rs_plan* p, float* h) {         /* sinc.rs make_sincs(fft_in, 1, cutoff, BlackmanHarris2)[0] / (2 fft_in), f32 throughout */
    const long n = p->fft_in; const float np_f = (float)n, pi = 3.14159265358979323846f;
    float sum = 0.f;
    for (long x = 0; x < n; x++) {
        const float xf = (float)x;
        float w = 0.35875f - 0.48829f * cosf(2.0f * pi * xf / np_f) + 0.14128f * cosf(4.0f * pi * xf / np_f) - 0.01168f * cosf(6.0f * pi * xf / np_f);
        w = w * w;
        const float arg = (xf - (float)(n / 2)) * p->cutoff;
        const float sc = arg == 0.f ? 1.0f : sinf(arg * pi) / (arg * pi);
        h[x] = w * sc; sum += h[x];
    }
    for (long x = 0; x < n; x++) h[x] = h[x] / sum / (float)(2 * n);
}
size_t orc_resample_len(size_t n_in, uint32_t sr_in, uint32_t sr_out) { return sr_in == sr_out ? n_in : (size_t)ceil(((double)sr_out / (double)sr_in) * (double)n_in); }
void orc_resample_plan(uint32_t sr_in, uint32_t sr_out, long* fft_in, long* fft_out, long* delay, float* cutoff, float* taps_or_null) {
    const orc_rs_plan p = orc_rs_make_plan(sr_in, sr_out);
    *fft_in = p.fft_in; *fft_out = p.fft_out; *delay = p.delay; *cutoff = p.cutoff; if (taps_or_null) orc_rs_taps(&p, taps_or_null);
}
void orc_resample(const float* in, size_t n_in, uint32_t sr_in, uint32_t sr_out, float* out) {
    if (sr_in == sr_out) { memcpy(out, in, n_in * sizeof(float)); return; }
    const size_t n_out = orc_resample_len(n_in, sr_in, sr_out); if (!n_out) return;
    const orc_rs_plan p = orc_rs_make_plan(sr_in, sr_out);
    const long Ni = p.fft_in, No = p.fft_out, L = p.new_len, Pi = 2 * Ni, Po = 2 * No;
    float* h = (float*)malloc(sizeof(float) * (size_t)Ni); orc_rs_taps(&p, h);
    double *ci = (double*)malloc(sizeof(double) * Pi), *si = (double*)malloc(sizeof(double) * Pi), *co = (double*)malloc(sizeof(double) * Po), *so = (double*)malloc(sizeof(double) * Po);
    for (long j = 0; j < Pi; j++) { ci[j] = cos(2.0 * M_PI * (double)j / (double)Pi); si[j] = sin(2.0 * M_PI * (double)j / (double)Pi); }
    for (long j = 0; j < Po; j++) { co[j] = cos(2.0 * M_PI * (double)j / (double)Po); so[j] = sin(2.0 * M_PI * (double)j / (double)Po); }
    double *Hr = (double*)calloc((size_t)L, sizeof(double)), *Hi = (double*)calloc((size_t)L, sizeof(double));
    for (long k = 0; k < L; k++) for (long n = 0; n < Ni; n++) { const long j = (k * n) % Pi; Hr[k] += (double)h[n] * ci[j]; Hi[k] -= (double)h[n] * si[j]; }   /* filter_f = FFT(filter_t) */
    const long n_blocks = ((long)n_out + p.delay + No - 1) / No;                 /* blocks whose first half reaches the last kept output sample */
    double* stream = (double*)calloc((size_t)(n_blocks + 1) * (size_t)No, sizeof(double));
    double* ob = (double*)malloc(sizeof(double) * (size_t)n_blocks * (size_t)Po);
    #pragma omp parallel
    {
        double *Yr = (double*)malloc(sizeof(double) * (size_t)L), *Yi = (double*)malloc(sizeof(double) * (size_t)L);
        #pragma omp for schedule(static)
        for (long c = 0; c < n_blocks; c++) {
            for (long k = 0; k < L; k++) {                                        /* forward transform of the zero-padded block, kept bins only, times the filter */
                double xr = 0.0, xi = 0.0;
                for (long n = 0; n < Ni; n++) {
                    const size_t i = (size_t)c * (size_t)Ni + (size_t)n; if (i >= n_in) break;
                    const long j = (k * n) % Pi; xr += (double)in[i] * ci[j]; xi -= (double)in[i] * si[j];
                }
                Yr[k] = xr * Hr[k] - xi * Hi[k]; Yi[k] = xr * Hi[k] + xi * Hr[k];
            }
            for (long m = 0; m < Po; m++) {                                       /* unnormalised c2r inverse of length 2 fft_out: bin 0 real part only, bins 1 .. L-1 twice (L - 1 < fft_out) */
                double acc = Yr[0];
                for (long k = 1; k < L; k++) { const long j = (k * m) % Po; acc += 2.0 * (Yr[k] * co[j] - Yi[k] * so[j]); }
                ob[(size_t)c * (size_t)Po + (size_t)m] = acc;
            }
        }
        free(Yr); free(Yi);
    }
    for (long c = 0; c < n_blocks; c++) for (long m = 0; m < Po; m++) stream[(size_t)c * No + m] += ob[(size_t)c * Po + m];      /* wave_out = output_buf[..fft_out] + overlap */
    for (size_t i = 0; i < n_out; i++) out[i] = (float)stream[(size_t)p.delay + i];
    free(h); free(ci); free(si); free(co); free(so); free(Hr); free(Hi); free(stream); free(ob);
}

/* ---- streaming encoder: Q4AudioEncoder::forward_with_cache (gguf/model.rs:437-452), Q4EncoderLayer::forward_with_cache (:299-317),
 * Q4Attention::forward_with_cache (:125-174: offset = cache length, k/v appended, causal + sliding-window masks with offset),
 * Q4VoxtralModel::encode_audio_with_cache (:791-799).  The cache is the reference's DYNAMIC (cat-based) mode, the only one
 * KVCache::apply_sliding_window (kv_cache.rs:176-203) supports: evict = keep the last `window` rows.  `abs` = positions seen so far
 * = RoPE offset of the next chunk (the reference uses cache.seq_len() for both; after an eviction its seq_len() restarts from the window
 * size, which would rotate new keys with the wrong phase -- nothing in the reference calls apply_sliding_window, so the streaming path here
 * keeps the ABSOLUTE position for RoPE: chunked == whole-utterance for any stream length, which is the property the tests pin). */
struct orc_enc_cache { int layers, heads, hd, cap, len, abs; float* k; float* v; };   /* [layer][cap][heads][hd] */
orc_enc_cache* orc_enc_cache_create(const orc_model* m, int cap) {
    orc_enc_cache* c = (orc_enc_cache*)calloc(1, sizeof *c);
    c->layers = m->cfg.enc_layers; c->heads = m->cfg.enc_heads; c->hd = m->cfg.enc_head_dim; c->cap = cap;
    size_t n = (size_t)c->layers * cap * c->heads * c->hd;
    c->k = (float*)calloc(n ? n : 1, sizeof(float)); c->v = (float*)calloc(n ? n : 1, sizeof(float)); return c;
}
void orc_enc_cache_free(orc_enc_cache* c) { if (c) { free(c->k); free(c->v); free(c); } }
int orc_enc_cache_len(const orc_enc_cache* c) { return c->len; }
int orc_enc_cache_abs(const orc_enc_cache* c) { return c->abs; }
/* kv_cache.rs:176-203 for every layer */
void orc_enc_cache_apply_sliding_window(orc_enc_cache* c, int window) {
    if (c->len <= window) return;
    const size_t row = (size_t)c->heads * c->hd; const int start = c->len - window;
    for (int l = 0; l < c->layers; l++) {
        float* kl = c->k + (size_t)l * c->cap * row; float* vl = c->v + (size_t)l * c->cap * row;
        memmove(kl, kl + (size_t)start * row, sizeof(float) * (size_t)window * row);
        memmove(vl, vl + (size_t)start * row, sizeof(float) * (size_t)window * row);
    }
    c->len = window;
}
/* returns the number of adapter rows written to out ([S4][dec_dim]); -1 if the chunk does not fit the cache */
int orc_encode_audio_with_cache(const orc_model* m, const float* mel, int T, orc_enc_cache* kc, float* out) {
    const orc_model_cfg* c = &m->cfg; int S = orc_enc_seq_len(m, T), D = c->enc_dim, H = c->enc_heads, hd = c->enc_head_dim;
    if (S <= 0) return 0;
    if (kc->len + S > kc->cap) return -1;
    const int off = kc->len; const size_t row = (size_t)H * hd, sd = (size_t)S * D, sh = (size_t)S * row;
    float* x = (float*)malloc(sizeof(float) * sd);
    orc_encoder_conv(m, mel, T, x);
    float *xn = (float*)malloc(sizeof(float) * sd), *q = (float*)malloc(sizeof(float) * sh), *at = (float*)malloc(sizeof(float) * sh), *o = (float*)malloc(sizeof(float) * sd);
    for (int l = 0; l < c->enc_layers; l++) {
        const enc_layer_t* L = &m->enc[l];
        float* kl = kc->k + (size_t)l * kc->cap * row; float* vl = kc->v + (size_t)l * kc->cap * row;
        orc_rms_norm(x, S, D, L->attn_norm, c->norm_eps, xn);
        linear_fwd(&L->wq, xn, S, q);
        linear_fwd(&L->wk, xn, S, kl + (size_t)off * row);
        linear_fwd(&L->wv, xn, S, vl + (size_t)off * row);
        orc_rope(q, S, H, hd, kc->abs, c->rope_theta); orc_rope(kl + (size_t)off * row, S, H, hd, kc->abs, c->rope_theta);
        orc_attention(q, kl, vl, S, off + S, H, H, hd, off, 1, c->enc_window, at);
        linear_fwd(&L->wo, at, S, o);
        for (size_t i = 0; i < sd; i++) x[i] = o[i] + x[i];
        orc_rms_norm(x, S, D, L->ffn_norm, c->norm_eps, xn);
        swiglu(&L->w1, &L->w2, &L->w3, xn, S, o);
        for (size_t i = 0; i < sd; i++) x[i] = o[i] + x[i];
    }
    kc->len = off + S; kc->abs += S;
    orc_encoder_final_norm(m, x, S);
    int S4 = S / c->reshape_factor;
    if (S4 > 0) {
        float* h = (float*)malloc(sizeof(float) * (size_t)S4 * m->ad0.N);
        linear_fwd(&m->ad0, x, S4, h);
        for (size_t i = 0; i < (size_t)S4 * m->ad0.N; i++) h[i] = orc_gelu(h[i]);
        linear_fwd(&m->ad2, h, S4, out); free(h);
    }
    free(x); free(xn); free(q); free(at); free(o);
    return S4;
}

/* gguf/model.rs:584-618 (row dequant) == select on the dequantised table (:568-576) */
void orc_embed_tokens(const orc_model* m, const int32_t* ids, int n, float* out) {
    int64_t D = m->tok.K;
    for (int i = 0; i < n; i++) {
        if (m->tok.q4) orc_q4_dequantize(m->tok.q4 + (size_t)ids[i] * (D / 32) * 18, D, out + (size_t)i * D);
        else memcpy(out + (size_t)i * D, m->tok.dense + (size_t)ids[i] * D, D * sizeof(float));
    }
}

/* models/layers/kv_cache.rs:52-65,221-234 */
orc_cache* orc_cache_create(const orc_model* m, int max_seq) {
    orc_cache* c = (orc_cache*)calloc(1, sizeof *c);
    c->layers = m->cfg.dec_layers; c->n_kv = m->cfg.dec_kv_heads; c->hd = m->cfg.dec_head_dim; c->max_seq = max_seq;
    size_t n = (size_t)c->layers * max_seq * c->n_kv * c->hd;
    c->k = (float*)calloc(n ? n : 1, sizeof(float)); c->v = (float*)calloc(n ? n : 1, sizeof(float)); return c;
}
void orc_cache_free(orc_cache* c) { if (c) { free(c->k); free(c->v); free(c); } }
int  orc_cache_len(const orc_cache* c) { return c->len; }
void orc_cache_reset(orc_cache* c) { c->len = 0; }
/* KVCache::update on one layer (models/layers/kv_cache.rs:116-136): k / v [n_kv][n][hd] -> rows pos .. pos + n of the layer; len = max(len, pos + n) */
void orc_cache_update(orc_cache* c, int layer, int pos, const float* k, const float* v, int n) {
    for (int h = 0; h < c->n_kv; h++) for (int r = 0; r < n; r++) {
        size_t dst = (((size_t)layer * c->max_seq + pos + r) * c->n_kv + h) * c->hd, src = ((size_t)h * n + r) * c->hd;
        memcpy(c->k + dst, k + src, sizeof(float) * c->hd); memcpy(c->v + dst, v + src, sizeof(float) * c->hd);
    }
    if (pos + n > c->len) c->len = pos + n;
}

/* gguf/model.rs:665-677 -> :370-387 -> :125-174, :250-255, :220-224 */
void orc_forward_hidden_with_cache(const orc_model* m, const float* xin, int M, const float* t_embed, orc_cache* kc, float* out) {
    const orc_model_cfg* c = &m->cfg; int D = c->dec_dim, H = c->dec_heads, KV = c->dec_kv_heads, hd = c->dec_head_dim;
    size_t md = (size_t)M * D; int off = kc->len;
    float *x = (float*)malloc(sizeof(float) * md), *xn = (float*)malloc(sizeof(float) * md), *o = (float*)malloc(sizeof(float) * md);
    float *q = (float*)malloc(sizeof(float) * (size_t)M * H * hd), *at = (float*)malloc(sizeof(float) * (size_t)M * H * hd);
    float *tc = (float*)malloc(sizeof(float) * c->t_cond_dim), *sc = (float*)malloc(sizeof(float) * D);
    memcpy(x, xin, sizeof(float) * md);
    for (int l = 0; l < c->dec_layers; l++) {
        const dec_layer_t* L = &m->dec[l];
        float* kl = kc->k + (size_t)l * kc->max_seq * KV * hd; float* vl = kc->v + (size_t)l * kc->max_seq * KV * hd;
        orc_rms_norm(x, M, D, L->attn_norm, c->norm_eps, xn);
        linear_fwd(&L->wq, xn, M, q);
        linear_fwd(&L->wk, xn, M, kl + (size_t)off * KV * hd);      /* kv_cache.rs:116-136: write at [len, len+M) */
        linear_fwd(&L->wv, xn, M, vl + (size_t)off * KV * hd);
        orc_rope(q, M, H, hd, off, c->rope_theta); orc_rope(kl + (size_t)off * KV * hd, M, KV, hd, off, c->rope_theta);
        orc_attention(q, kl, vl, M, off + M, H, KV, hd, off, 1, c->dec_window, at);
        linear_fwd(&L->wo, at, M, o);
        for (size_t i = 0; i < md; i++) x[i] = o[i] + x[i];
        orc_rms_norm(x, M, D, L->ffn_norm, c->norm_eps, xn);
        /* Q4AdaRmsNorm: x * (1 + w2(gelu(w0(t_embed)))) */
        linear_fwd(&L->ada0, t_embed, 1, tc);
        for (int i = 0; i < c->t_cond_dim; i++) tc[i] = orc_gelu(tc[i]);
        linear_fwd(&L->ada2, tc, 1, sc);
        for (int r = 0; r < M; r++) for (int i = 0; i < D; i++) xn[(size_t)r * D + i] = xn[(size_t)r * D + i] * (sc[i] + 1.0f);
        swiglu(&L->w1, &L->w2, &L->w3, xn, M, o);
        for (size_t i = 0; i < md; i++) x[i] = o[i] + x[i];
    }
    kc->len = off + M;
    orc_rms_norm(x, M, D, m->dec_norm, c->norm_eps, out);
    free(x); free(xn); free(o); free(q); free(at); free(tc); free(sc);
}

/* gguf/model.rs:680-691 */
void orc_lm_head(const orc_model* m, const float* hidden, int M, float* logits) { linear_fwd(&m->tok, hidden, M, logits); }

static __thread double g_enc_ms, g_dec_ms;
void orc_last_timings(double* e, double* d) { *e = g_enc_ms; *d = g_dec_ms; }

/* argmax with lowest-index tie-break (SURVEY.md 8c parity definition) */
static int argmax_f32(const float* v, int n) { int b = 0; for (int i = 1; i < n; i++) if (v[i] > v[b]) b = i; return b; }

/* gguf/model.rs:873-963 */
int orc_transcribe_streaming(const orc_model* m, const float* mel, int T, const float* t_embed, int32_t* out_ids, int cap, float* logits_out) {
    const orc_model_cfg* c = &m->cfg; const int PREFIX_LEN = 38, BOS = 1, PAD = 32; int D = c->dec_dim, V = c->vocab;
    double t0 = now_ms();
    int Senc = orc_enc_seq_len(m, T), S4max = Senc / c->reshape_factor;
    float* audio = (float*)malloc(sizeof(float) * (size_t)(S4max > 0 ? S4max : 1) * D);
    int S = orc_encode_audio(m, mel, T, audio);
    g_enc_ms = now_ms() - t0; t0 = now_ms();
    if (S < PREFIX_LEN) { free(audio); g_dec_ms = 0; return 0; }
    int32_t* gen = (int32_t*)malloc(sizeof(int32_t) * (size_t)(S + 1));
    gen[0] = BOS; for (int i = 1; i < PREFIX_LEN; i++) gen[i] = PAD;
    float* x = (float*)malloc(sizeof(float) * (size_t)PREFIX_LEN * D); float* h = (float*)malloc(sizeof(float) * (size_t)PREFIX_LEN * D);
    float* lg = (float*)malloc(sizeof(float) * (size_t)V);
    orc_embed_tokens(m, gen, PREFIX_LEN, x);
    for (size_t i = 0; i < (size_t)PREFIX_LEN * D; i++) x[i] = audio[i] + x[i];
    orc_cache* kc = orc_cache_create(m, S);
    orc_forward_hidden_with_cache(m, x, PREFIX_LEN, t_embed, kc, h);
    /* the reference computes all 38 logit rows and keeps the last (:916-923) */
    orc_lm_head(m, h + (size_t)(PREFIX_LEN - 1) * D, 1, lg);
    int n = 0; gen[PREFIX_LEN] = argmax_f32(lg, V);
    if (logits_out) memcpy(logits_out + (size_t)n * V, lg, sizeof(float) * V);
    n++;
    for (int pos = PREFIX_LEN + 1; pos < S; pos++) {
        orc_embed_tokens(m, &gen[pos - 1], 1, x);
        for (int i = 0; i < D; i++) x[i] = audio[(size_t)(pos - 1) * D + i] + x[i];
        orc_forward_hidden_with_cache(m, x, 1, t_embed, kc, h);
        orc_lm_head(m, h, 1, lg);
        gen[pos] = argmax_f32(lg, V);
        if (logits_out) memcpy(logits_out + (size_t)n * V, lg, sizeof(float) * V);
        n++;
    }
    int n_ids = n;   /* = max(S - 38, 1): at S == 38 the reference still returns the first predicted token (model.rs:922-926, loop :938 empty) */
    for (int i = 0; i < n_ids && i < cap; i++) out_ids[i] = gen[PREFIX_LEN + i];
    g_dec_ms = now_ms() - t0;
    orc_cache_free(kc); free(audio); free(gen); free(x); free(h); free(lg);
    return n_ids;
}
