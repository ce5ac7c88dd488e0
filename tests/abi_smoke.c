/* abi_smoke.c -- a C11 translation unit that includes include/voxtral_hip.h and calls through it (tests/test_abi_cpu.py compiles it with
 * gcc -std=c11 -Wall -Wextra -Werror -Iinclude and links it against libvoxtral_hip.so: ctypes checks symbol NAMES, this checks that the header is
 * valid C and that the declared signatures link and behave).  Host-only entry points are exercised for real; GPU entry points are called when a
 * device is present (argv[1] = path of a tiny GGUF written by the test) and must fail cleanly with VOX_ERR_HIP when it is not.
 * Exit code 0 = every check passed; prints one line per section. */
#include "voxtral_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "abi_smoke FAILED: %s (line %d): %s\n", msg, __LINE__, vox_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    /* ---- host-only: version, padding arithmetic (audio/pad.rs), chunk planning (audio/chunk.rs), peak normalise, time embedding */
    CHECK(vox_abi_version() >= 1, "vox_abi_version");
    vox_pad_cfg pc; vox_pad_cfg_voxtral(&pc);
    const size_t n = 16000;                                     /* 1 s */
    size_t total = 0;
    CHECK(vox_pad_len(n, &pc, &total) == VOX_OK && total > n && total % 160 == 0, "vox_pad_len aligns to the hop");
    float* x = (float*)malloc(n * sizeof(float)); float* xp = (float*)calloc(total, sizeof(float));
    for (size_t i = 0; i < n; i++) x[i] = 0.25f * sinf((float)i * 0.05f);
    CHECK(vox_peak_normalize(x, n, 0.95f) == VOX_OK, "vox_peak_normalize");
    float mx = 0.f; for (size_t i = 0; i < n; i++) mx = fmaxf(mx, fabsf(x[i]));
    CHECK(fabsf(mx - 0.95f) < 1e-5f, "peak is 0.95 after normalisation");
    CHECK(vox_pad_audio(x, n, &pc, xp) == VOX_OK, "vox_pad_audio");
    vox_chunk_cfg cc = {1200u, 160u, 16000u, 0u}; int32_t need = -1; vox_chunk ch[8]; size_t nch = 0;
    CHECK(vox_needs_chunking(30u * 16000u, &cc, &need) == VOX_OK && need == 1, "vox_needs_chunking: 30 s needs chunks at 1200 frames");
    CHECK(vox_chunk_plan(30u * 16000u, &cc, ch, 8, &nch) == VOX_OK && nch == 3 && ch[2].is_last == 1, "vox_chunk_plan: 3 chunks (chunk.rs:185-265)");
    float te[32]; CHECK(vox_time_embedding(6.0f, 32, te) == VOX_OK, "vox_time_embedding");
    CHECK(fabsf(te[0] - cosf(6.0f)) < 1e-5f, "time embedding layout: cos first (time_embedding.rs:41-71)");
    printf("host helpers ok: pad %zu -> %zu samples\n", n, total);

    /* ---- host-only: the GGUF reader (gguf/reader.rs:98-223) on an image built here -- a v3 file with one KV and one 2 x 4 F32 tensor -- then on EVERY truncation
     * of it (each must fail cleanly: this is the reader's pointer arithmetic under -fsanitize=address in tests/test_abi_cpu.py) and as two shards */
    {
        unsigned char img[512]; size_t o = 0;
#define PUT(ptr, len) do { memcpy(img + o, (ptr), (len)); o += (len); } while (0)
#define PUT_U32(v) do { const uint32_t v_ = (v); PUT(&v_, 4); } while (0)
#define PUT_U64(v) do { const uint64_t v_ = (v); PUT(&v_, 8); } while (0)
#define PUT_STR(str) do { PUT_U64(strlen(str)); PUT((str), strlen(str)); } while (0)
        PUT_U32(0x46554747u); PUT_U32(3u); PUT_U64(1); PUT_U64(1);
        PUT_STR("general.architecture"); PUT_U32(8u); PUT_STR("voxtral");
        PUT_STR("t.weight"); PUT_U32(2u); PUT_U64(4); PUT_U64(2); PUT_U32(0u); PUT_U64(0);
        while (o % 32) img[o++] = 0;
        float vals[8]; for (int i = 0; i < 8; i++) vals[i] = 1.5f * (float)i - 2.0f;
        PUT(vals, sizeof vals);
        const size_t size = o;
        vox_gguf* g = NULL;
        CHECK(vox_gguf_open_memory(img, size, &g) == VOX_OK, "vox_gguf_open_memory");
        uint32_t ver = 0; uint64_t cnt = 0; const char* nm = NULL;
        CHECK(vox_gguf_version(g, &ver) == VOX_OK && ver == 3, "version 3");
        CHECK(vox_gguf_tensor_count(g, &cnt) == VOX_OK && cnt == 1, "one tensor");
        CHECK(vox_gguf_tensor_name(g, 0, &nm) == VOX_OK && strcmp(nm, "t.weight") == 0, "tensor name");
        uint64_t dims[4] = {0, 0, 0, 0}, nbytes = 0; uint32_t nd = 0, dt = 99;
        CHECK(vox_gguf_tensor_info(g, "t.weight", dims, &nd, &dt, &nbytes) == VOX_OK && nd == 2 && dims[0] == 4 && dims[1] == 2 && dt == 0 && nbytes == 32, "tensor info");
        float back[8]; CHECK(vox_gguf_tensor_data(g, "t.weight", back, sizeof back) == VOX_OK && memcmp(back, vals, sizeof vals) == 0, "tensor data");
        CHECK(vox_gguf_tensor_data(g, "t.weight", back, 16) != VOX_OK, "short destination is refused");
        CHECK(vox_gguf_tensor_info(g, "missing", dims, &nd, &dt, &nbytes) != VOX_OK, "unknown tensor is refused");
        CHECK(vox_gguf_close(g) == VOX_OK, "close");
        int refused = 0;
        for (size_t cut = 0; cut < size; cut++) {                 /* heap copy of exactly `cut` bytes: a read past the end is an ASAN report */
            unsigned char* part = (unsigned char*)malloc(cut ? cut : 1); memcpy(part, img, cut);
            vox_gguf* gt = NULL;
            if (vox_gguf_open_memory(part, cut, &gt) != VOX_OK) refused++;
            else { float tmp[8]; if (vox_gguf_tensor_data(gt, "t.weight", tmp, sizeof tmp) != VOX_OK) refused++; (void)vox_gguf_close(gt); }
            free(part);
        }
        CHECK(refused == (int)size, "every truncated image is refused (at open or at the tensor read)");
        const void* shards[2] = {img, img + 50}; const size_t sizes[2] = {50, size - 50};
        CHECK(vox_gguf_open_shards(shards, sizes, 2, &g) == VOX_OK, "vox_gguf_open_shards");
        CHECK(vox_gguf_tensor_data(g, "t.weight", back, sizeof back) == VOX_OK && memcmp(back, vals, sizeof vals) == 0, "sharded image reads the same");
        CHECK(vox_gguf_close(g) == VOX_OK, "close shards");
        printf("gguf reader ok: %zu-byte image, %d truncations refused, shards ok\n", size, refused);
    }
    if (getenv("VOX_SMOKE_HOST_ONLY")) { free(x); free(xp); return 0; }

    /* ---- device: context, GGUF reader, model load, one transcription through the whole hot path */
    int32_t ndev = 0; CHECK(vox_device_count(&ndev) == VOX_OK, "vox_device_count");
    vox_ctx* ctx = NULL;
    const int32_t rc = vox_ctx_create(0, &ctx);
    if (ndev == 0 || rc != VOX_OK) {
        CHECK(rc == VOX_ERR_HIP && ctx == NULL, "without a GPU vox_ctx_create fails with VOX_ERR_HIP (no CPU fallback)");
        printf("no GPU: product entry points fail loudly, as they must\n");
        free(x); free(xp); return 0;
    }
    if (argc > 1) {
        vox_gguf* g = NULL; CHECK(vox_gguf_open(argv[1], &g) == VOX_OK, "vox_gguf_open");
        uint32_t version = 0; uint64_t n_tensors = 0;
        CHECK(vox_gguf_version(g, &version) == VOX_OK && (version == 2 || version == 3), "vox_gguf_version");
        CHECK(vox_gguf_tensor_count(g, &n_tensors) == VOX_OK && n_tensors > 0, "vox_gguf_tensor_count");
        vox_model* m = NULL; CHECK(vox_q4_model_load_gguf(ctx, g, 0, &m) == VOX_OK, "vox_q4_model_load_gguf");
        CHECK(vox_gguf_close(g) == VOX_OK, "vox_gguf_close");
        vox_model_cfg cfg; CHECK(vox_model_config(m, &cfg) == VOX_OK && cfg.dec_dim > 0, "vox_model_config");
        float* t_embed = (float*)malloc((size_t)cfg.dec_dim * sizeof(float));
        CHECK(vox_time_embedding(6.0f, cfg.dec_dim, t_embed) == VOX_OK, "vox_time_embedding(dec_dim)");
        int32_t active = -1; CHECK(vox_model_set_decode_engine(m, 1, &active) == VOX_OK && (active == 0 || active == 1), "vox_model_set_decode_engine");
        int32_t ids[512]; int32_t n_ids = 0;
        CHECK(vox_transcribe_audio(m, x, n, t_embed, ids, 512, &n_ids, VOX_MEM_HOST) == VOX_OK, "vox_transcribe_audio");
        vox_timings tm; CHECK(vox_get_stage_timings(m, &tm) == VOX_OK && tm.decode_tokens == n_ids, "vox_get_stage_timings");
        /* composite entry point: generate_step_with_cache == embed + forward_hidden_with_cache + lm_head */
        vox_cache* kc = NULL; CHECK(vox_decoder_cache_create(m, 64, &kc) == VOX_OK, "vox_decoder_cache_create");
        const int32_t toks[3] = {1, 32, 32};
        float* lg = (float*)malloc((size_t)3 * cfg.vocab * sizeof(float));
        CHECK(vox_generate_step_with_cache(m, toks, 3, t_embed, kc, lg) == VOX_OK, "vox_generate_step_with_cache");
        int32_t len = 0; CHECK(vox_cache_seq_len(kc, &len) == VOX_OK && len == 3, "cache advanced by 3");
        CHECK(vox_cache_free(kc) == VOX_OK, "vox_cache_free");
        printf("device path ok: %d ids from a 1 s clip (engine %s), generate_step logits[0][0] = %g\n", (int)n_ids, active ? "on" : "off", (double)lg[0]);
        free(lg); free(t_embed);
        CHECK(vox_model_free(m) == VOX_OK, "vox_model_free");
    }
    CHECK(vox_ctx_destroy(ctx) == VOX_OK, "vox_ctx_destroy");
    free(x); free(xp);
    return 0;
}
