/* e2e_piecewise.c -- the reference's e2e-bench loop (bin/e2e_bench.rs:96-254) replayed call for call through include/voxtral_hip.h from plain C11.
 *
 * The reference's headline metric is DEFINED by this loop (bin/e2e_bench.rs:179-224; web/bindings.rs:357-424 drives decode the same way): preprocess on the host,
 * encode_audio, then per position
 *     embed_tokens_from_ids(&[token]) -> audio_pos + text_embed -> forward_hidden_with_cache -> lm_head -> argmax(2) -> into_scalar
 * on device-resident tensors.  A maintainer who binds the C ABI under the reference's own Decoder type gets exactly this call sequence, so this program measures what a
 * drop-in `e2e-bench` would report -- not the fused vox_transcribe_audio path bench.py's `value` is quoted on.
 *
 *   e2e_piecewise <model.gguf> <samples.f32> [reps]
 * prints one JSON object: stage timings and tok/s (ids / decode-stage seconds, e2e_bench.rs:236-240) of the call-for-call loop ("lm_head+argmax") and of the same loop with
 * vox_lm_head_argmax in place of the last two calls ("lm_head_argmax"), and whether both id sequences equal vox_transcribe_audio's.
 * Built by bench.py / tests with: gcc -std=c11 -O2 -Wall -Wextra -Werror -Iinclude tools/e2e_piecewise.c -L<pkg> -lvoxtral_hip -lm */
#define _POSIX_C_SOURCE 200809L
#include "voxtral_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define OK(call) do { if ((call) != VOX_OK) { fprintf(stderr, "e2e_piecewise: %s failed (line %d): %s\n", #call, __LINE__, vox_last_error()); return 1; } } while (0)
#define PREFIX_LEN 38
#define BOS_TOKEN 1
#define STREAMING_PAD 32

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6; }

typedef struct { double encode_ms, decode_ms; int n_ids; } run_stats;

/* one pass of bin/e2e_bench.rs:158-231 on device tensors.  fused = 0: lm_head + argmax(2) (the reference's calls); 1: vox_lm_head_argmax */
static int decode_pass(vox_ctx* ctx, vox_model* m, const vox_model_cfg* cfg, const float* d_mel, int T, const float* t_embed, float* d_audio, int cap_rows,
                       float* d_text, float* d_input, float* d_logits, int fused, int32_t* generated, run_stats* st) {
    const size_t D = (size_t)cfg->dec_dim, V = (size_t)cfg->vocab;
    double t0 = now_ms();
    int32_t S = 0;
    OK(vox_encode_audio(m, d_mel, T, d_audio, cap_rows, &S, VOX_MEM_DEVICE));                 /* model.encode_audio(mel_tensor), :160 */
    OK(vox_ctx_synchronize(ctx));                                                              /* "Force GPU sync", :164-165 */
    st->encode_ms = now_ms() - t0; t0 = now_ms();
    st->n_ids = 0; st->decode_ms = 0;
    if (S < PREFIX_LEN) return 0;
    int32_t prefix[PREFIX_LEN]; prefix[0] = BOS_TOKEN; for (int i = 1; i < PREFIX_LEN; i++) prefix[i] = STREAMING_PAD;      /* :174-175 */
    OK(vox_embed_tokens_from_ids_ex(m, prefix, PREFIX_LEN, d_text, VOX_MEM_DEVICE));            /* :177-179 */
    OK(vox_tensor_add(ctx, d_audio, d_text, (size_t)PREFIX_LEN * D, d_input, VOX_MEM_DEVICE)); /* prefix_audio + prefix_text_embeds, :181-184 */
    vox_cache* cache = NULL;
    OK(vox_decoder_cache_create(m, S, &cache));                                                 /* create_decoder_cache_preallocated(seq_len), :186 */
    const float* hidden = NULL;
    OK(vox_forward_hidden_with_cache_ex(m, d_input, PREFIX_LEN, t_embed, cache, NULL, &hidden, VOX_MEM_DEVICE));      /* :188-192 */
    int32_t tok = 0;
    /* the reference multiplies all 38 rows and keeps the last (:193-199); only that row is read back here too, but all 38 are computed */
    if (!fused) {
        OK(vox_lm_head_ex(m, hidden, PREFIX_LEN, d_logits, VOX_MEM_DEVICE));
        OK(vox_argmax_rows(ctx, d_logits + (size_t)(PREFIX_LEN - 1) * V, 1, (int32_t)V, &tok, VOX_MEM_DEVICE));
    } else {
        int32_t all[PREFIX_LEN];
        OK(vox_lm_head_argmax(m, hidden, PREFIX_LEN, all, VOX_MEM_DEVICE)); tok = all[PREFIX_LEN - 1];
    }
    int n = 0; generated[n++] = tok;                                                            /* generated.push(first_token), :202-203 */
    for (int pos = PREFIX_LEN + 1; pos < S; pos++) {                                            /* :211-224 */
        const int32_t new_token = generated[n - 1];
        OK(vox_embed_tokens_from_ids_ex(m, &new_token, 1, d_text, VOX_MEM_DEVICE));
        OK(vox_tensor_add(ctx, d_audio + (size_t)(pos - 1) * D, d_text, D, d_input, VOX_MEM_DEVICE));
        OK(vox_forward_hidden_with_cache_ex(m, d_input, 1, t_embed, cache, NULL, &hidden, VOX_MEM_DEVICE));
        if (!fused) {
            OK(vox_lm_head_ex(m, hidden, 1, d_logits, VOX_MEM_DEVICE));
            OK(vox_argmax_rows(ctx, d_logits, 1, (int32_t)V, &tok, VOX_MEM_DEVICE));             /* pred.into_scalar(): the per-token synchronisation */
        } else {
            OK(vox_lm_head_argmax(m, hidden, 1, &tok, VOX_MEM_DEVICE));
        }
        generated[n++] = tok;
    }
    st->decode_ms = now_ms() - t0; st->n_ids = n;
    OK(vox_cache_free(cache));
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: e2e_piecewise <model.gguf> <samples.f32> [reps]\n"); return 2; }
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    FILE* f = fopen(argv[2], "rb"); if (!f) { perror(argv[2]); return 2; }
    fseek(f, 0, SEEK_END); const long fbytes = ftell(f); fseek(f, 0, SEEK_SET);
    const size_t n = (size_t)fbytes / 4; float* x = (float*)malloc(n * 4);
    if (fread(x, 4, n, f) != n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    vox_ctx* ctx = NULL; vox_model* m = NULL; vox_model_cfg cfg;
    OK(vox_ctx_create(0, &ctx));
    double t0 = now_ms();
    OK(vox_q4_model_load(ctx, argv[1], &m));
    const double load_ms = now_ms() - t0;
    OK(vox_model_config(m, &cfg));
    const size_t D = (size_t)cfg.dec_dim, V = (size_t)cfg.vocab;
    float* t_embed = (float*)malloc(D * 4); OK(vox_time_embedding(6.0f, cfg.dec_dim, t_embed));      /* TimeEmbedding::new(3072).embed(delay), :155-156 */

    /* ---- preprocess_audio (:97-135): peak normalise, pad, log-mel, transpose to [n_mels][T], "tensor from data" = upload */
    t0 = now_ms();
    float* xn = (float*)malloc(n * 4); memcpy(xn, x, n * 4);
    OK(vox_peak_normalize(xn, n, 0.95f));
    vox_pad_cfg pc; OK(vox_pad_cfg_voxtral(&pc));
    size_t total = 0; OK(vox_pad_len(n, &pc, &total));
    float* xp = (float*)malloc(total * 4); OK(vox_pad_audio(xn, n, &pc, xp));
    size_t T = 0; OK(vox_mel_num_frames(total, &T));
    float* mel = (float*)malloc(T * 128 * 4); OK(vox_mel_compute_log(ctx, xp, total, mel, VOX_MEM_HOST));
    float* mel_t = (float*)malloc(T * 128 * 4);
    for (size_t t = 0; t < T; t++) for (int k = 0; k < 128; k++) mel_t[(size_t)k * T + t] = mel[t * 128 + k];
    void* d_mel = NULL; OK(vox_dev_alloc(ctx, T * 128 * 4, &d_mel)); OK(vox_dev_upload(ctx, d_mel, mel_t, T * 128 * 4));
    const double preprocess_ms = now_ms() - t0;

    const int cap_rows = (int)(T / 16 + 2);
    void *d_audio = NULL, *d_text = NULL, *d_input = NULL, *d_logits = NULL;
    OK(vox_dev_alloc(ctx, (size_t)cap_rows * D * 4, &d_audio)); OK(vox_dev_alloc(ctx, (size_t)PREFIX_LEN * D * 4, &d_text));
    OK(vox_dev_alloc(ctx, (size_t)PREFIX_LEN * D * 4, &d_input)); OK(vox_dev_alloc(ctx, (size_t)PREFIX_LEN * V * 4, &d_logits));
    int32_t* gen[2] = {(int32_t*)calloc((size_t)cap_rows, 4), (int32_t*)calloc((size_t)cap_rows, 4)};
    run_stats best[2] = {{0, 0, 0}, {0, 0, 0}};
    for (int fused = 0; fused < 2; fused++) {
        double enc = 0, dec = 0; int cnt = 0;
        for (int r = 0; r < reps + 1; r++) {      /* pass 0 warms workspaces / the engine's weight stream (the reference amortises model load the same way) */
            run_stats st;
            if (decode_pass(ctx, m, &cfg, (const float*)d_mel, (int)T, t_embed, (float*)d_audio, cap_rows, (float*)d_text, (float*)d_input, (float*)d_logits, fused, gen[fused], &st)) return 1;
            if (r > 0) { enc += st.encode_ms; dec += st.decode_ms; cnt++; }
            best[fused].n_ids = st.n_ids;
        }
        best[fused].encode_ms = enc / cnt; best[fused].decode_ms = dec / cnt;
    }
    /* the fused product path on the same clip: the ids every variant must reproduce */
    int32_t* ref = (int32_t*)calloc((size_t)cap_rows + 128, 4); int32_t n_ref = 0;
    OK(vox_transcribe_audio(m, x, n, t_embed, ref, cap_rows + 128, &n_ref, VOX_MEM_HOST));
    int same[2];
    for (int v = 0; v < 2; v++) same[v] = n_ref == best[v].n_ids && memcmp(ref, gen[v], (size_t)n_ref * 4) == 0;
    int32_t eng = 0; OK(vox_model_set_decode_engine(m, 1, &eng));
    const double audio_s = (double)n / 16000.0;
    printf("{\"audio_s\": %.3f, \"mel_frames\": %zu, \"load_ms\": %.1f, \"preprocess_ms\": %.3f, \"decode_engine\": %s, \"reps\": %d", audio_s, T, load_ms, preprocess_ms, eng ? "true" : "false", reps);
    const char* names[2] = {"lm_head+argmax", "lm_head_argmax"};
    for (int v = 0; v < 2; v++) {
        const double total_ms = preprocess_ms + best[v].encode_ms + best[v].decode_ms;
        printf(", \"%s\": {\"encode_ms\": %.3f, \"decode_ms\": %.3f, \"decode_tokens\": %d, \"tok_per_s\": %.1f, \"total_ms\": %.3f, \"rtf\": %.5f, \"ids_equal_transcribe_audio\": %s}",
               names[v], best[v].encode_ms, best[v].decode_ms, best[v].n_ids, best[v].decode_ms > 0 ? best[v].n_ids / (best[v].decode_ms / 1e3) : 0.0, total_ms,
               total_ms / 1e3 / audio_s, same[v] ? "true" : "false");
    }
    printf(", \"ids\": [");
    for (int i = 0; i < best[0].n_ids; i++) printf(i ? ", %d" : "%d", gen[0][i]);
    printf("]}\n");
    OK(vox_dev_free(ctx, d_mel)); OK(vox_dev_free(ctx, d_audio)); OK(vox_dev_free(ctx, d_text)); OK(vox_dev_free(ctx, d_input)); OK(vox_dev_free(ctx, d_logits));
    OK(vox_model_free(m)); OK(vox_ctx_destroy(ctx));
    free(x); free(xn); free(xp); free(mel); free(mel_t); free(t_embed); free(gen[0]); free(gen[1]); free(ref);
    return same[0] && same[1] ? 0 : 3;
}
