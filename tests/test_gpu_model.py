"""GPU parity of the model-forward surface (Q4ModelLoader / Q4VoxtralModel / Q4LanguageModel) against
the CPU oracle and against the golden vectors produced by the reference's PyTorch restatement.

Tolerances (stated here, used below):
  * hidden states / audio embeddings / logits: max|d| <= 2e-4 * max|ref|  (the Q4 epsilon of SURVEY 8c is
    1e-2 * max(1,|logit|); the HIP path is ~2 orders tighter because activations are split hi+lo bf16
    before the MFMA and everything else is f32)
  * greedy token ids: identical to the oracle wherever the oracle's top-2 logit margin exceeds 10x the
    logit tolerance (ties are decided by lowest index on both sides)."""
import numpy as np
import pytest

from model_fixtures import check_batch_rows, check_greedy_ids, fake_mel, golden, golden_gguf, rel_err, tiny_gguf

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def tiny(pkg, orc, ctx):
    path, dims = tiny_gguf()
    m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    o = orc.Model(path)
    yield m, o, dims
    m.close(); o.close()


def test_loader_config_and_errors(pkg, ctx, tiny, tmp_path):
    m, o, dims = tiny
    c = m.config
    assert (c.enc_layers, c.enc_dim, c.enc_heads, c.enc_head_dim, c.enc_ffn, c.enc_window) == (2, 128, 2, 64, 256, 750)
    assert (c.dec_layers, c.dec_dim, c.dec_heads, c.dec_kv_heads, c.dec_head_dim, c.dec_ffn, c.dec_window, c.vocab) == (2, 256, 4, 2, 128, 512, 8192, 512)
    assert c.t_cond_dim == 32 and c.reshape_factor == 4 and m.decoder().n_layers() == 2 and m.decoder().d_model() == 256
    assert m.weight_bytes() > 0
    with pytest.raises(pkg.VoxError):
        pkg.Q4ModelLoader.from_file(str(tmp_path / "missing.gguf")).load(ctx)
    S = pkg.synth
    p = str(tmp_path / "partial.gguf")
    S.write_gguf(p, [("layers.0.attention.wq.weight", (512, 256), S.GGML_Q4_0, S.synth_q4_blocks(np.random.default_rng(0), 512 * 256, 0.02))])
    with pytest.raises(pkg.VoxError, match="not found|no encoder"):
        pkg.Q4ModelLoader.from_file(p).load(ctx)


def test_embed_tokens(tiny):
    m, o, _ = tiny
    ids = np.array([1, 32, 0, 511, 77], dtype=np.int32)
    assert (m.decoder().embed_tokens_from_ids(ids, 1, 5)[0] == o.embed_tokens(ids)).all()     # exact: pure dequant


@pytest.mark.parametrize("T", [64, 250, 1144, 3400])      # 3400 frames -> 850 encoder positions > the 750 sliding window (masking.rs:26-44)
def test_encode_audio(tiny, T):
    m, o, _ = tiny
    mel = fake_mel(T, seed=T)
    ref = o.encode_audio(mel); out = m.encode_audio(mel[None])
    assert out.shape == (1,) + ref.shape and ref.shape[0] == o.enc_seq_len(T) // 4
    assert rel_err(out[0], ref) < TOL, rel_err(out[0], ref)


def test_encode_audio_conv_valu_cross_check_path(tiny, monkeypatch):
    """VOX_CONV_VALU=1 (the conv stem as plain VALU kernels instead of the im2col MFMA GEMM -- a cross-check switch that ships in the library) against the
    default path and the oracle."""
    m, o, _ = tiny
    mel = fake_mel(250, seed=3)
    ref = o.encode_audio(mel); a = m.encode_audio(mel[None])
    monkeypatch.setenv("VOX_CONV_VALU", "1")
    b = m.encode_audio(mel[None])
    monkeypatch.delenv("VOX_CONV_VALU")
    assert rel_err(b[0], ref) < TOL and rel_err(a[0], b[0]) < TOL, (rel_err(b[0], ref), rel_err(a[0], b[0]))


def test_encode_audio_too_short(tiny):
    m, o, _ = tiny
    out = m.encode_audio(fake_mel(9))            # S_enc = 3 -> 0 tokens
    assert out.shape == (1, 0, 256)


def test_forward_hidden_with_cache_and_lm_head(pkg, tiny):
    m, o, _ = tiny
    rng = np.random.default_rng(3)
    x = (0.5 * rng.standard_normal((14, 256))).astype(np.float32)
    t = pkg.TimeEmbedding(256).embed(6.0)
    oc = o.cache(32); ref = np.concatenate([o.forward_hidden_with_cache(x[:9], t, oc)] + [o.forward_hidden_with_cache(x[i:i + 1], t, oc) for i in range(9, 14)])
    dec = m.decoder(); c = dec.create_cache_preallocated(32)
    h1 = dec.forward_hidden_with_cache(x[None, :9], t, c)
    assert c.seq_len() == 9                                       # kv_cache.rs:311-336 semantics
    hs = [dec.forward_hidden_with_cache(x[None, i:i + 1], t, c) for i in range(9, 14)]
    assert c.seq_len() == 14
    out = np.concatenate([h1[0]] + [h[0] for h in hs])
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    lg = dec.lm_head(out[None]); lref = o.lm_head(ref)
    assert lg.shape == (1, 14, 512) and rel_err(lg[0], lref) < TOL
    # cached == full causal pass (attention.rs:429-474), on the GPU path itself
    c.reset(); full = dec.forward_hidden_with_cache(x[None], t, c)
    assert rel_err(full[0], out) < 1e-4
    with pytest.raises(pkg.VoxError, match="overflow"):
        dec.forward_hidden_with_cache(np.zeros((1, 30, 256), np.float32), t, c)
    o.cache_free(oc)


def _check_ids(ids, lg, rids, rlg):
    assert ids.shape == rids.shape and lg.shape == rlg.shape
    e = np.abs(lg - rlg).max(); scale = max(1.0, np.abs(rlg).max())
    assert e <= TOL * scale, e
    check_greedy_ids(ids, rids, rlg, TOL)                                     # identical up to the first near-tie
    assert (lg.argmax(1) == ids).all()                                        # ids are the argmax of the returned logits


@pytest.mark.parametrize("T", [700, 1144])
def test_transcribe_streaming(pkg, tiny, T):
    m, o, _ = tiny
    mel = fake_mel(T, seed=10 + T); t = pkg.TimeEmbedding(256).embed(6.0)
    rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
    ids, lg = m.transcribe_streaming(mel[None], t, return_logits=True)      # eager path (logits tap)
    S = o.enc_seq_len(T) // 4
    assert len(rids) == S - 38 == len(ids)                                   # gguf/model.rs:962
    _check_ids(ids, lg, rids, rlg)
    ids_g = m.transcribe_streaming(mel[None], t)                             # hipGraph-replayed decode steps
    ids_g2 = m.transcribe_streaming(mel[None], t)
    assert (ids_g == ids).all() and (ids_g2 == ids).all()                    # graph == eager, and deterministic
    tm = m.timings()
    assert tm["decode_tokens"] == S - 38 and tm["graph_replays"] >= S - 41


def test_transcribe_short_returns_empty(pkg, tiny):
    m, o, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    assert len(m.transcribe_streaming(fake_mel(300)[None], t)) == 0 == len(o.transcribe_streaming(fake_mel(300), t))   # S=18 < 38


def test_transcribe_audio_full_path(pkg, orc, tiny, ctx):
    """16 kHz samples -> peak-normalise -> pad -> log-mel -> encoder -> decoder, all on the GPU, vs the oracle pipeline."""
    m, o, _ = tiny
    x = pkg.synth.synth_audio(3.0, seed=11); t = pkg.TimeEmbedding(256).embed(6.0)
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)     # [128][T], transcribe.rs:295-305
    rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
    ids = m.transcribe_audio(x, t)
    assert len(ids) == len(rids) == o.enc_seq_len(mel.shape[1]) // 4 - 38
    check_greedy_ids(ids, rids, rlg, TOL)
    d = ctx.upload(x)                                                        # device-resident samples (bench path)
    ids_d = m.transcribe_audio(None, t, device_ptr=d, n_samples=x.size)
    assert (ids_d == ids).all()
    ctx.free(d)
    tm = m.timings()
    assert tm["preprocess_ms"] > 0 and tm["encode_ms"] > 0 and tm["decode_ms"] > 0


def test_golden_reference_python(pkg, ctx):
    """HIP path vs the golden vectors generated by the reference's own PyTorch restatement
    (tests/golden/make_golden.py): 32-layer encoder + adapter, 26-layer GQA decoder + tied lm_head."""
    g = golden(); path, dims = golden_gguf()
    m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    assert (m.config.enc_layers, m.config.enc_heads, m.config.dec_layers, m.config.dec_heads, m.config.dec_kv_heads) == (32, 32, 26, 32, 8)
    out = m.encode_audio(g["in_mel"][None])
    assert out.shape == (1, 10, 256) and rel_err(out[0], g["out_encoder_out"]) < 3e-4, rel_err(out[0], g["out_encoder_out"])
    dec = m.decoder(); t = pkg.TimeEmbedding(256).embed(6.0)
    assert np.abs(t - g["out_time_embedding_6"]).max() < 1e-5
    c = dec.create_cache_preallocated(16)
    hid = dec.forward_hidden_with_cache(g["in_dec_x"][None], t, c)
    assert rel_err(hid[0], g["out_decoder_hidden"]) < 3e-4
    lg = dec.lm_head(hid)
    assert rel_err(lg[0], g["out_decoder_logits"]) < 3e-4
    assert (lg[0].argmax(1) == g["out_decoder_logits"].argmax(1)).all()
    c.reset()
    h1 = dec.forward_hidden_with_cache(g["in_dec_x"][None, :8], t, c)
    hs = [dec.forward_hidden_with_cache(g["in_dec_x"][None, i:i + 1], t, c) for i in range(8, 12)]      # decode-step kernels
    assert rel_err(np.concatenate([h1[0]] + [h[0] for h in hs]), g["out_decoder_hidden"]) < 3e-4
    m.close()


def test_transcribe_batch(pkg, orc, tiny):
    """Batched decode (vox_transcribe_batch): ragged batch of independent utterances == per-utterance oracle / single-stream path."""
    m, o, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    clips = [pkg.synth.synth_audio(sec, seed=30 + i) for i, sec in enumerate((2.0, 3.3, 0.4, 2.0))]
    outs = m.transcribe_batch(clips, t)
    assert len(outs) == 4
    tm = m.timings()
    assert tm["decode_tokens"] == sum(len(x) for x in outs) and tm["graph_replays"] > 0
    for x, ids in zip(clips, outs):
        xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
        mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
        rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
        assert len(ids) == len(rids) == o.enc_seq_len(mel.shape[1]) // 4 - 38
        check_greedy_ids(ids, rids, rlg, TOL)
    single = [m.transcribe_audio(x, t) for x in clips]
    outs2 = m.transcribe_batch(clips, t)
    for a, b, c in zip(outs, outs2, single):
        assert (a == b).all()                               # deterministic
        assert len(a) == len(c)
    assert len(m.transcribe_batch(clips[:1], t)[0]) == len(single[0])


def test_transcribe_batch_wide_and_fallback_paths(pkg, ctx, tiny, monkeypatch):
    """More than 16 utterances (two 16-row groups), exactly 16 (one XF group) and the VOX_BATCH_NO_XF knob (f32-activation step): EVERY
    sequence must give the single-stream ids up to the first near-tie of its own single-stream logits (check_greedy_ids per sequence)."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    secs = [2.0 + 0.17 * (i % 5) for i in range(20)]
    clips = [pkg.synth.synth_audio(s, seed=70 + i) for i, s in enumerate(secs)]
    wide = m.transcribe_batch(clips, t)                       # n = 20 > 16
    exact = m.transcribe_batch(clips[:16], t)                 # n = 16: XF step
    monkeypatch.setenv("VOX_BATCH_NO_XF", "1")
    noxf = m.transcribe_batch(clips[:16], t)
    monkeypatch.delenv("VOX_BATCH_NO_XF")
    n_same = check_batch_rows(pkg, ctx, m, clips, t, wide, TOL) + check_batch_rows(pkg, ctx, m, clips[:16], t, exact, TOL) + \
        check_batch_rows(pkg, ctx, m, clips[:16], t, noxf, TOL)
    print(f"batched paths: {n_same}/52 sequences identical to single-stream end to end (the rest diverge at a verified near-tie)")
    with pytest.raises(pkg.VoxError):
        m.transcribe_batch([clips[0]] * 4097, t)              # batch size limit (1..4096)


def test_transcribe_batch_ragged_groups_retire(pkg, ctx, tiny, monkeypatch):
    """40 utterances of very different lengths in arbitrary order (three 16-row groups): rows run longest first and a group is retired when its longest
    member is done -- every caller slot must get exactly the ids of the run without sorting / retirement, the single-stream ids (up to a near-tie), and
    the right count."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    secs = [0.4 + 0.23 * ((7 * i) % 19) for i in range(40)]               # 0.4 .. 4.5 s, arbitrary order
    clips = [pkg.synth.synth_audio(s, seed=900 + i) for i, s in enumerate(secs)]
    outs = m.transcribe_batch(clips, t)
    tm = m.timings(); assert tm["decode_tokens"] == sum(len(o) for o in outs)
    monkeypatch.setenv("VOX_BATCH_NO_RETIRE", "1"); monkeypatch.setenv("VOX_BATCH_NO_SORT", "1")
    ref = m.transcribe_batch(clips, t)
    monkeypatch.delenv("VOX_BATCH_NO_RETIRE"); monkeypatch.delenv("VOX_BATCH_NO_SORT")
    assert len(outs) == len(ref) == 40 and len({len(o) for o in outs}) > 8          # really ragged
    for r, (a, b) in enumerate(zip(outs, ref)):
        assert len(a) == len(b) and (a == b).all(), f"slot {r}: retiring / sorting changed the ids"
    n_same = check_batch_rows(pkg, ctx, m, clips, t, outs, TOL)
    print(f"ragged batch with retiring groups: {n_same}/40 sequences identical to single-stream end to end")
    again = m.transcribe_batch(clips, t)
    assert all((a == b).all() for a, b in zip(outs, again))               # deterministic (graphs are re-captured per call)


def test_transcribe_batch_continuous_slots(pkg, ctx, tiny, monkeypatch):
    """Round 5: batches wider than one group run as CONTINUOUS batching over decode slots (vox_api.cpp transcribe_continuous_impl): every utterance is encoded and prefilled
    up front, the host packs them longest-first onto 16 G slots, and a slot whose utterance gets its last token takes the next one of its queue inside the step's argmax /
    next-input launch (argmax_embed_slots_kernel; cache slices picked per slot through kv_row).  90 utterances (two encoder chunks of <= 64) of 0.4 .. 4.5 s in arbitrary
    order, some too short to take a decode step at all: every caller slot must get exactly the ids of (a) the lock-step batches of <= 64 (VOX_BATCH_NO_CONTINUOUS=1), (b) every
    forced slot-group count 1 .. 4 (different packings, different slots: rows are independent of where they run), (c) the single-stream path up to a near-tie; deterministic."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    secs = [0.4 + 0.23 * ((7 * i) % 19) for i in range(90)]
    secs[5] = 0.05; secs[41] = 0.12                                           # 76 + 17 pad tokens + almost nothing: S = 46 / 47 -> a handful of steps
    clips = [pkg.synth.synth_audio(s, seed=1900 + i) for i, s in enumerate(secs)]
    outs = m.transcribe_batch(clips, t)
    tm = m.timings(); assert tm["decode_tokens"] == sum(len(o) for o in outs) and tm["graph_replays"] > 0
    assert len(outs) == 90 and len({len(o) for o in outs}) > 8
    monkeypatch.setenv("VOX_BATCH_NO_CONTINUOUS", "1")
    ref = m.transcribe_batch(clips, t)                                        # two lock-step batches of 45
    monkeypatch.delenv("VOX_BATCH_NO_CONTINUOUS")
    for r, (a, b) in enumerate(zip(outs, ref)):
        assert len(a) == len(b) and (a == b).all(), f"slot {r}: continuous batching changed the ids"
    for G in (1, 2, 3, 4, 5, 6):      # (5, 6: 80 / 96 slots -- two wide chains of ceil / floor halves per step, end of round 6)
        monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", str(G))
        alt = m.transcribe_batch(clips, t)
        assert all(len(a) == len(b) and (a == b).all() for a, b in zip(outs, alt)), f"{G} slot groups: ids differ"
    monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", "6"); monkeypatch.setenv("VOX_BATCH_VERBOSE", "1")
    m.transcribe_batch(clips, t)
    monkeypatch.delenv("VOX_BATCH_VERBOSE")
    # round 6: the WIDE step (launch_q4_wide: the active groups' layer operators as one GEMM + finishing launch each, one attention launch, one lm_head) at 2, 3 and 4 groups
    monkeypatch.setenv("VOX_BATCH_WIDE_MIN", "2"); monkeypatch.setenv("VOX_BATCH_CONT_NO_ENGINE", "1")
    for G in (2, 3, 4):
        monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", str(G))
        alt = m.transcribe_batch(clips, t)
        assert all(len(a) == len(b) and (a == b).all() for a, b in zip(outs, alt)), f"wide step, {G} slot groups: ids differ"
    monkeypatch.delenv("VOX_BATCH_WIDE_MIN"); monkeypatch.delenv("VOX_BATCH_CONT_NO_ENGINE")
    monkeypatch.delenv("VOX_BATCH_SLOT_GROUPS")
    n_same = check_batch_rows(pkg, ctx, m, clips[:24], t, outs[:24], TOL)
    again = m.transcribe_batch(clips, t)
    assert all((a == b).all() for a, b in zip(outs, again))
    few = m.transcribe_batch(clips[:17], t)                                   # 17 utterances: the narrowest continuous batch
    assert all((a == b).all() for a, b in zip(few, outs[:17]))
    ptrs = [ctx.upload(c) for c in clips[:40]]                                # device-resident samples (VOX_MEM_DEVICE), two and a half groups
    try:
        dev = m.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=[c.size for c in clips[:40]])
        assert all((a == b).all() for a, b in zip(dev, outs[:40]))
    finally:
        for p_ in ptrs:
            ctx.free(p_)
    print(f"continuous batching, 90 ragged utterances: identical to the lock-step batches and for 1..4 slot groups; {n_same}/24 checked rows identical to single-stream end to end")


def test_transcribe_batch_ex_chunks_share_their_files_peak(pkg, ctx, tiny):
    """vox_transcribe_batch_ex: the reference's CLI peak-normalises the FILE, then chunks it (bin/transcribe.rs:207-226) -- a chunk is normalised by its
    file's peak, not its own.  Units = the 600-frame chunks of five files with different peaks (one of them with a SILENT chunk, one silent altogether) + short files, in one
    call: (a) norm_group = file index on the RAW samples (group peaks reduced on the device), host and device pointers, (b) norm_group = -1 on host-normalised samples --
    both must give, unit for unit, exactly the ids of the same call made on separately normalised copies, which equal the serial path (pad -> log-mel ->
    transcribe_streaming per chunk, cli.transcribe_one) up to a near-tie."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0); S = pkg.synth
    cfg = pkg.ChunkConfig.voxtral().with_max_frames(600)                       # 6 s chunks
    files = [0.31 * S.synth_audio(14.0, seed=41), 0.8 * S.synth_audio(7.5, seed=42), 0.11 * S.synth_audio(19.0, seed=43)]
    files[0][:96000] *= np.float32(1e-4)      # a first chunk 80 dB below the file's peak: under the FILE's scale its log-mel sits at the floor, under its own it would be full scale
    q = 0.5 * S.synth_audio(13.0, seed=44); q[:96000] = 0.0; files.append(q)                      # a file whose FIRST chunk is silent (its own peak would give scale 1)
    files.append(np.zeros(100000, np.float32))                                                   # a silent file: scale 1 (audio/io.rs:61-63)
    files += [(0.05 + 0.04 * i) * S.synth_audio(1.0 + 0.5 * i, seed=50 + i) for i in range(9)]
    raw, nrm, grp = [], [], []
    for fi, x in enumerate(files):
        xn = pkg.peak_normalize(x, 0.95)
        plan = pkg.chunk_plan(x.size, cfg) if pkg.needs_chunking(x.size, cfg) else [(0, x.size)]
        for a, b in plan:
            raw.append(x[a:b]); nrm.append(xn[a:b]); grp.append(1000 + 7 * fi)                    # arbitrary (non-dense) group ids
    assert len(raw) == 3 + 2 + 4 + 3 + 2 + 9 and len(raw) > 16
    by_group = m.transcribe_batch(raw, t, norm_group=grp)
    as_is = m.transcribe_batch(nrm, t, norm_group=[-1] * len(nrm))
    for u, (a, b) in enumerate(zip(by_group, as_is)):
        assert len(a) == len(b) and (a == b).all(), f"unit {u}: device-side group normalisation != host-normalised file"
    ptrs = [ctx.upload(np.ascontiguousarray(c)) for c in raw]
    try:
        dev = m.transcribe_batch(None, t, device_ptrs=ptrs, n_samples=[c.size for c in raw], norm_group=grp)
        assert all((a == b).all() for a, b in zip(dev, by_group))
    finally:
        for p_ in ptrs:
            ctx.free(p_)
    mixed = m.transcribe_batch(raw[:5] + nrm[5:], t, norm_group=grp[:5] + [-1] * (len(raw) - 5))   # groups and as-is units in one call
    assert all((a == b).all() for a, b in zip(mixed, by_group))
    few = m.transcribe_batch(raw[:5], t, norm_group=grp[:5])                                       # <= 16 units: the one-group path; files 0 and 1 complete
    assert all((a == b).all() for a, b in zip(few, by_group[:5]))
    mel = pkg.MelSpectrogram.voxtral(ctx); n_same = 0
    for u, c in enumerate(nrm):                                                                    # the serial CLI path, chunk by chunk
        rids, rlg = m.transcribe_streaming(np.ascontiguousarray(mel.compute_log(pkg.pad_audio(c)).T)[None], t, return_logits=True)
        n_same += int(check_greedy_ids(by_group[u], rids, rlg, TOL) == len(rids))
    # (that a chunk normalised by ITS OWN peak decodes differently is asserted at full size, test_full_chunked_files_are_units_of_the_wide_batch: this tiny random
    # model's argmax does not depend on its input)
    with pytest.raises(ValueError):
        m.transcribe_batch(raw, t, norm_group=grp[:-1])
    print(f"vox_transcribe_batch_ex: {len(raw)} chunk units of {len(files)} files; device group peaks == host-normalised files; {n_same}/{len(raw)} units identical to the serial path end to end")


def test_model_set_sessions_splits_large_calls_and_keeps_every_units_ids(pkg, ctx, tiny):
    """vox_model_set_sessions(m, 2): a batch call with >= 128 units per session runs as two concurrent sessions inside the library (hidden context + replica + library
    thread); ids, their order and n_ids per unit equal the one-session call's -- un-chunked, with normalisation groups (a group's units stay in one session: its peak is a
    device reduction per session) and with already-normalised units; smaller calls stay on one session; sessions = 1 frees the replicas; bad arguments fail."""
    m, o, _ = tiny
    t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
    rng = np.random.default_rng(3)
    clips = [pkg.synth.synth_audio(0.6 + 0.05 * (i % 23), seed=7000 + i) * float(0.2 + 0.8 * rng.random()) for i in range(300)]
    grp = [i // 3 for i in range(300)]      # 100 "files" of three units
    mixed = [g if g % 2 else -1 for g in grp]
    ctx.set_shared(True)      # (the reference runs under the same rules the split sessions run under: launch chains, table costs)
    try:
        ref = m.transcribe_batch(clips, t); ref_g = m.transcribe_batch(clips, t, norm_group=grp); ref_m = m.transcribe_batch(clips, t, norm_group=mixed)
    finally:
        ctx.set_shared(False)
    m.set_sessions(2)
    try:
        for rep in range(2):
            got = m.transcribe_batch(clips, t)
            assert len(got) == 300 and all(np.array_equal(a, b) for a, b in zip(ref, got)), rep
        tm = m.timings(); assert tm["decode_tokens"] == sum(len(g) for g in got) and tm["total_ms"] > 0
        assert all(np.array_equal(a, b) for a, b in zip(ref_g, m.transcribe_batch(clips, t, norm_group=grp)))
        assert all(np.array_equal(a, b) for a, b in zip(ref_m, m.transcribe_batch(clips, t, norm_group=mixed)))
        small = m.transcribe_batch(clips[:40], t)      # 40 units: one session
        assert all(np.array_equal(a, b) for a, b in zip(ref[:40], small))
        m.set_sessions(3); got3 = m.transcribe_batch(clips, t)      # 300 units / 128 = 2 sessions used of the 3
        assert all(np.array_equal(a, b) for a, b in zip(ref, got3))
    finally:
        m.set_sessions(1)
    assert all(np.array_equal(a, b) for a, b in zip(ref[:20], m.transcribe_batch(clips[:20], t)))      # the replicas are gone, the model works as before
    for bad in (0, 5):
        with pytest.raises(pkg.VoxError):
            m.set_sessions(bad)
    # vox_ctx_set_shared on its own: a shared context stays off the batched decode engines (their persistent workgroups need the GPU to themselves), same ids
    if m.set_batch_engine()[0]:
        n0 = m.set_batch_engine()[1]
        ctx.set_shared(True)
        try:
            sh = m.transcribe_batch(clips[:20], t)
            assert m.set_batch_engine()[1] == n0
        finally:
            ctx.set_shared(False)
        own = m.transcribe_batch(clips[:20], t)
        assert m.set_batch_engine()[1] > n0 and all(np.array_equal(a, b) for a, b in zip(sh, own))


def test_two_contexts_two_threads_and_model_replicate(pkg, ctx, tiny):
    """The header promises "a handle is not thread-safe, distinct contexts are independent" (include/voxtral_hip.h:15-16) and SURVEY.md section 8(e) describes the
    in-process multi-GPU shape -- one host thread + one vox_ctx + one model replica per GPU (what a Rust `voxtral-transcribe`, bin/transcribe.rs:60-128, would do).
    Without a second GPU: TWO contexts on device 0, each with its own replica (the second made by vox_model_replicate: layout from the manifest, device-to-device copy
    of the primary arena, derived copies rebuilt -- no file, no collective), driven CONCURRENTLY from two host threads on the per-operator path (the persistent engines
    need the whole GPU to themselves): single-stream transcriptions, a batch and a wide continuous batch per thread; every result must equal the serial run's."""
    import threading
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0); S = pkg.synth
    ctx2 = pkg.Context(0)
    m2 = m.replicate(ctx2)
    try:
        assert m2.weight_bytes() == m.weight_bytes() and m2.arena()[0] != m.arena()[0]
        for mm in (m, m2):
            mm.set_decode_engine(False); mm.set_batch_engine(False)
        clips = [S.synth_audio(0.6 + 0.17 * ((3 * i) % 11), seed=3100 + i) for i in range(40)]
        work = {0: clips[:20], 1: clips[20:]}
        def job(mm, mine):
            return [mm.transcribe_audio(x, t) for x in mine[:6]], mm.transcribe_batch(mine[:9], t), mm.transcribe_batch(mine, t)
        serial = {k: job(m, work[k]) for k in (0, 1)}                       # reference: everything on the first context, one after the other
        assert all((a == b).all() for a, b in zip(job(m2, work[1])[2], serial[1][2]))      # the replica alone gives the same ids
        res, err = {}, []
        def run(k, mm):
            try:
                for _ in range(3):
                    res[k] = job(mm, work[k])
            except Exception as e:                                          # noqa: BLE001
                err.append(e)
        th = [threading.Thread(target=run, args=(0, m)), threading.Thread(target=run, args=(1, m2))]
        [x.start() for x in th]; [x.join(300) for x in th]
        assert not err, err
        for k in (0, 1):
            for got, want in zip(res[k], serial[k]):
                assert len(got) == len(want) and all(len(a) == len(b) and (a == b).all() for a, b in zip(got, want)), f"thread {k}: concurrent result differs from the serial run"
    finally:
        m.set_decode_engine(True); m.set_batch_engine(True)
        m2.close(); ctx2.close()


def test_wide_step_measurement_hook(pkg, tiny):
    """vox_bench_wide (tools/wide_bench.py's hook): every operator of the wide step at 2, 3 and 4 slot groups launches and is timed on the tiny model's shapes too
    (K = 256 / 512: one or two K steps per slice, the run-time step loop's shortest trips)."""
    import ctypes as C
    m, _, _ = tiny
    for mt in (2, 3, 4):
        for which in range(5):
            out = (C.c_double * 4)()
            assert pkg.lib().vox_bench_wide(m.h, which, mt, 2, out) == 0, (mt, which, pkg.lib().vox_last_error())
            assert out[0] > 0 and out[2] > 0 and out[3] > 0 and (which == 4 or out[1] > 0)
    out = (C.c_double * 4)()
    assert pkg.lib().vox_bench_wide(m.h, 7, 2, 2, out) != 0 and pkg.lib().vox_bench_wide(m.h, 0, 5, 2, out) != 0      # bad operator / group count


def test_transcribe_exactly_prefix_len(pkg, orc, tiny):
    """S == 38 decoder positions (= PREFIX_LEN): the reference prefills, predicts the first token and returns ONE id (gguf/model.rs:887-889
    only returns empty below 38; the decode loop :938 is empty).  T = 606 mel frames -> 303 -> 152 encoder rows -> 38."""
    m, o, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    for T, want in ((606, 1), (624, 1), (640, 2), (602, 0)):      # S = 38, 39 (loop empty: 1 id), 40 (one step), 37
        mel = fake_mel(T, seed=T)
        S = o.enc_seq_len(T) // 4
        assert (S - 38 if S > 38 else (1 if S == 38 else 0)) == want, (T, S)
        rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
        ids, lg = m.transcribe_streaming(mel[None], t, return_logits=True)
        assert len(rids) == want == len(ids)
        if want:
            _check_ids(ids, lg, rids, rlg)
            assert (m.transcribe_streaming(mel[None], t) == ids).all()     # graph / no-logits path


def test_audio_entry_points_never_reach_prefix_len_edge(pkg, ctx, tiny):
    """Through the audio entry points (vox_transcribe_audio / _batch) the S == 38 edge cannot occur: the 76 + 17 pad tokens alone give
    T >= 744 frames -> S >= 46 (pad.rs:32-46), so even a 10 ms clip emits S - 38 >= 8 ids, the same on both paths."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    clips = [pkg.synth.synth_audio(0.01, seed=3), pkg.synth.synth_audio(1.0, seed=4)]      # 160 samples -> T = 752 -> S = 47
    a = m.transcribe_audio(clips[0], t); b = m.transcribe_batch(clips, t)
    assert len(a) == len(b[0]) == 47 - 38
    check_batch_rows(pkg, ctx, m, clips, t, b, TOL)


def test_token_buffer_survives_longer_sequence(pkg, ctx, tiny):
    """Regression (round-1 advisor finding): the captured decode graph bakes the token-buffer pointer in; a later, longer utterance must not
    make the replayed graph read / write a re-allocated buffer.  S = 800 builds cache + graph, S = 1023 then needs S + 2 > 1024 tokens."""
    m, _, _ = tiny
    t = pkg.TimeEmbedding(256).embed(6.0)
    def mel_for(S):
        return fake_mel(16 * S, seed=S)
    a = m.transcribe_streaming(mel_for(800)[None], t)
    b = m.transcribe_streaming(mel_for(1023)[None], t)
    m2 = pkg.Q4ModelLoader.from_file(tiny_gguf()[0]).load(ctx)            # fresh model: no cached graph / buffers
    b2, lg = m2.transcribe_streaming(mel_for(1023)[None], t, return_logits=True)
    assert len(a) == 800 - 38 and len(b) == 1023 - 38
    check_greedy_ids(b, b2, lg, TOL)
    assert (m.transcribe_streaming(mel_for(1023)[None], t) == b).all()
    m2.close()


def test_load_from_bytes_and_shards(pkg, ctx, tiny):
    """Q4ModelLoader::from_bytes / from_shards (gguf/loader.rs:92-107): the same model as from_file -> identical ids."""
    m, _, _ = tiny
    img = open(tiny_gguf()[0], "rb").read()
    t = pkg.TimeEmbedding(256).embed(6.0); x = pkg.synth.synth_audio(2.5, seed=4)
    ref = m.transcribe_audio(x, t)
    mb = pkg.Q4ModelLoader.from_bytes(img).load(ctx)
    third = len(img) // 3
    ms = pkg.Q4ModelLoader.from_shards([img[:third], img[third:2 * third], img[2 * third:]]).load(ctx)
    assert (mb.transcribe_audio(x, t) == ref).all() and (ms.transcribe_audio(x, t) == ref).all()
    assert mb.weight_bytes() == m.weight_bytes() == ms.weight_bytes()
    mb.close(); ms.close()


def test_abi_smoke_c_program_on_device(pkg, tmp_path):
    """tests/abi_smoke.c (C11 translation unit over include/voxtral_hip.h) with a GPU: context, GGUF reader, model load, vox_transcribe_audio, stage timings,
    vox_generate_step_with_cache, cache accessors -- every call through the C header's own declarations."""
    import subprocess
    from test_abi_cpu import _build_abi_smoke
    path, _ = tiny_gguf()
    r = subprocess.run([_build_abi_smoke(tmp_path), path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "device path ok" in r.stdout, r.stdout


def test_e2e_piecewise_c_program_on_device_tiny(pkg, tmp_path):
    """tools/e2e_piecewise.c on the tiny model (per-operator launches: the engine needs the real geometry): the reference's decode loop call for call over the
    device-resident decoder surface reproduces vox_transcribe_audio's ids, with lm_head + argmax and with vox_lm_head_argmax."""
    import json, subprocess
    from test_abi_cpu import _build_e2e_piecewise
    path, _ = tiny_gguf()
    wav = tmp_path / "clip.f32"; pkg.synth.synth_audio(6.0, seed=5).astype(np.float32).tofile(str(wav))
    r = subprocess.run([_build_e2e_piecewise(tmp_path), path, str(wav), "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["lm_head+argmax"]["ids_equal_transcribe_audio"] and res["lm_head_argmax"]["ids_equal_transcribe_audio"] and res["lm_head+argmax"]["decode_tokens"] > 20


def test_generate_step_with_cache_equals_composed_calls(pkg, tiny):
    """Q4VoxtralModel::generate_step_with_cache (gguf/model.rs:857-867) as ONE C-ABI call against embed_tokens_from_ids -> forward_hidden_with_cache -> lm_head:
    the same kernels in the same order, so the logits are bit-identical, for a multi-token prefix and for single-token steps; the cache advances alike."""
    m = tiny[0]
    t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
    dec = m.decoder()
    ids = np.array([1, 32, 32, 5, 7], dtype=np.int32)
    ca = dec.create_cache_preallocated(32); cb = dec.create_cache_preallocated(32)
    la = m.generate_step_with_cache(ids, t, ca)
    hb = dec.forward_hidden_with_cache(dec.embed_tokens_from_ids(ids, 1, ids.size), t, cb)
    lb = dec.lm_head(hb)[0]
    assert la.shape == lb.shape == (5, m.config.vocab) and np.array_equal(la, lb)
    for tok in (9, 11):
        one = np.array([tok], dtype=np.int32)
        la = m.generate_step_with_cache(one, t, ca)
        lb = dec.lm_head(dec.forward_hidden_with_cache(dec.embed_tokens_from_ids(one, 1, 1), t, cb))[0]
        assert np.array_equal(la, lb)
    assert ca.seq_len() == cb.seq_len() == 7


def test_composite_forwards_equal_composed_calls_and_the_oracle(pkg, orc, tiny):
    """Q4VoxtralModel::forward / forward_streaming / forward_with_cache (gguf/model.rs:802-843: mel -> logits in one call) against the pieces they are made of
    (encode_audio -> [+ embed] -> forward_hidden_with_cache on a fresh cache -> lm_head: bit-identical) and against the oracle's composition (<= 2e-4)."""
    m, o = tiny[0], tiny[1]
    t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
    mel = fake_mel(700, seed=11); dec = m.decoder()
    audio = m.encode_audio(mel[None])[0]; S = audio.shape[0]
    ids = (np.arange(S) * 7 % 200 + 3).astype(np.int32)
    # forward: audio embeddings alone
    la = m.forward(mel[None], t)[0]
    c = dec.create_cache_preallocated(max(S, 8)); lb = dec.lm_head(dec.forward_hidden_with_cache(audio[None], t, c))[0]
    assert la.shape == (S, m.config.vocab) and np.array_equal(la, lb)
    oc = o.cache(max(S, 8)); ra = o.encode_audio(mel); rl = o.lm_head(o.forward_hidden_with_cache(ra, t, oc)); o.cache_free(oc)
    assert rel_err(la, rl) < 3e-4
    # forward_streaming: + one token embedding per position
    ls = m.forward_streaming(mel[None], ids, t)[0]
    c2 = dec.create_cache_preallocated(max(S, 8)); x = audio + dec.embed_tokens_from_ids(ids, 1, S)[0]
    assert np.array_equal(ls, dec.lm_head(dec.forward_hidden_with_cache(x[None], t, c2))[0])
    with pytest.raises(pkg.VoxError):
        m.forward_streaming(mel[None], ids[:-1], t)                      # one id per audio position
    # forward_with_cache: two chunks through the streaming encoder and the cached decoder == the two-call composition on twin caches
    ea, da = m.create_encoder_cache(), dec.create_cache_preallocated(256); eb, db = m.create_encoder_cache(), dec.create_cache_preallocated(256)
    for lo_, hi_ in ((0, 352), (352, 700)):
        chunk = np.ascontiguousarray(mel[:, lo_:hi_])
        lw = m.forward_with_cache(chunk[None], t, ea, da)[0]
        au = m.encode_audio_with_cache(chunk[None], eb)
        assert np.array_equal(lw, dec.lm_head(dec.forward_hidden_with_cache(au, t, db))[0])
    assert da.seq_len() == db.seq_len() > 0


@pytest.mark.parametrize("scheme", ["reference", "ggml"])
def test_exported_gguf_round_trip_on_device(pkg, orc, ctx, tmp_path, scheme):
    """SURVEY 8(f4) on the GPU: a dense checkpoint (SafeTensors) exported by export.py (reference test quantiser gguf/tests.rs:24-57, and llama.cpp's quantize_row_q4_0_ref)
    -> vox_q4_model_load -> whole hot path, against the CPU oracle loading THE SAME exported file: encoder output and decoder logits within the stated 2e-4, greedy ids
    equal up to a near-tie; and the Q4 model stays close to the dense model it was exported from (quantisation error, not a format error)."""
    S = pkg.synth; E = pkg.export
    d = S.tiny_dims()
    st = str(tmp_path / "dense.safetensors"); S.write_synthetic_safetensors(st, d, seed=5)
    out = str(tmp_path / f"exported_{scheme}.gguf")
    stats = E.export_q4_gguf(st, out, scheme=scheme)
    assert stats["q4"] > 0
    m = pkg.Q4ModelLoader.from_file(out).load(ctx); o = orc.Model(out)
    try:
        t = pkg.TimeEmbedding(d.dec_dim).embed(6.0); mel = fake_mel(900, seed=11)
        a_ref = o.encode_audio(mel); a_hip = m.encode_audio(mel[None])[0]
        assert rel_err(a_hip, a_ref) < 3e-4, rel_err(a_hip, a_ref)
        ids, lg = m.transcribe_streaming(mel[None], t, return_logits=True)
        rids, rlg = o.transcribe_streaming(mel, t, want_logits=True)
        stop = check_greedy_ids(ids, rids, rlg, 2e-4)
        assert stop >= 1 and rel_err(lg[:stop], rlg[:stop]) < 2e-4
        # the dense model the file was exported from: same geometry, outputs within quantisation error (loose, structural check)
        md = pkg.VoxtralModelLoader.from_file(st).load(ctx)
        try:
            a_dense = md.encode_audio(mel[None])[0]
            assert a_dense.shape == a_hip.shape and rel_err(a_hip, a_dense) < 0.5
        finally:
            md.close()
    finally:
        m.close(); o.close()
