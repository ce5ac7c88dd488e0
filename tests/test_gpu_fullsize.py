"""Parity at BASELINE.json's FULL sizes (the real Voxtral-Mini-4B-Realtime shapes, synthetic weights): the oracle is too
slow to replay a whole 16 s clip in a test, so full-size coverage is (a) the oracle on a bounded slice -- a short clip's
encoder output and its first decode steps (logits within tolerance) -- and (b) size-independent properties of the HIP path:
graph-replayed == eager ids, run-to-run determinism, device-pointer == host-pointer path, batch row == single row,
linearity of the Q4 operator at the lm_head shape, and the 16 s clip geometry of the published metric."""
import os

import numpy as np
import pytest

from model_fixtures import cache_dir, check_batch_rows, check_greedy_ids, dense_head_sha, full_dense_safetensors, rel_err

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(autouse=True)
def _engine_default(request):
    """Every test starts with the model's default decode path (the persistent engine where the device allows it)."""
    yield
    if "full" in request.fixturenames:
        request.getfixturevalue("full")[0].set_decode_engine(True)


@pytest.fixture(scope="module")
def full(pkg, orc):
    S = pkg.synth
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    if not os.path.exists(path):
        S.write_synthetic_gguf(path + ".tmp", S.ModelDims(), seed=42); os.replace(path + ".tmp", path)
    ctx = pkg.Context(0)
    m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    o = orc.Model(path)
    yield m, o, ctx
    m.close(); o.close(); ctx.close()


def test_full_config(full):
    m, o, _ = full
    c = m.config
    assert (c.enc_layers, c.enc_dim, c.enc_heads, c.enc_ffn, c.dec_layers, c.dec_dim, c.dec_heads, c.dec_kv_heads, c.dec_ffn, c.vocab) == \
           (32, 1280, 32, 5120, 26, 3072, 32, 8, 9216, 131072)                         # gguf/loader.rs:567-591, config.rs tests
    assert 2.4e9 < m.weight_bytes()


def test_full_short_clip_vs_oracle(pkg, orc, full):
    """0.5 s clip: whole pipeline vs the oracle at full model size (encoder S_enc=143, 35 decoder positions short of 38 -> use 1.2 s)."""
    m, o, _ = full
    x = pkg.synth.synth_audio(1.2, seed=77); t = pkg.TimeEmbedding(3072).embed(6.0)
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
    S = o.enc_seq_len(mel.shape[1]) // 4
    assert S >= 40
    ref_audio = o.encode_audio(mel)
    out_audio = m.encode_audio(mel[None])[0]
    assert rel_err(out_audio, ref_audio) < 5e-4, rel_err(out_audio, ref_audio)         # 32 layers of accumulated f32-class error
    # decoder: prefill + 3 steps on the oracle's own audio embeddings (isolates the decoder from encoder error)
    dec = m.decoder(); ids = np.array([1] + [32] * 37, dtype=np.int32)
    x0 = ref_audio[:38] + o.embed_tokens(ids)
    assert (dec.embed_tokens_from_ids(ids, 1, 38)[0] == o.embed_tokens(ids)).all()
    oc = o.cache(64); c = dec.create_cache_preallocated(64)
    rh = o.forward_hidden_with_cache(x0, t, oc); gh = dec.forward_hidden_with_cache(x0[None], t, c)[0]
    assert rel_err(gh, rh) < TOL, rel_err(gh, rh)
    rl = o.lm_head(rh[-1:]); gl = dec.lm_head(gh[None, -1:])[0]
    assert gl.shape == (1, 131072) and rel_err(gl, rl) < TOL
    tok = int(rl.argmax())
    for step in range(3):
        xs = ref_audio[38 + step:39 + step] + o.embed_tokens(np.array([tok], np.int32))
        rh = o.forward_hidden_with_cache(xs, t, oc); gh = dec.forward_hidden_with_cache(xs[None], t, c)[0]      # decode-step GEMV kernels
        assert rel_err(gh, rh) < TOL
        rl = o.lm_head(rh); gl = dec.lm_head(gh[None])[0]
        assert rel_err(gl, rl) < TOL and int(gl.argmax()) == int(rl.argmax())
        tok = int(rl.argmax())
    o.cache_free(oc)


def test_full_16s_clip_properties(pkg, full):
    """The published-metric workload (16 s, un-chunked): geometry + eager == graph == device-pointer path, deterministic."""
    m, _, ctx = full
    x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
    assert pkg.PadConfig.voxtral().padded_len(x.size) == 375040
    mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
    assert mel.shape == (2344, 128)
    ids_e, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)   # eager, logits tap
    assert len(ids_e) == 108 and lg.shape == (108, 131072) and (lg.argmax(1) == ids_e).all()
    ids_g = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t)                  # hipGraph replay
    ids_a = m.transcribe_audio(x, t)                                                      # whole path from samples (device mel)
    d = ctx.upload(x); ids_d = m.transcribe_audio(None, t, device_ptr=d, n_samples=x.size); ctx.free(d)
    assert (ids_g == ids_e).all() and (ids_a == ids_d).all() and len(ids_a) == 108
    srt = np.sort(lg, axis=1); safe = (srt[:, -1] - srt[:, -2]) > 10 * TOL * max(1.0, np.abs(lg).max())
    stop = len(safe) if safe.all() else int(np.argmin(safe))
    assert (ids_a[:stop] == ids_e[:stop]).all()                                           # device mel vs host-fed mel: same ids up to a near-tie
    assert (m.transcribe_audio(x, t) == ids_a).all()                                      # run-to-run deterministic
    tm = m.timings(); assert tm["decode_tokens"] == 108 and tm["graph_replays"] >= 105


def test_full_batch_rows_match_single(pkg, full):
    """Batched path at full size (XF fragment step, folded RMSNorm, GQA-grouped decode attention: needs >= 4 sequences and the real
    32:8 head ratio) against the single-stream path, ragged lengths: EVERY sequence equals the single-stream ids up to the first near-tie of
    its own single-stream logits."""
    m, _, ctx = full
    t = pkg.TimeEmbedding(3072).embed(6.0)
    clips = [pkg.synth.synth_audio(s, seed=50 + i) for i, s in enumerate((2.0, 3.0, 2.0, 2.6, 3.4))]
    outs = m.transcribe_batch(clips, t)
    n_same = check_batch_rows(pkg, ctx, m, clips, t, outs, TOL)
    again = m.transcribe_batch(clips, t)
    assert all((a == b).all() for a, b in zip(outs, again))
    os.environ["VOX_ATTN_GQA"] = "1"                          # the wide-batch attention kernel (normally > 16 utterances) on the same batch
    try:
        gqa = m.transcribe_batch(clips, t)
    finally:
        del os.environ["VOX_ATTN_GQA"]
    n_same += check_batch_rows(pkg, ctx, m, clips, t, gqa, TOL)
    print(f"full-size batch: {n_same}/10 sequences identical to single-stream end to end")


def test_q4_operator_linearity_at_lm_head_shape(pkg, ctx_full=None):
    """Size-independent property at the largest operator shape (131072 x 3072): W(ax + by) == a Wx + b Wy within f32 round-off,
    and a batch of rows equals the same rows one at a time (GEMV vs skinny vs MFMA GEMM paths)."""
    ctx = pkg.Context(0)
    rng = np.random.default_rng(3); n, k = 131072, 3072
    w = pkg.Q4Tensor.from_q4_bytes(pkg.synth.synth_q4_blocks(rng, n * k, 0.02), [n, k], ctx)
    x = rng.standard_normal((1, 1, k)).astype(np.float32); y = rng.standard_normal((1, 1, k)).astype(np.float32)
    wx = pkg.q4_matmul(x, w); wy = pkg.q4_matmul(y, w); wz = pkg.q4_matmul((2 * x - 3 * y).astype(np.float32), w)
    assert np.abs(wz - (2 * wx - 3 * wy)).max() < 1e-4 * np.abs(wz).max()
    rows = rng.standard_normal((1, 40, k)).astype(np.float32)
    big = pkg.q4_matmul(rows, w)                                   # MFMA GEMM
    mid = pkg.q4_matmul(rows[:, :16], w)                           # skinny
    one = np.concatenate([pkg.q4_matmul(rows[:, i:i + 1], w) for i in (0, 7, 15)], axis=1)   # GEMV
    assert np.abs(big[:, :16] - mid).max() < 1e-4 * np.abs(big).max()
    assert np.abs(mid[:, [0, 7, 15]] - one).max() < 1e-4 * np.abs(big).max()
    w.close(); ctx.close()


def test_full_16s_clip_vs_oracle_golden(pkg, full):
    """THE published-metric workload end to end at full size against the CPU oracle: ids and per-step top logits of the 16 s clip
    (oracle run once on the GPU box's host CPU by tests/golden/make_fullsize_golden.py, 67 s on 128 threads; result committed)."""
    import hashlib
    m, _, ctx = full
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_oracle.npz"))
    x = pkg.synth.synth_audio(16.0, seed=1234)
    assert hashlib.sha256(x.tobytes()).digest() == g["audio_sha256"].tobytes()                 # same clip
    hs = hashlib.sha256()
    with open(os.path.join(cache_dir(), "full_q4_seed42.gguf"), "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            hs.update(chunk)
    assert hs.digest() == g["gguf_sha256"].tobytes()                                           # same weights
    t = pkg.TimeEmbedding(3072).embed(6.0)
    mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
    assert mel.shape[0] == int(g["mel_frames"])
    ids, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)
    rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
    assert len(ids) == len(rids) == 108
    # first disagreement (if any) must be at a near-tie of the oracle; up to there the top logit matches within the stated tolerance
    agree = ids == rids
    stop = len(ids) if agree.all() else int(np.argmin(agree))
    if stop < len(ids):
        assert top1[stop] - top2[stop] <= 10 * TOL * max(1.0, amax), f"ids differ at step {stop} with a clear margin"
    near = (top1 - top2) <= 10 * TOL * max(1.0, amax); first_tie = int(np.argmax(near)) if near.any() else len(rids)
    assert stop >= first_tie, f"ids agree for {stop} steps only; the oracle's first near-tie is at step {first_tie}"      # (measured: all 108)
    assert np.abs(lg[:stop].max(axis=1) - top1[:stop]).max() <= TOL * max(1.0, amax)
    assert np.abs(lg[0, :4096] - g["logits_step0"]).max() <= TOL * max(1.0, amax)
    # the product path (device mel + graph replay) and the batch path give the same ids up to that point
    ids_a = m.transcribe_audio(x, t)
    ids_b = m.transcribe_batch([x, x], t)
    assert (ids_a[:stop] == rids[:stop]).all() and (ids_b[0][:stop] == rids[:stop]).all() and (ids_b[1] == ids_b[0]).all()
    print(f"full-size golden: ids agree for {stop}/108 steps; min oracle margin {float((top1 - top2).min()):.4g}")


def test_full_batch16_vs_oracle_golden(pkg, full):
    """BASELINE configs[3] (16 x 16 s, one XF group of 16 rows) against the CPU oracle's golden for the 16 s clip: 16 rows of the golden
    clip, every row must reproduce the oracle ids up to the oracle's first near-tie; all rows identical to each other."""
    m, _, _ = full
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_oracle.npz"))
    x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
    rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
    safe = (top1 - top2) > 10 * TOL * max(1.0, amax)
    stop = len(rids) if safe.all() else int(np.argmin(safe))           # ids are pinned up to (not including) the first near-tie step
    outs = m.transcribe_batch([x] * 16, t)
    assert len(outs) == 16 and all(len(o) == 108 for o in outs)
    for r, o in enumerate(outs):
        assert (o[:stop] == rids[:stop]).all(), f"row {r} differs from the oracle before its first near-tie (step {stop})"
        assert (o == outs[0]).all(), f"row {r} differs from row 0"
    tm = m.timings(); assert tm["decode_tokens"] == 16 * 108
    print(f"batch-16 golden: all 16 rows agree with the oracle for {stop}/108 steps ({int((outs[0] == rids).sum())} ids equal)")


def test_full_peaked_golden_all_ids_single_batch16_ragged(pkg):
    """A full-size Q4 golden with COMFORTABLE margins (tests/golden/make_fullsize_peaked_golden.py: peaked logit distribution, |logit| up to 64, smallest top-2 margin of the
    oracle's 108 steps 1.28 = 100 x the stated tolerance): every path must reproduce ALL 108 oracle ids -- single stream on the decode engine (eager with logits, graph
    replay) and on the per-operator launches, 16 rows of the clip through vox_transcribe_batch on the batched decode-layer engine and on the launch-based step, and a ragged
    22-row batch with the clip in the caller's last slot.  (The 16 s golden of the bench model has its first near-tie at step 14: tests on it can only assert 14 ids.)"""
    import hashlib
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_peaked_oracle.npz"))
    path = os.path.join(cache_dir(), "full_q4_peaked_seed44.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=44, peaked=True); os.replace(path + ".tmp", path)
    hs = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            hs.update(chunk)
    assert hs.digest() == g["gguf_sha256"].tobytes()
    x = pkg.synth.synth_audio(16.0, seed=7049)
    assert hashlib.sha256(x.tobytes()).digest() == g["audio_sha256"].tobytes()
    rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
    assert len(rids) == 108 and float((top1 - top2).min()) > 50 * TOL * amax          # the fixture's point: no near-tie anywhere
    ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    try:
        t = pkg.TimeEmbedding(3072).embed(6.0)
        mel = np.ascontiguousarray(pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]
        for engine in (True, False):
            if m.set_decode_engine(engine) != engine:
                continue
            ids, lg = m.transcribe_streaming(mel, t, return_logits=True)
            assert np.array_equal(ids, rids), f"single stream (engine={engine}): ids differ from the oracle"
            err = float(np.abs(np.sort(lg, axis=1)[:, -1] - top1).max())
            assert err <= 1e-2 * amax, (engine, err)
            assert np.array_equal(m.transcribe_streaming(mel, t), rids)              # graph-replayed decode loop
            assert np.array_equal(m.transcribe_audio(x, t), rids)                    # device front-end
            print(f"peaked golden, single stream (engine={engine}): 108 / 108 ids, max top-logit error {err:.3e} at |logit| max {amax:.1f}")
        m.set_decode_engine(True)
        for batch_engine in (True, False):
            active, n0 = m.set_batch_engine(batch_engine)
            if active != batch_engine:
                continue
            outs = m.transcribe_batch([x] * 16, t)
            assert len(outs) == 16 and all(np.array_equal(o, rids) for o in outs), f"batch of 16 (engine={batch_engine}): a row differs from the oracle"
            _, n1 = m.set_batch_engine()
            assert (n1 - n0 == 107) if batch_engine else (n1 == n0)                  # one engine launch per decode step, none on the launch-based step
            short = [pkg.synth.synth_audio(3.0 + 0.37 * (i % 17), seed=600 + i) for i in range(21)]
            ragged = m.transcribe_batch(short + [x], t)                              # 22 rows, two groups: the launch-based step (the engine serves one-group batches)
            assert np.array_equal(ragged[-1], rids)
            five = m.transcribe_batch(short[:4] + [x], t)                            # a ragged ONE-group batch: idle rows + rows that finish early
            assert np.array_equal(five[-1], rids) and all(np.array_equal(a, b) for a, b in zip(five[:4], ragged[:4]))
            print(f"peaked golden, batch (engine={batch_engine}): 16 / 16 rows, ragged 22-row and 5-row batches: all 108 ids")
        m.set_batch_engine(True)
    finally:
        m.close(); ctx.close()


def test_full_wide_batches_peaked_golden_64_lockstep_and_81_continuous(pkg, monkeypatch):
    """VERDICT r4 items 2 / 3(a), full size, on the peaked golden (no near-tie anywhere: ALL 108 ids are asserted):
    (a) a 64-row batch -- the width BASELINE configs[4] ran at in round 4 -- with the golden clip in the caller's slots 0, 31 and 63 among 3 .. 9 s clips, on the four forked
        lock-step launch chains (VOX_BATCH_NO_CONTINUOUS=1: the widest batch asserted at full size before was 22 rows) and on the default path (continuous batching);
    (b) an 81-row batch (a rank's share of the 647-clip corpus at 8 GPUs) in ONE call: continuous batching over slots, two encoder / prefill chunks, refilled slots --
        golden clip in slots 0, 40 and 80; every other row equal to what the 64-row lock-step batch gave for the same clip (rows are independent of slot, group and batch)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_peaked_oracle.npz"))
    path = os.path.join(cache_dir(), "full_q4_peaked_seed44.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=44, peaked=True); os.replace(path + ".tmp", path)
    x = pkg.synth.synth_audio(16.0, seed=7049); rids = g["ids"]
    ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    try:
        t = pkg.TimeEmbedding(3072).embed(6.0)
        short = [pkg.synth.synth_audio(3.0 + 0.37 * (i % 17), seed=700 + i) for i in range(78)]
        b64 = list(short[:61]); b64.insert(0, x); b64.insert(31, x); b64.append(x); assert len(b64) == 64 and b64[31] is x
        monkeypatch.setenv("VOX_BATCH_NO_CONTINUOUS", "1")
        lock = m.transcribe_batch(b64, t)
        monkeypatch.delenv("VOX_BATCH_NO_CONTINUOUS")
        for sl in (0, 31, 63):
            assert np.array_equal(lock[sl], rids), f"64-row lock-step batch: slot {sl} differs from the oracle"
        cont = m.transcribe_batch(b64, t)
        assert all(len(a) == len(b) and (a == b).all() for a, b in zip(cont, lock)), "64 rows: continuous batching and the lock-step batch disagree"
        b81 = list(short); b81.insert(0, x); b81.insert(40, x); b81.append(x); assert len(b81) == 81 and b81[40] is x
        o81 = m.transcribe_batch(b81, t)
        tm = m.timings(); assert tm["decode_tokens"] == sum(len(o) for o in o81)
        for sl in (0, 40, 80):
            assert np.array_equal(o81[sl], rids), f"81-row continuous batch: slot {sl} differs from the oracle"
        ref_of = {id(c): o for c, o in zip(b64, lock)}
        n_cmp = 0
        for c, o in zip(b81, o81):
            if id(c) in ref_of:
                assert np.array_equal(o, ref_of[id(c)]); n_cmp += 1
        assert n_cmp >= 64
        assert all((a == b).all() for a, b in zip(o81, m.transcribe_batch(b81, t)))      # deterministic
        # round 6: the WIDE step (2..4 active slot groups as ONE GEMM per operator, launch_q4_wide) on forced 3 and 4 slot groups: same ids, golden clip ALL 108
        for G, wmin in ((4, 2), (3, 3), (4, 4), (5, 4), (6, 4)):      # (5, 6 groups: 80 / 96 slots, two wide chains per step)
            monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", str(G)); monkeypatch.setenv("VOX_BATCH_WIDE_MIN", str(wmin))
            w81 = m.transcribe_batch(b81, t)
            for sl in (0, 40, 80):
                assert np.array_equal(w81[sl], rids), f"wide step ({G} groups, from {wmin}): slot {sl} differs from the oracle"
            assert all(len(a) == len(b) and (a == b).all() for a, b in zip(w81, o81)), f"wide step ({G} groups, from {wmin}) and the default path disagree"
        monkeypatch.delenv("VOX_BATCH_SLOT_GROUPS"); monkeypatch.delenv("VOX_BATCH_WIDE_MIN")
        print(f"peaked golden: 64-row lock-step and continuous batches, 81-row continuous batch -- golden clip ALL 108 ids in every placed slot, {n_cmp} rows identical across the batches")
    finally:
        m.close(); ctx.close()


def test_full_chunked_files_are_units_of_the_wide_batch(pkg):
    """VERDICT r5 item 1 at FULL size, on the peaked checkpoint: the reference's CLI semantics (bin/transcribe.rs:207-265, default --max-mel-frames 1200) -- the FILE is
    peak-normalised once, split at 192 000 samples, every chunk an independent unit.  A 30 s, a 25 s and the 16 s golden-clip file (-> 3 + 3 + 2 chunks) among 12 short
    files = 20 units in ONE vox_transcribe_batch_ex call (continuous batching; group peaks reduced on the device from the raw samples): every unit's ids must be those of
    the serial CLI path on the same chunk -- pad -> log-mel -> transcribe_streaming of the host-normalised file's slice (cli.transcribe_one) -- up to a near-tie of the
    serial path's own logits, and identical to the call made on host-normalised slices (norm_group -1, what cli.py --batch passes)."""
    path = os.path.join(cache_dir(), "full_q4_peaked_seed44.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=44, peaked=True); os.replace(path + ".tmp", path)
    S = pkg.synth; t = pkg.TimeEmbedding(3072).embed(6.0)
    files = [0.4 * S.synth_audio(30.0, seed=9101), 0.7 * S.synth_audio(25.0, seed=9102), 0.25 * S.synth_audio(16.0, seed=7049)]
    files[0][:192000] *= np.float32(1e-3)      # a first chunk 60 dB below its file's peak: under the FILE's scale most of its log-mel sits at the floor
    files += [(0.1 + 0.05 * i) * S.synth_audio(3.0 + 0.7 * i, seed=9200 + i) for i in range(12)]
    cfg = pkg.ChunkConfig.voxtral().with_max_frames(1200)
    raw, nrm, grp = [], [], []
    for fi, x in enumerate(files):
        xn = pkg.peak_normalize(x, 0.95)
        plan = pkg.chunk_plan(x.size, cfg) if pkg.needs_chunking(x.size, cfg) else [(0, x.size)]
        for a, b in plan:
            raw.append(x[a:b]); nrm.append(xn[a:b]); grp.append(fi)
    assert len(raw) == 20 and [c.size for c in raw[:8]] == [192000, 192000, 96000, 192000, 192000, 16000, 192000, 64000]      # chunk A / chunk B of a 16 s file
    ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    try:
        outs = m.transcribe_batch(raw, t, norm_group=grp)
        assert [len(o) for o in outs[6:8]] == [83, 33]                                        # CLI default chunk A / chunk B emit 83 / 33 ids
        as_is = m.transcribe_batch(nrm, t, norm_group=[-1] * len(nrm))
        assert all(len(a) == len(b) and (a == b).all() for a, b in zip(outs, as_is)), "device-side file peaks != host-normalised files"
        own = m.transcribe_batch(raw, t)                                                        # every chunk normalised by ITS OWN peak: not the reference's arithmetic
        assert (own[0] != outs[0]).any(), "the quiet first chunk of file 0 must decode differently under its own peak than under its file's"
        assert all((a == b).all() for a, b in zip(own[8:], outs[8:]))                           # un-chunked files: the file IS the unit
        mel = pkg.MelSpectrogram.voxtral(ctx); n_same = 0
        for u, c in enumerate(nrm):
            rids, rlg = m.transcribe_streaming(np.ascontiguousarray(mel.compute_log(pkg.pad_audio(c)).T)[None], t, return_logits=True)
            n_same += int(check_greedy_ids(outs[u], rids, rlg, TOL) == len(rids))
        assert n_same >= 18, n_same
        print(f"chunks as units at full size: 20 units of 15 files, {n_same}/20 identical to the serial CLI path end to end (the rest up to a near-tie)")
    finally:
        m.close(); ctx.close()


def test_full_continuous_batch_engine_forms_two_groups_per_launch(pkg, monkeypatch, capfd):
    """The steps of a wide batch with one or two active slot groups go through the batched decode-layer engine: TWO groups per launch (decode_engine_b16_kernel<2>: group B's
    phase runs while group A's hand-off resolves; cache slices per slot through EngBParams::kv_row), one group per launch once the second group has retired.  Full size,
    peaked checkpoint (no near-tie: every id is asserted): (a) a ragged 40-row batch -- 32 slots, refilled -- on the engine forms == on the forked launch chains
    (VOX_BATCH_CONT_NO_ENGINE=1) == the golden ids for the golden clip; the engine's launch counter moves by one launch per decode step; (b) a second model whose two-group
    launch loses a publish (fault-injection flag 16384): the launch ends on its bounded wait, the call says so on stderr and serves the session again on the launch
    chains -- same ids -- and the model is re-armed."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_peaked_oracle.npz"))
    path = os.path.join(cache_dir(), "full_q4_peaked_seed44.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=44, peaked=True); os.replace(path + ".tmp", path)
    x = pkg.synth.synth_audio(16.0, seed=7049); rids = g["ids"]
    t = pkg.TimeEmbedding(3072).embed(6.0)
    clips = [pkg.synth.synth_audio(2.5 + 0.41 * (i % 19), seed=1700 + i) for i in range(39)]; clips.insert(17, x)
    monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", "2")      # (the planner alone would serve 40 short rows from ONE group's 16 slots: the one-group launch at 1.17 ms per step)
    ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx)
    try:
        active, n0 = m.set_batch_engine(True)
        if not active:
            pytest.skip("batched decode engine not available on this device (needs 256 CUs)")
        eng = m.transcribe_batch(clips, t)
        _, n1 = m.set_batch_engine()
        steps = m.timings()["graph_replays"]
        assert n1 - n0 >= steps > 50, f"{n1 - n0} engine launches for {steps} replayed steps"      # every step of this 2-group session is ONE engine launch (+ the eager first step of a process)
        monkeypatch.setenv("VOX_BATCH_CONT_NO_ENGINE", "1")
        ref = m.transcribe_batch(clips, t)
        _, n2 = m.set_batch_engine()
        monkeypatch.delenv("VOX_BATCH_CONT_NO_ENGINE")
        assert n2 == n1                                                      # none on the launch chains
        assert len(eng) == 40 and all(len(a) == len(b) and (a == b).all() for a, b in zip(eng, ref)), "engine forms and launch chains disagree"
        assert np.array_equal(eng[17], rids), "the golden clip (slot 17) differs from the oracle"
        assert all((a == b).all() for a, b in zip(eng, m.transcribe_batch(clips, t)))      # run to run
        capfd.readouterr()
        monkeypatch.setenv("VOX_BATCH_ENGINE_FLAGS2", str(128 | 1024 | 16384))
        b = pkg.Q4ModelLoader.from_file(path).load(ctx)
        monkeypatch.delenv("VOX_BATCH_ENGINE_FLAGS2")
        try:
            import time
            t0 = time.time()
            out = b.transcribe_batch(clips, t)
            assert time.time() - t0 < 60.0                                   # bounded waits, not a hang
            err = capfd.readouterr().err
            assert "continuous batch" in err and "hand-off timeout" in err and "strike 1 of 3" in err
            assert all((a == r).all() for a, r in zip(out, ref)), "the re-run on the launch chains gave other ids"
            assert b.set_batch_engine()[0]                                   # re-armed
        finally:
            b.close()
        print(f"continuous batch, 40 rows: engine forms (two groups per launch, {n1 - n0} launches) == launch chains == golden; a lost publish is survived")
    finally:
        m.close(); ctx.close()


def test_full_ragged_batch_groups_retire(pkg, full):
    """A ragged batch wider than one 16-row group at FULL size: vox_transcribe_batch runs the rows longest first and RETIRES a group's layer chain
    once its longest member is done.  The 16 s golden clip sits in the caller's LAST slot between 3..9 s clips: it must still reproduce the oracle's
    ids in that slot, the 16-row batch of the same clip, and the run with retirement / sorting switched off (rows are independent of their group)."""
    m, _, _ = full
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_oracle.npz"))
    t = pkg.TimeEmbedding(3072).embed(6.0)
    rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
    safe = (top1 - top2) > 10 * TOL * max(1.0, amax)
    stop = len(rids) if safe.all() else int(np.argmin(safe))
    secs = [3.0 + 0.37 * (i % 17) for i in range(21)]                     # 3 .. 8.9 s, not sorted
    clips = [pkg.synth.synth_audio(s, seed=500 + i) for i, s in enumerate(secs)] + [pkg.synth.synth_audio(16.0, seed=1234)]
    outs = m.transcribe_batch(clips, t)                                   # 22 rows: groups of 16 + 6; the short group retires after ~40 steps
    assert len(outs) == 22 and len(outs[-1]) == 108
    assert (outs[-1][:stop] == rids[:stop]).all(), "the golden clip (caller's last slot) differs from the oracle before its first near-tie"
    os.environ["VOX_BATCH_NO_RETIRE"] = "1"; os.environ["VOX_BATCH_NO_SORT"] = "1"
    try:
        ref = m.transcribe_batch(clips, t)
    finally:
        del os.environ["VOX_BATCH_NO_RETIRE"]; del os.environ["VOX_BATCH_NO_SORT"]
    for r, (a, b) in enumerate(zip(outs, ref)):
        assert len(a) == len(b) and (a == b).all(), f"slot {r}: retiring / sorting changed the ids"
    print(f"ragged full-size batch: 22 rows identical with and without group retirement; golden clip agrees with the oracle for {stop}/108 steps")


def test_full_16s_clip_f32_vs_oracle_golden(pkg):
    """BASELINE configs[0-1] at FULL size: the f32 SafeTensors path (VoxtralModelLoader -> transcribe_f32_with_model, bin/transcribe.rs:362-438;
    dense bf16 weights exact on device, f32 activations / accumulation) on the 16 s bench clip against the CPU oracle's golden
    (tests/golden/make_fullsize_f32_golden.py, run once on the GPU box's host).  north_star: "greedy token IDs bit-exact on the f32 path" --
    the count of identical ids is printed; a disagreement is only admissible at a near-tie of the oracle (top-2 margin < 10 x tolerance)."""
    import hashlib
    gp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_16s_f32_oracle.npz")
    if not os.path.exists(gp):
        pytest.skip("full-size f32 golden not generated yet (tests/golden/make_fullsize_f32_golden.py)")
    g = np.load(gp)
    st = full_dense_safetensors(int(g["seed"]))
    assert os.path.getsize(st) == int(g["st_size"]) and dense_head_sha(st) == g["st_head_sha256"].tobytes()      # same weights
    x = pkg.synth.synth_audio(16.0, seed=1234)
    assert hashlib.sha256(x.tobytes()).digest() == g["audio_sha256"].tobytes()                                    # same clip
    ctx = pkg.Context(0)
    m = pkg.VoxtralModelLoader.from_file(st).load(ctx)
    c = m.config
    assert (c.enc_layers, c.enc_dim, c.dec_layers, c.dec_dim, c.dec_heads, c.dec_kv_heads, c.dec_ffn, c.vocab) == (32, 1280, 26, 3072, 32, 8, 9216, 131072)
    t = pkg.TimeEmbedding(3072).embed(6.0)
    mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
    assert mel.shape[0] == int(g["mel_frames"])
    ids, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)
    rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
    assert len(ids) == len(rids) == 108
    agree = ids == rids
    stop = len(ids) if agree.all() else int(np.argmin(agree))
    if stop < len(ids):
        assert top1[stop] - top2[stop] <= 10 * TOL * max(1.0, amax), f"f32 ids differ at step {stop} with a clear margin {top1[stop] - top2[stop]}"
    near = (top1 - top2) <= 10 * TOL * max(1.0, amax); first_tie = int(np.argmax(near)) if near.any() else len(rids)
    assert stop >= first_tie, f"f32 ids agree for {stop} steps only; the oracle's first near-tie is at step {first_tie}"      # (measured: all 108)
    assert np.abs(lg[:stop].max(axis=1) - top1[:stop]).max() <= TOL * max(1.0, amax)
    assert np.abs(lg[0, :4096] - g["logits_step0"]).max() <= TOL * max(1.0, amax)
    ids_a = m.transcribe_audio(x, t)                                          # product path: device mel + graph replay
    assert (ids_a[:stop] == rids[:stop]).all() and (m.transcribe_audio(x, t) == ids_a).all()
    print(f"full-size f32 golden: {int(agree.sum())}/108 ids identical to the CPU oracle (first difference: {'none' if stop == 108 else stop}; "
          f"min oracle top-2 margin {float((top1 - top2).min()):.4g})")
    m.close(); ctx.close()


def test_full_30s_f32_heavytail_vs_oracle_golden(pkg):
    """The f32 SafeTensors path under the STRESS statistics (tests/golden/make_fullsize_f32_heavytail_golden.py: power-of-two Student-t(4) block scales, six
    outlier channels x 64 in the decoder's residual stream, final norm centred on 5 => |logit| in the hundreds) on the 30 s clip (234 decoder positions): dense
    bf16 weights exact on device, activations through the hi/lo-bf16 MFMA GEMMs (encoder, prefill) and the f32 GEMV chain (decode).  Ids identical to the CPU
    oracle up to a near-tie; top logits within 2e-2 of the largest |logit| end to end (the same conditioning statement as the Q4 stress golden)."""
    import hashlib
    gp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_30s_f32_heavytail_oracle.npz")
    if not os.path.exists(gp):
        pytest.skip("heavy-tailed f32 golden not generated yet (tests/golden/make_fullsize_f32_heavytail_golden.py)")
    g = np.load(gp)
    st = full_dense_safetensors(int(g["seed"]), heavy_tail=True)
    assert os.path.getsize(st) == int(g["st_size"]) and dense_head_sha(st) == g["st_head_sha256"].tobytes()      # same weights
    x = pkg.synth.synth_audio(float(g["seconds"]), seed=4321)
    assert hashlib.sha256(x.tobytes()).digest() == g["audio_sha256"].tobytes()                                    # same clip
    ctx = pkg.Context(0)
    m = pkg.VoxtralModelLoader.from_file(st).load(ctx)
    try:
        t = pkg.TimeEmbedding(3072).embed(6.0)
        mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
        assert mel.shape[0] == int(g["mel_frames"])
        ids, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)
        rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
        assert len(ids) == len(rids) > 100 and amax > 15.0
        agree = ids == rids
        stop = len(ids) if agree.all() else int(np.argmin(agree))
        if stop < len(ids):
            assert top1[stop] - top2[stop] <= 2e-2 * amax, f"f32 heavy-tail ids differ at step {stop} with a clear margin {top1[stop] - top2[stop]}"
        near = (top1 - top2) <= 2e-2 * amax; first_tie = int(np.argmax(near)) if near.any() else len(rids)
        assert stop >= first_tie, f"f32 heavy-tail ids agree for {stop} steps only; the oracle's first near-tie is at step {first_tie}"      # (measured: all 196)
        err = float(np.abs(lg[:stop].max(axis=1) - top1[:stop]).max())
        assert err <= 1e-2 * amax, (err, amax)
        ids_a = m.transcribe_audio(x, t)                                          # product path: device mel + graph replay
        assert (ids_a[:stop] == rids[:stop]).all() and (m.transcribe_audio(x, t) == ids_a).all()
        print(f"heavy-tail f32 golden: ids agree for {stop}/{len(ids)} steps; max top-logit error {err:.3e} at |logit| max {amax:.1f}")
    finally:
        m.close(); ctx.close()


def test_full_attention_wo_launch(pkg, full, monkeypatch):
    """Full size, 16 s clip: the default decode layer runs attention + wo as ONE launch whose 32-way K split is combined with int64 fixed-point
    atomics (attn_wo_kernel).  (a) order independence: two eager runs give bit-identical logits for all 108 steps although the 256 workgroups'
    atomics interleave differently; (b) against the separate attention and wo launches (VOX_NO_ATTN_WO=1): the same ids, logits equal to
    summation-order noise (the stated bound: 2e-4 of the largest |logit|); (c) the replayed graph emits the same ids."""
    m, _, ctx = full
    m.set_decode_engine(False)      # these tests compare variants of the per-operator decode path
    x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
    mel = np.ascontiguousarray(pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]
    ids_a, lg_a = m.transcribe_streaming(mel, t, return_logits=True)
    ids_b, lg_b = m.transcribe_streaming(mel, t, return_logits=True)
    assert np.array_equal(ids_a, ids_b) and np.array_equal(lg_a, lg_b)
    monkeypatch.setenv("VOX_NO_ATTN_WO", "1")
    ids_s, lg_s = m.transcribe_streaming(mel, t, return_logits=True)
    monkeypatch.delenv("VOX_NO_ATTN_WO")
    err = float(np.max(np.abs(lg_a - lg_s))); top = float(np.max(np.abs(lg_s)))
    print(f"attention+wo launch vs separate launches: max |dlogit| {err:.3e} (largest |logit| {top:.2f}), ids equal: {np.array_equal(ids_a, ids_s)}")
    assert np.array_equal(ids_a, ids_s) and err <= 2e-4 * top
    assert np.array_equal(m.transcribe_streaming(mel, t), ids_a)          # graph replay


def test_full_attention_wo_launch_long_context(pkg, full, monkeypatch):
    """30 s clip (234 decoder positions): the attention + wo launch beyond the 160 keys it requests up front (the looped K / V passes of
    attn_decode_core) against the separate launches -- same ids, logits equal to summation-order noise."""
    m, _, ctx = full
    m.set_decode_engine(False)      # these tests compare variants of the per-operator decode path
    x = pkg.synth.synth_audio(30.0, seed=4321); t = pkg.TimeEmbedding(3072).embed(6.0)
    mel = np.ascontiguousarray(pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]
    ids_a, lg_a = m.transcribe_streaming(mel, t, return_logits=True)
    monkeypatch.setenv("VOX_NO_ATTN_WO", "1")
    ids_s, lg_s = m.transcribe_streaming(mel, t, return_logits=True)
    monkeypatch.delenv("VOX_NO_ATTN_WO")
    assert len(ids_a) > 180                     # 38 + 196 positions > the 160 keys requested up front
    err = float(np.max(np.abs(lg_a - lg_s))); top = float(np.max(np.abs(lg_s)))
    srt = np.sort(lg_s, axis=1); safe = (srt[:, -1] - srt[:, -2]) > 10 * TOL * max(1.0, top)
    stop = len(safe) if safe.all() else int(np.argmin(safe))
    print(f"30 s clip, {len(ids_a)} ids: attention+wo launch vs separate launches max |dlogit| {err:.3e} (largest |logit| {top:.2f}); ids equal up to step {stop}")
    assert (ids_a[:stop] == ids_s[:stop]).all() and (stop < len(ids_a) or err <= 2e-4 * top)


def _mel_of(pkg, ctx, seconds, seed):
    x = pkg.synth.synth_audio(seconds, seed=seed)
    return np.ascontiguousarray(pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x))).T)[None]


@pytest.mark.parametrize("seconds,seed", [(16.0, 1234), (30.0, 4321)])
def test_full_decode_engine_vs_per_operator_path(pkg, full, seconds, seed):
    """The persistent decode-step engine (ONE launch per token: vox_engine.hip) against the per-operator launches it replaces, full size: same ids, logits equal to
    summation-order noise (stated bound 2e-4 of the largest |logit|) on every step; the engine is run-to-run bit-identical (every cross-CU sum has a fixed order);
    its graph-replayed ids equal its eager ids.  30 s clip: 234 decoder positions, past the 144 keys the engine's attention requests up front."""
    m, _, ctx = full
    if not m.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    t = pkg.TimeEmbedding(3072).embed(6.0); mel = _mel_of(pkg, ctx, seconds, seed)
    ids_e, lg_e = m.transcribe_streaming(mel, t, return_logits=True)
    ids_e2, lg_e2 = m.transcribe_streaming(mel, t, return_logits=True)
    assert np.array_equal(ids_e, ids_e2) and np.array_equal(lg_e, lg_e2)
    ids_g = m.transcribe_streaming(mel, t)                                  # graph replay
    assert not m.set_decode_engine(False)
    ids_o, lg_o = m.transcribe_streaming(mel, t, return_logits=True)
    ids_og = m.transcribe_streaming(mel, t)
    err = float(np.max(np.abs(lg_e - lg_o))); top = float(np.max(np.abs(lg_o)))
    print(f"decode engine vs per-operator path ({seconds:.0f} s, {len(ids_e)} ids): max |dlogit| {err:.3e} (largest |logit| {top:.2f}), ids equal: {np.array_equal(ids_e, ids_o)}")
    assert len(ids_e) == len(ids_o) and (seconds < 20 or len(ids_e) > 180)
    assert np.array_equal(ids_e, ids_o) and err <= 2e-4 * top
    assert np.array_equal(ids_g, ids_e) and np.array_equal(ids_og, ids_o)


def test_full_decode_engine_long_positions_75s_unchunked(pkg, full):
    """the reference's e2e-bench never chunks (bin/e2e_bench.rs:98-135) and the decoder window is 8 192, so a 75 s clip runs the engine to
    position ~ 515 -- the third and later 192-key attention rounds (positions 384 ... 1024) that no test reached before (the longest was 234).  Engine == per-operator
    launches: ids up to the first near-tie of the per-operator logits, logits <= 2e-4 of the largest on every step up to there; engine run-to-run bit-identical and
    graph replay == eager."""
    m, _, ctx = full
    if not m.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    t = pkg.TimeEmbedding(3072).embed(6.0); mel = _mel_of(pkg, ctx, 75.0, 606)
    ids_e, lg_e = m.transcribe_streaming(mel, t, return_logits=True)
    ids_e2 = m.transcribe_streaming(mel, t, return_logits=True)[0]
    ids_g = m.transcribe_streaming(mel, t)
    assert not m.set_decode_engine(False)
    ids_o, lg_o = m.transcribe_streaming(mel, t, return_logits=True)
    assert len(ids_e) == len(ids_o) and len(ids_e) > 450 and len(ids_e) + 38 <= 1024      # ~ 515 decoder positions, inside the engine's 1024-row cache limit
    top = float(np.abs(lg_o).max())
    first = check_greedy_ids(ids_e, ids_o, lg_o, TOL)
    err = float(np.abs(lg_e[:first + 1] - lg_o[:first + 1]).max())
    print(f"decode engine vs per-operator path, 75 s un-chunked ({len(ids_e)} ids, positions up to {len(ids_e) + 37}): ids agree for {first}/{len(ids_e)} steps, max |dlogit| {err:.3e} of {top:.2f}")
    assert err <= TOL * top and first + 38 >= 420      # well into the third 192-key round (positions >= 384) at the very least
    assert np.array_equal(ids_e, ids_e2) and np.array_equal(ids_g, ids_e)


@pytest.mark.parametrize("engine", [True, False])
def test_full_single_decode_launch_at_position_700_vs_oracle(pkg, orc, full, engine):
    """ONE decode step at position 700 against the oracle on every logit, with a SYNTHETIC cache -- the same N(0, 1) K rows (RoPE is already applied
    to what a cache holds) and V rows written into the oracle's and the library's cache through KVCache::update (vox_cache_update / orc_cache_update), so no 700-row CPU
    prefill is needed.  The engine's attention walks four 192-key rounds here; bound 2e-4 of the largest, as at position 38."""
    m, o, _ = full
    if m.set_decode_engine(engine) != engine:
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    P, KV, HD, D, L = 700, 8, 128, 3072, 26
    t = pkg.TimeEmbedding(D).embed(6.0); dec = m.decoder()
    oc = o.cache(768); c = dec.create_cache_preallocated(768)
    try:
        for l in range(L):
            rng = np.random.default_rng([2026, l])
            k = rng.standard_normal((KV, P, HD)).astype(np.float32); v = rng.standard_normal((KV, P, HD)).astype(np.float32)
            o.cache_update(oc, l, 0, k, v); c.update(l, 0, k, v)
        assert c.seq_len() == P and orc.lib().orc_cache_len(oc) == P
        worst_h = worst_l = 0.0
        for step in range(2):      # positions 700 and 701 (the second step attends to the first one's freshly written row)
            xs = (0.3 * np.random.default_rng([77, step]).standard_normal((1, D))).astype(np.float32)
            rh = o.forward_hidden_with_cache(xs, t, oc); gh = dec.forward_hidden_with_cache(xs[None], t, c)[0]
            rl = o.lm_head(rh); gl = dec.lm_head(gh[None])[0]
            worst_h = max(worst_h, rel_err(gh, rh)); worst_l = max(worst_l, rel_err(gl, rl))
            assert int(gl.argmax()) == int(rl.argmax())
        assert c.seq_len() == P + 2
        c.truncate(P); assert c.seq_len() == P      # vox_cache_truncate: the next forward appends at 700 again ...
        xs = (0.3 * np.random.default_rng([77, 0]).standard_normal((1, D))).astype(np.float32)
        gh2 = dec.forward_hidden_with_cache(xs[None], t, c)[0]
        o2 = o.cache(768)
        try:
            for l in range(L):
                rng = np.random.default_rng([2026, l])
                k = rng.standard_normal((KV, P, HD)).astype(np.float32); v = rng.standard_normal((KV, P, HD)).astype(np.float32)
                o.cache_update(o2, l, 0, k, v)
            assert rel_err(gh2, o.forward_hidden_with_cache(xs, t, o2)) < TOL      # ... and reproduces the first step
        finally:
            o.cache_free(o2)
    finally:
        o.cache_free(oc); c.close()
    print(f"one decode launch at position {P} vs oracle (engine {engine}): hidden {worst_h:.2e}, all 131072 logits {worst_l:.2e} of the largest")
    assert worst_h < TOL and worst_l < TOL


def _piecewise_loop(pkg, m, ctx, mel, t, fused):
    """bin/e2e_bench.rs:158-231 on device pointers through the C ABI's device-resident decoder surface."""
    D, V = m.config.dec_dim, m.config.vocab
    audio = m.encode_audio(mel)[0]; S = audio.shape[0]
    d_audio = ctx.upload(audio); d_text = ctx.alloc(38 * D * 4); d_in = ctx.alloc(38 * D * 4); d_log = ctx.alloc(38 * V * 4)
    dec = m.decoder(); cache = dec.create_cache_preallocated(S)
    try:
        dec.embed_tokens_from_ids_dev(np.array([1] + [32] * 37, np.int32), d_text)
        pkg.tensor_add_dev(ctx, d_audio, d_text, 38 * D, d_in)
        hid = dec.forward_hidden_with_cache_dev(d_in, 38, t, cache)
        if fused:
            tok = int(dec.lm_head_argmax(hid, 38)[-1])
        else:
            dec.lm_head_dev(hid, 38, d_log); tok = int(pkg.argmax_rows_dev(ctx, d_log + 37 * V * 4, 1, V)[0])
        gen = [tok]; last_logits = None
        for pos in range(39, S):
            dec.embed_tokens_from_ids_dev(np.array([gen[-1]], np.int32), d_text)
            pkg.tensor_add_dev(ctx, d_audio + (pos - 1) * D * 4, d_text, D, d_in)
            hid = dec.forward_hidden_with_cache_dev(d_in, 1, t, cache)
            if fused:
                gen.append(int(dec.lm_head_argmax(hid, 1)[0]))
            else:
                dec.lm_head_dev(hid, 1, d_log); gen.append(int(pkg.argmax_rows_dev(ctx, d_log, 1, V)[0]))
        if not fused:
            last_logits = ctx.download(d_log, (V,)); last_hidden = ctx.download(hid, (D,))
            return np.array(gen, np.int32), last_logits, last_hidden
        return np.array(gen, np.int32), None, None
    finally:
        cache.close()
        for p_ in (d_audio, d_text, d_in, d_log):
            ctx.free(p_)


@pytest.mark.parametrize("engine", [True, False])
def test_full_piecewise_decoder_surface_equals_transcribe_streaming(pkg, full, engine):
    """The reference's own decode loop (bin/e2e_bench.rs:179-224: embed_tokens_from_ids -> + audio row -> forward_hidden_with_cache -> lm_head -> argmax) on the
    device-resident C-ABI surface gives the ids of the fused vox_transcribe_streaming, with the decode engine (a single-row forward_hidden_with_cache = ONE engine launch
    that also produces the logits lm_head returns) and on the per-operator launches; vox_lm_head_argmax agrees with lm_head + argmax; the hidden row the engine hands
    out multiplies (for real, through the GEMM) to the logits it produced itself."""
    m, _, ctx = full
    if m.set_decode_engine(engine) != engine:
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    t = pkg.TimeEmbedding(3072).embed(6.0); mel = _mel_of(pkg, ctx, 16.0, 1234)
    ref = m.transcribe_streaming(mel, t)
    ids_a, lg, hid = _piecewise_loop(pkg, m, ctx, mel, t, fused=False)
    ids_b, _, _ = _piecewise_loop(pkg, m, ctx, mel, t, fused=True)
    assert len(ref) == 108 and np.array_equal(ids_a, ref) and np.array_equal(ids_b, ref)
    # the handed-out hidden row from ANOTHER buffer: multiplied for real by the Q4 operator -> the logits the engine launch produced itself
    d_h = ctx.upload(hid); d_l = ctx.alloc(131072 * 4)
    m.decoder().lm_head_dev(d_h, 1, d_l); real = ctx.download(d_l, (131072,)); ctx.free(d_h); ctx.free(d_l)
    top = float(np.abs(real).max()); err = float(np.abs(real - lg).max())
    print(f"piecewise loop (engine {engine}): 108 ids equal; engine-made logits vs lm_head GEMM of the handed-out hidden row: {err:.2e} (largest |logit| {top:.2f})")
    assert err <= 2e-4 * top and int(real.argmax()) == int(lg.argmax())


def test_full_e2e_piecewise_c_program(pkg, full, tmp_path):
    """tools/e2e_piecewise.c: the reference's e2e-bench loop, call for call, from plain C11 over include/voxtral_hip.h (its own process, its own model load): both variants'
    ids equal vox_transcribe_audio's; prints the tok/s a drop-in e2e-bench would report."""
    import json, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); pkg_dir = os.path.dirname(pkg.build.LIB_PATH)
    exe = str(tmp_path / "e2e_piecewise")
    r = subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "e2e_piecewise.c"), "-o", exe,
                        "-L" + pkg_dir, "-lvoxtral_hip", "-lm", "-Wl,-rpath," + pkg_dir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    x = pkg.synth.synth_audio(16.0, seed=1234); wav = str(tmp_path / "clip.f32"); x.astype(np.float32).tofile(wav)
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    r = subprocess.run([exe, path, wav, "2"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = res["lm_head+argmax"], res["lm_head_argmax"]
    print(f"e2e_piecewise (C, engine {res['decode_engine']}): call-for-call {a['tok_per_s']:.0f} tok/s (decode {a['decode_ms']:.1f} ms, encode {a['encode_ms']:.1f} ms), "
          f"with vox_lm_head_argmax {b['tok_per_s']:.0f} tok/s")
    assert a["ids_equal_transcribe_audio"] and b["ids_equal_transcribe_audio"] and a["decode_tokens"] == 108
    m, _, ctx = full
    assert np.array_equal(np.array(res["ids"], np.int32), m.transcribe_audio(x, pkg.TimeEmbedding(3072).embed(6.0)))


@pytest.mark.parametrize("engine", [True, False])
def test_full_single_decode_launch_vs_oracle_all_logits(pkg, orc, full, engine):
    """The oracle's own decoder state -> ONE decode step of the HIP path -> the hidden row and ALL 131 072 logits against the oracle (<= 2e-4 of the largest), for the
    persistent engine (one launch = 26 layers + final norm + lm_head) and for the per-operator launches: prefill on the oracle's audio embeddings, then three steps fed
    with the ORACLE's token and audio rows (no error is carried from step to step)."""
    m, o, _ = full
    if m.set_decode_engine(engine) != engine:
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    x = pkg.synth.synth_audio(1.2, seed=77); t = pkg.TimeEmbedding(3072).embed(6.0)
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
    ref_audio = o.encode_audio(mel)
    dec = m.decoder(); ids = np.array([1] + [32] * 37, dtype=np.int32)
    x0 = ref_audio[:38] + o.embed_tokens(ids)
    oc = o.cache(64); c = dec.create_cache_preallocated(64)
    rh = o.forward_hidden_with_cache(x0, t, oc); dec.forward_hidden_with_cache(x0[None], t, c)
    tok = int(o.lm_head(rh[-1:]).argmax()); worst_h = worst_l = 0.0
    for step in range(3):
        xs = ref_audio[38 + step:39 + step] + o.embed_tokens(np.array([tok], np.int32))
        rh = o.forward_hidden_with_cache(xs, t, oc); gh = dec.forward_hidden_with_cache(xs[None], t, c)[0]
        rl = o.lm_head(rh); gl = dec.lm_head(gh[None])[0]      # engine: the logits of the SAME launch (the rows handed back are recognised); per-operator: the GEMV
        worst_h = max(worst_h, rel_err(gh, rh)); worst_l = max(worst_l, rel_err(gl, rl))
        assert gl.shape == (1, 131072) and int(gl.argmax()) == int(rl.argmax())
        tok = int(rl.argmax())
    o.cache_free(oc)
    print(f"one decode launch vs oracle (engine {engine}): hidden {worst_h:.2e}, all 131072 logits {worst_l:.2e} of the largest")
    assert worst_h < TOL and worst_l < TOL


def test_full_decode_loop_and_prefill_knob_paths(pkg, full, monkeypatch):
    """The measurement knobs that select the OLDER forms of two round-3 changes still give the product's results at full size: the two-launch decode step
    (VOX_ENGINE_ARGMAX_IN=0: engine launch + argmax / embedding launch) and one step per graph (VOX_DECODE_UNROLL=1) -> identical ids; the prefill's separate
    finishing / RoPE / cache-write / conversion launches (VOX_PREFILL_NO_FUSED_FIN=1) -> the same hidden states to rounding (the fused kernels restate the same
    arithmetic; FMA contraction may differ by an ulp per element, which 26 layers carry to a few 1e-5)."""
    m, _, ctx = full
    if not m.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    t = pkg.TimeEmbedding(3072).embed(6.0); x = pkg.synth.synth_audio(6.0, seed=77)
    ids = m.transcribe_audio(x, t); assert len(ids) > 30
    for env in ({"VOX_ENGINE_ARGMAX_IN": "0"}, {"VOX_DECODE_UNROLL": "1"}, {"VOX_ENGINE_ARGMAX_IN": "0", "VOX_DECODE_UNROLL": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = m.transcribe_audio(x, t)
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(got, ids), f"{env}: ids differ from the default decode loop"
    assert np.array_equal(m.transcribe_audio(x, t), ids)
    dec = m.decoder(); x0 = (0.5 * np.random.default_rng(3).standard_normal((1, 38, 3072))).astype(np.float32)
    c = dec.create_cache_preallocated(64); a = dec.forward_hidden_with_cache(x0, t, c); c.close()
    monkeypatch.setenv("VOX_PREFILL_NO_FUSED_FIN", "1")
    c = dec.create_cache_preallocated(64); b = dec.forward_hidden_with_cache(x0, t, c); c.close()
    monkeypatch.delenv("VOX_PREFILL_NO_FUSED_FIN")
    err = float(np.abs(a - b).max() / np.abs(b).max())
    print(f"prefill, fused vs separate finishing launches: max rel diff {err:.2e}")
    assert err <= TOL                     # (ulp-level differences of the RoPE / SwiGLU expressions' contraction, carried through 26 layers: 3.6e-5 measured)


def test_full_decode_engine_lost_publish_times_out_and_the_utterance_is_served_anyway(pkg, full, monkeypatch, capfd):
    """Every wait inside the engine is bounded (20 ms): with one workgroup's publish suppressed (fault-injection flag 16384) the launch must END, the call must NOT return
    wrong ids -- the same utterance is decoded again on the per-operator path (a warning on stderr says so) -- the engine is re-armed for the next utterance, and after three
    strikes it is switched off for the life of the model.  Ids = those of the healthy engine throughout."""
    m0, _, ctx = full
    if not m0.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    x = pkg.synth.synth_audio(2.0, seed=5); t = pkg.TimeEmbedding(3072).embed(6.0)
    good = m0.transcribe_audio(x, t)
    monkeypatch.setenv("VOX_ENGINE_FLAGS", str(128 | 512 | 1 | 16384))
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    b = pkg.Q4ModelLoader.from_file(path).load(ctx)
    monkeypatch.delenv("VOX_ENGINE_FLAGS")
    try:
        import time
        for strike in (1, 2, 3):
            t0 = time.time()
            assert np.array_equal(b.transcribe_audio(x, t), good)        # served by the re-run on the per-operator launches
            assert time.time() - t0 < 30.0                               # bounded: a few 20 ms waits per launch, not a hang
            err = capfd.readouterr().err
            assert "hand-off timeout" in err and f"strike {strike} of 3" in err
            assert b.set_decode_engine(True) == (strike < 3)             # re-armed twice, off for good after the third strike
        assert np.array_equal(b.transcribe_audio(x, t), good)            # per-operator path from now on, silently
        assert "hand-off timeout" not in capfd.readouterr().err
    finally:
        b.close()


def test_full_piecewise_surface_reports_an_engine_timeout_on_the_device_resident_path(pkg, full, monkeypatch):
    """The piecewise decoder surface with one workgroup's publish suppressed (fault-injection flag 16384): a caller that stays on the device-resident entries -- whose only
    synchronisation is vox_argmax_rows on the CONTEXT, not on the model -- must not be handed silently wrong logits: the engine's error word reaches the host through a
    pinned buffer refreshed behind every launch, and the synchronising call of the SAME step fails loudly (round 5: vox_argmax_rows / vox_ctx_synchronize read the verdict of
    the context's pending engine steps; VOX_ERR_HIP, "hand-off timeout", a strike); after three strikes the per-operator launches serve the calls."""
    m0, _, ctx = full
    if not m0.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    monkeypatch.setenv("VOX_ENGINE_FLAGS", str(128 | 512 | 1 | 16384))
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    b = pkg.Q4ModelLoader.from_file(path).load(ctx)
    monkeypatch.delenv("VOX_ENGINE_FLAGS")
    D, V = 3072, 131072
    t = pkg.TimeEmbedding(D).embed(6.0); dec = b.decoder(); cache = dec.create_cache_preallocated(64)
    d_x = ctx.upload((0.1 * np.random.default_rng(3).standard_normal(D)).astype(np.float32)); d_l = ctx.alloc(V * 4)
    try:
        strikes = 0
        for step in range(8):
            try:
                hid = dec.forward_hidden_with_cache_dev(d_x, 1, t, cache)      # engine launch (a verdict that landed since the last call is checked here too)
                dec.lm_head_dev(hid, 1, d_l); pkg.argmax_rows_dev(ctx, d_l, 1, V)      # device-resident: synchronises the context only
            except pkg.VoxError as e:
                assert "hand-off timeout" in str(e); strikes += 1
        assert strikes == 3 and not b.set_decode_engine(True)      # three launches failed loudly, then the engine is off for the model
        hid = dec.forward_hidden_with_cache_dev(d_x, 1, t, cache)     # served by the per-operator launches now
        assert np.isfinite(ctx.download(hid, (D,))).all()
    finally:
        cache.close(); ctx.free(d_x); ctx.free(d_l); b.close()


def test_full_piecewise_surface_recovers_from_an_engine_timeout(pkg, full, monkeypatch):
    """VERDICT r4 item 3(c) / ADVICE r4: after an engine hand-off timeout on the device-resident piecewise path the header promises "repeat the step".  With every engine
    launch of a second model made to fail (fault-injection flag 16384) a caller that reads a token per step (vox_argmax_rows on the context, as bin/e2e_bench.rs:219-220
    does) sees the error in the step that failed, finds the cache length rolled back to that step's position, repeats the step -- and ends with EXACTLY the ids, the
    hidden rows and the cache length of an undisturbed model fed the same rows (the failed launches' garbage K / V rows are overwritten in place: no stale row is ever
    attended to).  Before round 5 the length stayed advanced and the repeated step appended behind a garbage row."""
    m0, _, ctx = full
    if not m0.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    D, V, N = 3072, 131072, 12
    t = pkg.TimeEmbedding(D).embed(6.0)
    rows = (0.1 * np.random.default_rng(11).standard_normal((N, D))).astype(np.float32)
    d_rows = ctx.upload(rows); d_l = ctx.alloc(V * 4)

    def loop(model, expect_failures):
        dec = model.decoder(); cache = dec.create_cache_preallocated(64)
        ids, hids, fails = [], [], 0
        try:
            for i in range(N):
                for attempt in range(6):
                    assert cache.seq_len() == i
                    try:
                        hid = dec.forward_hidden_with_cache_dev(d_rows + 4 * D * i, 1, t, cache)
                        dec.lm_head_dev(hid, 1, d_l); tok = pkg.argmax_rows_dev(ctx, d_l, 1, V)
                    except pkg.VoxError as e:
                        assert "hand-off timeout" in str(e) and "repeat the step" in str(e), str(e)
                        assert cache.seq_len() == i, (cache.seq_len(), i)      # rolled back to the failed step's position
                        fails += 1
                        continue
                    ids.append(int(tok[0])); hids.append(ctx.download(hid, (D,)).copy())
                    break
                else:
                    raise AssertionError(f"step {i} never succeeded")
            assert cache.seq_len() == N
        finally:
            cache.close()
        assert (fails > 0) == expect_failures, fails
        return np.array(ids), np.stack(hids), fails

    m0.set_decode_engine(False)
    ids_ref, hid_ref, _ = loop(m0, False)            # undisturbed, per-operator launches (what serves the faulty model once its engine is off)
    m0.set_decode_engine(True)
    ids_eng, hid_eng, _ = loop(m0, False)            # undisturbed, engine
    monkeypatch.setenv("VOX_ENGINE_FLAGS", str(128 | 512 | 1 | 16384))
    b = pkg.Q4ModelLoader.from_file(os.path.join(cache_dir(), "full_q4_seed42.gguf")).load(ctx)
    monkeypatch.delenv("VOX_ENGINE_FLAGS")
    try:
        ids_b, hid_b, fails = loop(b, True)
        assert fails == 3 and not b.set_decode_engine(True)      # three strikes, all on step 0, then the per-operator launches
        assert np.array_equal(ids_b, ids_ref) and np.array_equal(hid_b, hid_ref)      # bit-identical to the undisturbed per-operator run: no garbage row was left behind
        assert np.array_equal(ids_b, ids_eng) and np.abs(hid_b - hid_eng).max() <= 2e-4 * np.abs(hid_eng).max()
    finally:
        ctx.free(d_rows); ctx.free(d_l); b.close()


def test_full_engines_on_a_gpu_that_is_not_theirs_alone(pkg, full, capfd):
    """The engines' workgroups wait for each other, so all 256 must be resident: a long kernel on ANOTHER stream (64 x 1024-thread workgroups spinning for 60 ms --
    vox_debug_occupy) takes CUs away while an utterance / a batch is decoded.  Required: correct ids (whichever path produced them), no hang, and a working engine
    afterwards (the single-stream engine is re-armed after a timeout; it may also simply have waited the spin out, which is fine)."""
    m, _, ctx = full
    if not m.set_decode_engine(True):
        pytest.skip("decode engine not available on this device (needs 256 CUs)")
    x = pkg.synth.synth_audio(3.0, seed=9); t = pkg.TimeEmbedding(3072).embed(6.0)
    good = m.transcribe_audio(x, t)
    good_b = m.transcribe_batch([x] * 3, t)
    import time
    t0 = time.time()
    ctx.occupy(64, 60000)
    ids = m.transcribe_audio(x, t)
    assert np.array_equal(ids, good)
    ctx.occupy(64, 60000)
    outs = m.transcribe_batch([x] * 3, t)
    assert all(np.array_equal(a, b_) for a, b_ in zip(outs, good_b))
    assert time.time() - t0 < 60.0
    err = capfd.readouterr().err
    print("stderr during the occupied runs:", err.strip()[:400] or "(nothing: the engines waited the spin out or were not disturbed)")
    time.sleep(0.1); ctx.synchronize()
    assert m.set_decode_engine(True)                                     # still armed (at most two strikes here)
    assert np.array_equal(m.transcribe_audio(x, t), good)
    assert "hand-off timeout" not in capfd.readouterr().err              # ... and healthy again once the GPU is its own


def test_full_layout_only_arena_copy_start_up(pkg, full):
    """What ranks > 0 do at multi-GPU start-up, at full size on one GPU: VOX_LOAD_LAYOUT_ONLY model (no tensor data read) + the PRIMARY part of rank 0's arena
    copied in (the RCCL broadcast's payload: 2.5 GB, the Q4 row planes -- not the 5 GB with the tile-ordered copies) + vox_model_arena_finalize => the same 108 ids as the
    model that parsed the file, on the single-stream path (decode engine packed from the received planes) and on the batch path (tile-ordered copies rebuilt)."""
    m, _, ctx = full
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    b = pkg.Q4ModelLoader.from_file(path).load(ctx, layout_only=True)
    try:
        pa, na = m.arena(); pb, nb = b.arena()
        assert na == nb and 2.4e9 < na < 2.7e9, na                       # north_star: "RCCL broadcast of the 2.5 GB Q4 weights"
        assert m.weight_bytes() > 4.5e9                                   # the whole arena still holds both copies
        ctx.copy(pb, pa, na); b.arena_finalize()
        x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
        ids_a = m.transcribe_audio(x, t); ids_b = b.transcribe_audio(x, t)
        assert len(ids_a) == 108 and np.array_equal(ids_a, ids_b)
        bb = b.transcribe_batch([x, x], t)
        assert np.array_equal(bb[0], m.transcribe_batch([x, x], t)[0]) and np.array_equal(bb[0], bb[1])
    finally:
        b.close()


def test_full_model_replicate_second_context(pkg, full):
    """vox_model_replicate at full size: a second context on this GPU gets its replica from the loaded model alone -- arena laid out from the
    tensor manifest (no file), the 2.5 GB primary part copied device to device (hipMemcpyPeerAsync between two GPUs; a plain device copy here), derived copies
    rebuilt there -- and decodes the same 108 ids on the engine path and on the batch path; the source keeps working."""
    m, _, ctx = full
    ctx2 = pkg.Context(0)
    try:
        import time
        t0 = time.time(); r = m.replicate(ctx2); dt = time.time() - t0
        try:
            assert r.arena()[1] == m.arena()[1] and r.arena()[0] != m.arena()[0] and r.weight_bytes() == m.weight_bytes()
            x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
            ids_a = m.transcribe_audio(x, t); ids_r = r.transcribe_audio(x, t)
            assert len(ids_a) == 108 and np.array_equal(ids_a, ids_r)
            rb = r.transcribe_batch([x, x], t)
            assert np.array_equal(rb[0], m.transcribe_batch([x, x], t)[0]) and np.array_equal(rb[0], rb[1])
            print(f"vox_model_replicate: 2.5 GB primary arena + derived copies in {dt:.2f} s, same 108 ids")
        finally:
            r.close()
    finally:
        ctx2.close()


def test_full_two_sessions_one_gpu_same_ids(pkg, full, monkeypatch):
    """Two concurrent sessions on ONE GPU (shard.SessionPool: second context + vox_model_replicate + host thread) at full size: 140 clips of
    3 .. 20 s, un-chunked and as CLI chunks with file peaks -- every unit's ids equal, bit for bit and twice in a row, the ids of the SAME two sessions run one after the
    other (same inputs, same slot plans: VOX_BATCH_NO_CALIB pins the planner), and all but near-ties equal the one-session run's (a clip may meet the wide step in one
    plan and the launch chains in another: different K-split orders).  This is the test that found the
    MI355X packed-FP32 behaviour (csrc/vox_kernels.h VOX_NO_PK_F32): with rope_kernel's v_pk_mul_f32 op_sel:[0,1] the encoder output of a session changed whenever the other
    session's MFMA kernels shared its CUs, and 2 - 4 % of the clips came back with other ids, differently every run."""
    m, _, ctx = full
    shard = pkg.shard
    t = pkg.TimeEmbedding(3072).embed(6.0)
    clips = [pkg.synth.synth_audio(3.0 + 17.0 * ((37 * i) % 101) / 100.0, seed=4000 + i) * (0.3 + 0.1 * (i % 5)) for i in range(140)]
    was = m.set_batch_engine()[0]; m.set_batch_engine(False)      # (the pool runs without the batched engines; the launch chains and the wide step are bit-identical paths)
    monkeypatch.setenv("VOX_BATCH_NO_CALIB", "1")
    try:
        ref = m.transcribe_batch(clips, t)
        cc = pkg.ChunkConfig.voxtral().with_max_frames(1200)
        units, grp = [], []
        for i, x in enumerate(clips[:60]):
            for a, b in (pkg.chunk_plan(x.size, cc) if pkg.needs_chunking(x.size, cc) else [(0, x.size)]):
                units.append(x[a:b]); grp.append(i)
        assert len(units) > 70
        ref_u = m.transcribe_batch(units, t, norm_group=grp)
        with shard.SessionPool(pkg, ctx, m, 2) as pool:
            assert len(pool.models) == 2
            pool.MIN_UNITS_PER_SESSION = 32      # (the default keeps shares this small on one session)
            parts = pool.split([float(x.size) for x in clips])
            seq = [None] * len(clips)
            for k in range(2):      # the two sessions one after the other, each on its own (context, model)
                for i, o in zip(parts[k], pool.models[k].transcribe_batch([clips[i] for i in parts[k]], t)):
                    seq[i] = o
            for rep in range(2):
                got = pool.transcribe_batch(clips, t)
                bad = [i for i in range(len(clips)) if not np.array_equal(seq[i], got[i])]
                assert not bad, (rep, bad[:10])
            assert sum(int(np.array_equal(a, b)) for a, b in zip(ref, got)) >= len(clips) - 4
            parts_u = pool.split([float(x.size) for x in units], grp)
            assert all(len({k for k in range(2) for u in parts_u[k] if grp[u] == g}) == 1 for g in set(grp))      # a file's chunks stay in one session (device peak per call)
            seq_u = [None] * len(units)
            for k in range(2):
                for i, o in zip(parts_u[k], pool.models[k].transcribe_batch([units[i] for i in parts_u[k]], t, norm_group=[grp[i] for i in parts_u[k]])):
                    seq_u[i] = o
            got_u = pool.transcribe_batch(units, t, norm_group=grp)
            assert all(np.array_equal(a, b) for a, b in zip(seq_u, got_u))
            assert sum(int(np.array_equal(a, b)) for a, b in zip(ref_u, got_u)) >= len(units) - 3
        assert np.array_equal(m.transcribe_batch(clips[:20], t)[7], ref[7])      # the source model keeps working after the pool is gone
    finally:
        m.set_batch_engine(was)


def test_full_128_slots_two_wide_chains(pkg, full, monkeypatch):
    """More than four slot groups (end of round 6): 150 clips of 2 .. 8 s on up to 128 slots -- every step with n >= 4 active groups runs as two wide chains of ceil(n / 2) and
    floor(n / 2) groups on two streams.  Same ids as on <= 64 slots (VOX_BATCH_MAX_GROUPS=4) except at near-ties (a clip meets the three-group launch chains at other steps),
    reproducible call to call, and forced 7 and 8 groups agree with the planner's choice."""
    m, _, ctx = full
    t = pkg.TimeEmbedding(3072).embed(6.0)
    clips = [pkg.synth.synth_audio(2.0 + 6.0 * ((29 * i) % 89) / 88.0, seed=8000 + i) * (0.25 + 0.15 * (i % 5)) for i in range(150)]
    monkeypatch.setenv("VOX_BATCH_NO_CALIB", "1")
    monkeypatch.setenv("VOX_BATCH_MAX_GROUPS", "4"); ref = m.transcribe_batch(clips, t); monkeypatch.delenv("VOX_BATCH_MAX_GROUPS")
    a = m.transcribe_batch(clips, t); b = m.transcribe_batch(clips, t)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert sum(int(np.array_equal(x, y)) for x, y in zip(ref, a)) >= len(clips) - 4
    for G in (7, 8):
        monkeypatch.setenv("VOX_BATCH_SLOT_GROUPS", str(G))
        g = m.transcribe_batch(clips, t)
        assert sum(int(np.array_equal(x, y)) for x, y in zip(ref, g)) >= len(clips) - 4, G
    monkeypatch.delenv("VOX_BATCH_SLOT_GROUPS")


def test_full_model_set_sessions_two(pkg, full, monkeypatch):
    """vox_model_set_sessions at full size: 280 clips of 2 .. 6 s in ONE vox_transcribe_batch call, two sessions inside the library -- reproducible to the bit from call to
    call, and equal to the one-session call (on a shared context: launch chains) except at near-ties (a clip meets other step forms in another plan)."""
    m, _, ctx = full
    t = pkg.TimeEmbedding(3072).embed(6.0)
    clips = [pkg.synth.synth_audio(2.0 + 4.0 * ((53 * i) % 97) / 96.0, seed=6000 + i) * (0.3 + 0.1 * (i % 6)) for i in range(280)]
    monkeypatch.setenv("VOX_BATCH_NO_CALIB", "1")
    ctx.set_shared(True)
    try:
        ref = m.transcribe_batch(clips, t)
    finally:
        ctx.set_shared(False)
    m.set_sessions(2)
    try:
        a = m.transcribe_batch(clips, t); tm = m.timings(); b = m.transcribe_batch(clips, t)
    finally:
        m.set_sessions(1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert sum(int(np.array_equal(x, y)) for x, y in zip(ref, a)) >= len(clips) - 5
    assert tm["decode_tokens"] == sum(len(x) for x in a) and tm["graph_replays"] > 0
    x = pkg.synth.synth_audio(16.0, seed=1234)
    assert len(m.transcribe_audio(x, t)) == 108      # the single-stream engine path of the source model is untouched


def test_full_load_replicated_rccl_world1(pkg, full):
    """The multi-GPU start-up the product uses (shard.load_replicated: cli.py / wer.py / bench.py --gpus N) with a REAL RCCL process group on this one GPU (world 1, `nccl`
    backend; tests/rccl_startup_worker.py, its own process so torch's HIP runtime is loaded first): RCCL initialises, the broadcast executes on the library's arena memory
    (zero-copy torch view), a layout-only model filled by a broadcast and finalised decodes the same 108 ids as the model that parsed the file -- and as this process's."""
    import json, subprocess, sys
    shard = pkg.shard
    path = os.path.join(cache_dir(), "full_q4_seed42.gguf")
    env = dict(os.environ); env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_startup_worker.py"), path, str(shard.free_port())],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"ids_a"')]
    assert lines, "no result line from the worker:\n" + r.stdout[-1500:] + "\n--- stderr ---\n" + r.stderr[-3000:]
    res = json.loads(lines[-1])
    m, _, ctx = full
    x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(3072).embed(6.0)
    ids = [int(v) for v in m.transcribe_audio(x, t)]
    print(f"RCCL world-1 start-up: backend {res['backend']} {res['nccl_version']}, broadcast of {res['stats']['bytes'] / 1e9:.2f} GB in {res['stats']['seconds']:.3f} s")
    assert res["backend"] == "nccl" and res["stats"]["broadcast"] and 2.4e9 < res["stats"]["bytes"] < 2.7e9
    assert len(ids) == 108 and res["ids_a"] == ids and res["ids_b"] == ids


def test_full_30s_heavytail_vs_oracle_golden(pkg, orc):
    """Stress statistics at full size against the CPU oracle (tests/golden/make_fullsize_heavytail_golden.py): Student-t(4) block scales, six x50 outlier channels in
    the decoder's residual stream, |logit| up to 200, on a 30 s clip (234 decoder positions; the encoder's 750-frame window bites).  Exercises the hi/lo-bf16 splits of the
    MFMA GEMMs, the engine's x * 512 pre-scale and every fixed-order cross-CU sum on data with real-checkpoint dynamics.
      (a) end to end: all ids identical to the oracle's (up to a near-tie), on the decode engine and on the per-operator path; engine vs per-operator logits <= 2e-4 max;
      (b) stage by stage on the ORACLE's intermediate values: encoder + adapter output and lm_head within the stated 2e-4, the 38-token decoder prefill within 4e-4
          (the f32 summation-order noise of this fixture, measured: see the assertion);
      (c) end-to-end top logits within 1e-2 of the largest |logit| (SURVEY section 8(c)'s eps_q4; measured 7.4e-3): this synthetic decoder is ill-conditioned (measured: a 5.6e-5 relative perturbation of its input -- the
          two encoders' f32 summation-order noise -- moves its hidden state by 2.0e-3), so (b) is the precision statement and (c) only bounds the amplification."""
    import hashlib
    gpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_30s_heavytail_oracle.npz")
    g = np.load(gpath)
    path = os.path.join(cache_dir(), "full_q4_heavytail_seed43.gguf")
    if not os.path.exists(path):
        pkg.synth.write_synthetic_gguf(path + ".tmp", pkg.synth.ModelDims(), seed=43, heavy_tail=True); os.replace(path + ".tmp", path)
    hs = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            hs.update(chunk)
    assert hs.digest() == g["gguf_sha256"].tobytes()
    x = pkg.synth.synth_audio(30.0, seed=4321)
    assert hashlib.sha256(x.tobytes()).digest() == g["audio_sha256"].tobytes()
    ctx = pkg.Context(0); m = pkg.Q4ModelLoader.from_file(path).load(ctx); o = orc.Model(path)
    try:
        t = pkg.TimeEmbedding(3072).embed(6.0)
        mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
        assert mel.shape[0] == int(g["mel_frames"])
        rids, top1, top2, amax = g["ids"], g["top1"], g["top2"], float(g["logit_absmax"])
        assert amax > 15.0                                                         # the fixture really has large logits
        lgs = {}
        for engine in (True, False):
            if m.set_decode_engine(engine) != engine:
                continue
            ids, lg = m.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t, return_logits=True)
            assert len(ids) == len(rids) > 180
            agree = ids == rids
            stop = len(ids) if agree.all() else int(np.argmin(agree))
            if stop < len(ids):
                assert top1[stop] - top2[stop] <= 2e-2 * amax, f"engine={engine}: ids differ at step {stop} with a clear margin"
            near = (top1 - top2) <= 2e-2 * amax; first_tie = int(np.argmax(near)) if near.any() else len(rids)
            assert stop >= first_tie, f"engine={engine}: ids agree for {stop} steps only; the oracle's first near-tie is at step {first_tie}"      # (measured: all 196)
            err = float(np.abs(lg[:stop].max(axis=1) - top1[:stop]).max())
            assert err <= 1e-2 * amax, (engine, err, amax)                                             # (c) SURVEY section 8(c)'s eps_q4 (measured 7.4e-3: conditioning, see the docstring)
            print(f"heavy-tail golden (engine={engine}): ids agree for {stop}/{len(ids)} steps; max top-logit error {err:.3e} at |logit| max {amax:.1f}")
            ids_b = m.transcribe_batch([x], t)[0]
            assert (ids_b[:stop] == rids[:stop]).all()
            lgs[engine] = lg
        if True in lgs and False in lgs:
            d = float(np.abs(lgs[True] - lgs[False]).max()); top = float(np.abs(lgs[False]).max())
            print(f"heavy-tail: decode engine vs per-operator path max |dlogit| {d:.3e} of {top:.1f}")
            assert d <= TOL * top
        # (b) stage by stage on the oracle's own intermediates (4 s clip: the oracle's encoder takes seconds)
        xs = pkg.synth.synth_audio(4.0, seed=4321); xn = xs.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
        mel_s = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
        ref_audio = o.encode_audio(mel_s); out_audio = m.encode_audio(mel_s[None])[0]
        e_enc = rel_err(out_audio, ref_audio)
        dec = m.decoder(); pid = np.array([1] + [32] * 37, dtype=np.int32)
        x0 = ref_audio[:38] + o.embed_tokens(pid)
        oc = o.cache(64); c = dec.create_cache_preallocated(64)
        rh = o.forward_hidden_with_cache(x0, t, oc); gh = dec.forward_hidden_with_cache(x0[None], t, c)[0]
        e_dec = rel_err(gh, rh)
        e_lm = rel_err(dec.lm_head(rh[None, -1:])[0], o.lm_head(rh[-1:]))
        # The 38-token prefill of THIS fixture is bounded at 2 x TOL: its outlier channels make the result depend on the f32 summation order at the 1e-4 level on
        # both sides (the oracle sums sequentially in f32).  Measured over eleven K decompositions of the same GEMMs (profiles/r03_heavytail_prefill_decompositions.txt):
        # 1.4e-4 .. 3.3e-4 against the oracle, and the oracle itself 1.6e-4 from the mean of the eleven -- no decomposition is "the" accurate one.
        print(f"heavy-tail stage parity: encoder+adapter {e_enc:.3e}, decoder prefill {e_dec:.3e} (bound {2 * TOL:.0e}), lm_head {e_lm:.3e} (bound {TOL:.0e})")
        assert e_enc < TOL and e_dec < 2 * TOL and e_lm < TOL
    finally:
        m.close(); o.close(); ctx.close()
