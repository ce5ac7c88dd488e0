"""smoke(): one tiny end-to-end invocation of the hot path on cuda:0, checked against the CPU oracle."""
import numpy as np


def run_smoke():
    from __graft_entry__ import load_package
    import oracle_lib as orc
    from model_fixtures import tiny_gguf
    pkg = load_package()
    pkg.lib()                                   # raises if libvoxtral_hip.so is missing: no fallback
    ctx = pkg.Context(0)
    path, _ = tiny_gguf()
    model = pkg.Q4ModelLoader.from_file(path).load(ctx)
    oracle = orc.Model(path)
    t = pkg.TimeEmbedding(model.config.dec_dim).embed(6.0)
    x = pkg.synth.synth_audio(2.0, seed=21)
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
    rids, rlg = oracle.transcribe_streaming(mel, t, want_logits=True)
    gmel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
    assert np.abs(gmel.T - mel).max() < 1e-4, "log-mel mismatch"
    ids, lg = model.transcribe_streaming(mel[None], t, return_logits=True)
    assert len(ids) == len(rids) > 0
    err = np.abs(lg - rlg).max() / max(1.0, np.abs(rlg).max())
    assert err < 2e-4, f"decoder logits differ from the oracle: {err}"
    ids_audio = model.transcribe_audio(x, t)    # full path from samples, graph-replayed decode
    from model_fixtures import check_greedy_ids
    check_greedy_ids(ids, rids, rlg, 2e-4); check_greedy_ids(ids_audio, rids, rlg, 2e-4)
    print(f"smoke ok: {len(ids)} ids, max logit err {err:.2e}, timings {model.timings()}")
    model.close(); oracle.close(); ctx.close()
