"""Shared helpers for model-level tests: deterministic synthetic GGUFs (cached in the temp dir)."""
import os
import tempfile

import numpy as np

from __graft_entry__ import load_package

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_golden.npz")


def cache_dir():
    d = os.path.join(tempfile.gettempdir(), "vox_test_models")
    os.makedirs(d, exist_ok=True)
    return d


def synth_gguf(dims, seed, tag):
    S = load_package().synth
    p = os.path.join(cache_dir(), f"{tag}_{seed}.gguf")
    if not os.path.exists(p):
        tmp = p + f".tmp{os.getpid()}"
        S.write_synthetic_gguf(tmp, dims, seed=seed)
        os.replace(tmp, p)
    return p


def golden():
    return np.load(GOLDEN)


def golden_gguf():
    S = load_package().synth; g = golden()
    dims = S.ModelDims(**{str(k): int(v) for k, v in zip(g["dims_keys"], g["dims"])})
    return synth_gguf(dims, int(g["seed"]), "golden"), dims


def tiny_gguf(seed=7, **kw):
    S = load_package().synth
    tag = "tiny" + "".join(f"_{k}{v}" for k, v in sorted(kw.items()))
    return synth_gguf(S.tiny_dims(**kw), seed, tag), S.tiny_dims(**kw)


def fake_mel(T, seed=0, n_mels=128):
    rng = np.random.default_rng(seed)
    return (0.6 * rng.standard_normal((n_mels, T)) + 0.3).astype(np.float32)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def tiny_f32_pair(seed=9):
    """(safetensors path for the HIP f32 path, all-F32 GGUF with the same values for the oracle)."""
    S = load_package().synth; d = S.tiny_dims()
    st = os.path.join(cache_dir(), f"tiny_f32_{seed}.safetensors"); gg = os.path.join(cache_dir(), f"tiny_f32_{seed}.gguf")
    if not os.path.exists(st):
        S.write_synthetic_safetensors(st + ".tmp", d, seed); os.replace(st + ".tmp", st)
    if not os.path.exists(gg):
        S.write_synthetic_dense_gguf(gg + ".tmp", d, seed); os.replace(gg + ".tmp", gg)
    return st, gg, d


def check_greedy_ids(ids, rids, rlg, tol):
    """Greedy ids must equal the oracle's; the only admissible first disagreement is at a step whose oracle top-2
    logit margin is below 10x the logit tolerance (a near-tie; after it the autoregressive sequences may diverge)."""
    ids = np.asarray(ids); rids = np.asarray(rids)
    assert ids.shape == rids.shape
    agree = ids == rids
    if agree.all():
        return len(ids)
    first = int(np.argmin(agree))
    srt = np.sort(rlg[first]); margin = srt[-1] - srt[-2]
    assert margin <= 10 * tol * max(1.0, float(np.abs(rlg).max())), f"ids differ at step {first} with a clear margin {margin}"
    return first


def single_stream_reference(pkg, ctx, model, x, t_embed):
    """(ids, logits) of the single-stream EAGER path (logits tap) for 16 kHz samples x: the per-sequence reference the batched paths are
    held to -- same device log-mel kernel as the batch path, then vox_transcribe_streaming with return_logits."""
    mel = pkg.MelSpectrogram.voxtral(ctx).compute_log(pkg.pad_audio(pkg.peak_normalize(x)))
    return model.transcribe_streaming(np.ascontiguousarray(mel.T)[None], t_embed, return_logits=True)


def check_batch_rows(pkg, ctx, model, clips, t_embed, outs, tol):
    """Every sequence of a batched transcription must equal the single-stream ids up to the first near-tie of ITS OWN single-stream logits
    (top-2 margin < 10 x tol x max|logit|); returns how many sequences are identical end to end."""
    assert len(outs) == len(clips)
    identical = 0
    for x, ids in zip(clips, outs):
        rids, rlg = single_stream_reference(pkg, ctx, model, x, t_embed)
        first = check_greedy_ids(ids, rids, rlg, tol)
        identical += int(first == len(rids))
    return identical


def dense_head_sha(path, nbytes=64 << 20):
    """sha256 of the first 64 MB of a (8.9 GB) synthetic dense checkpoint: cheap identity check of the deterministic generator's output."""
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha256(f.read(nbytes)).digest()


def full_dense_safetensors(seed=7, heavy_tail=False):
    """The full-size synthetic BF16 SafeTensors checkpoint of the f32 path (8.9 GB, written in a few seconds, cached in the temp dir);
    heavy_tail: the stress statistics of synth.dense_checkpoint_tensors."""
    S = load_package().synth
    st = os.path.join(cache_dir(), f"full_dense{'_heavytail' if heavy_tail else ''}_seed{seed}.safetensors")
    if not os.path.exists(st):
        S.write_fast_dense_checkpoint(st + ".tmp", None, S.ModelDims(), seed=seed, heavy_tail=heavy_tail); os.replace(st + ".tmp", st)
    return st


# ---- synthetic checkpoint tensors for the reference's per-component forward script (tests/golden/make_component_golden.py): real shapes, regenerable by name
COMPONENT_SEED = 20260925
_ENC = "mm_streams_embeddings.embedding_module.whisper_encoder."
COMPONENT_SHAPES = {
    _ENC + "transformer.layers.0.feed_forward.w1.weight": (5120, 1280), _ENC + "transformer.layers.0.feed_forward.w2.weight": (1280, 5120),
    _ENC + "transformer.layers.0.feed_forward.w3.weight": (5120, 1280),
    _ENC + "conv_layers.0.conv.weight": (1280, 128, 3), _ENC + "conv_layers.0.conv.bias": (1280,),
    _ENC + "conv_layers.1.conv.weight": (1280, 1280, 3), _ENC + "conv_layers.1.conv.bias": (1280,),
    _ENC + "transformer.layers.0.attention.wq.weight": (2048, 1280), _ENC + "transformer.layers.0.attention.wk.weight": (2048, 1280),
    _ENC + "transformer.layers.0.attention.wv.weight": (2048, 1280), _ENC + "transformer.layers.0.attention.wo.weight": (1280, 2048),
    _ENC + "transformer.layers.0.attention.wq.bias": (2048,), _ENC + "transformer.layers.0.attention.wv.bias": (2048,), _ENC + "transformer.layers.0.attention.wo.bias": (1280,),
    "layers.0.ada_rms_norm_t_cond.0.weight": (32, 3072), "layers.0.ada_rms_norm_t_cond.2.weight": (3072, 32),
}


def component_weight(name):
    """The synthetic tensor `name` of COMPONENT_SHAPES: N(0, sigma^2) f32, sigma = 0.03 (weights) / 0.02 (biases); numpy default_rng keyed by (seed, crc32(name))."""
    import zlib
    shape = COMPONENT_SHAPES[name]
    rng = np.random.default_rng([COMPONENT_SEED, zlib.crc32(name.encode())])
    return ((0.02 if name.endswith(".bias") else 0.03) * rng.standard_normal(shape)).astype(np.float32)
