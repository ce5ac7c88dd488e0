"""The oracle's isolated operators against the reference's OWN per-component PyTorch forward (scripts/reference_forward.py, run on synthetic weights of the real shapes by
tests/golden/make_component_golden.py): SwiGLU MLP, the two-stage conv downsampler, an encoder attention block (biases, RoPE theta 1e6, causal) and the Ada modulation.
These are the vectors the reference's Rust tests test_swiglu_vs_reference (models/layers/swiglu.rs:101), test_conv_vs_reference (conv.rs), test_attention_vs_reference
(attention.rs) and test_ada_modulation_vs_reference (rms_norm.rs) load from test_data/ -- regenerated here because the checkpoint they were made from is not available offline."""
import os

import numpy as np
import pytest

from model_fixtures import component_weight, COMPONENT_SEED, rel_err

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_python_components.npz")
ENC = "mm_streams_embeddings.embedding_module.whisper_encoder."
TOL = 2e-4


@pytest.fixture(scope="module")
def g():
    d = np.load(G)
    assert int(d["seed"]) == COMPONENT_SEED
    return d


def _vec(fn, a):
    return np.array([fn(float(v)) for v in a.ravel()], dtype=np.float32).reshape(a.shape)


def test_swiglu_vs_reference_python(orc, g):
    """w2(silu(w1 x) * w3 x) (reference_forward.py:86-88; the oracle's FFN is exactly this composition of its matmul and silu, vox_oracle.c)."""
    x = g["swiglu_input"][0]
    w1, w2, w3 = (component_weight(ENC + f"transformer.layers.0.feed_forward.w{i}.weight") for i in (1, 2, 3))
    h = _vec(orc.lib().orc_silu, orc.reference_matmul(x, w1)) * orc.reference_matmul(x, w3)
    out = orc.reference_matmul(h, w2)
    assert rel_err(out, g["swiglu_output"][0]) < TOL


def test_conv_downsampler_vs_reference_python(orc, g):
    """gelu(conv1d k3 s2 p1) twice at the real shapes 128 -> 1280 -> 1280 (reference_forward.py:185-192, models/layers/conv.rs:78-83)."""
    x = np.ascontiguousarray(g["conv_input"][0]); L = orc.lib()
    w1, b1 = component_weight(ENC + "conv_layers.0.conv.weight"), component_weight(ENC + "conv_layers.0.conv.bias")
    w2, b2 = component_weight(ENC + "conv_layers.1.conv.weight"), component_weight(ENC + "conv_layers.1.conv.bias")
    l1 = L.orc_conv_out_len(x.shape[1]); y1 = np.zeros((1280, l1), np.float32); L.orc_conv1d_gelu(x, 128, x.shape[1], w1, b1, 1280, y1)
    l2 = L.orc_conv_out_len(l1); y2 = np.zeros((1280, l2), np.float32); L.orc_conv1d_gelu(y1, 1280, l1, w2, b2, 1280, y2)
    assert (l1, l2) == (50, 25) and y2.shape == g["conv_output"][0].shape
    assert rel_err(y2, g["conv_output"][0]) < TOL


def test_encoder_attention_block_vs_reference_python(orc, g):
    """q / k / v projections (q, v biased, k not), interleaved RoPE theta 1e6, causal softmax attention, output projection + bias over 10 tokens, 32 heads x 64
    (reference_forward.py:226-262; gguf/model.rs:77-122, rope.rs:77-141)."""
    x = g["attn_input"][0]; S, H, hd = x.shape[0], 32, 64; L = orc.lib()
    w = {k: component_weight(ENC + f"transformer.layers.0.attention.{k}.weight") for k in ("wq", "wk", "wv", "wo")}
    b = {k: component_weight(ENC + f"transformer.layers.0.attention.{k}.bias") for k in ("wq", "wv", "wo")}
    q = orc.reference_matmul(x, w["wq"]) + b["wq"]; k = orc.reference_matmul(x, w["wk"]); v = orc.reference_matmul(x, w["wv"]) + b["wv"]
    q = np.ascontiguousarray(q.reshape(S, H, hd)); k = np.ascontiguousarray(k.reshape(S, H, hd)); v = np.ascontiguousarray(v.reshape(S, H, hd))
    L.orc_rope(q, S, H, hd, 0, 1e6); L.orc_rope(k, S, H, hd, 0, 1e6)
    att = np.zeros((S, H * hd), np.float32); L.orc_attention(q, k, v, S, S, H, H, hd, 0, 1, -1, att)
    out = orc.reference_matmul(att, w["wo"]) + b["wo"]
    assert rel_err(out, g["attn_output"][0]) < TOL


def test_ada_modulation_vs_reference_python(orc, g):
    """x * (1 + w2 gelu(w0 t)) (reference_forward.py:308-317; gguf/model.rs:382-385, the scale the HIP path precomputes per t_embed)."""
    x, t = g["ada_rms_norm_input"][0], g["ada_rms_norm_t_embed"][0]
    w0, w2 = component_weight("layers.0.ada_rms_norm_t_cond.0.weight"), component_weight("layers.0.ada_rms_norm_t_cond.2.weight")
    scale = orc.reference_matmul(_vec(orc.lib().orc_gelu, orc.reference_matmul(t, w0)), w2)
    assert rel_err(scale, g["ada_rms_norm_scale"][0]) < TOL
    assert rel_err(x * (1.0 + scale), g["ada_rms_norm_output"][0]) < TOL
