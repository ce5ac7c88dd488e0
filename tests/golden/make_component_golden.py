#!/usr/bin/env python3
"""Generate tests/golden/ref_python_components.npz by RUNNING the reference's own per-component PyTorch forward (/root/reference/scripts/reference_forward.py:
swiglu :86-88 / test_swiglu :143-168, test_conv :171-204, test_attention :207-273, test_ada_rms_norm :276-326 -- the script that produces the test_data the reference's
Rust tests `test_swiglu_vs_reference`, `test_conv_vs_reference`, `test_attention_vs_reference`, `test_ada_modulation_vs_reference` load) on SYNTHETIC weights of the REAL
shapes.  The script reads its weights through safetensors.safe_open and writes through save_tensor: both are replaced before its test_* functions are called, so the
arithmetic is the reference's, line for line, and only the checkpoint (absent offline) is synthetic.

Runs ONLY in the build container (the reference tree does not exist on the GPU box).  The weights are not stored: tests regenerate them with
model_fixtures.component_weight (numpy default_rng keyed by the tensor name); the fixture holds the script's inputs and outputs (a few hundred KB).
    python tests/golden/make_component_golden.py"""
import importlib.util, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from model_fixtures import component_weight, COMPONENT_SEED

REF = "/root/reference/scripts/reference_forward.py"
spec = importlib.util.spec_from_file_location("ref_component_forward", REF)
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)


class FakeCheckpoint:
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def get_tensor(self, name): return torch.from_numpy(component_weight(name))


captured = {}
mod.safe_open = lambda *a, **k: FakeCheckpoint()
mod.save_tensor = lambda name, tensor, output_dir=None: captured.__setitem__(name, tensor.float().cpu().numpy().copy())
for fn in ("test_swiglu", "test_conv", "test_attention", "test_ada_rms_norm"):
    getattr(mod, fn)()
keep = {k: v for k, v in captured.items() if not any(t in k for t in ("_w1", "_w2", "_w3", "_weight", "_bias", "_wq", "_wk", "_wv", "_wo", "_w0"))}
for k in ("swiglu_w1", "conv1_weight", "attn_wq", "ada_rms_norm_w0"):      # the weights the script saw ARE the regenerable ones
    name = {"swiglu_w1": "mm_streams_embeddings.embedding_module.whisper_encoder.transformer.layers.0.feed_forward.w1.weight",
            "conv1_weight": "mm_streams_embeddings.embedding_module.whisper_encoder.conv_layers.0.conv.weight",
            "attn_wq": "mm_streams_embeddings.embedding_module.whisper_encoder.transformer.layers.0.attention.wq.weight",
            "ada_rms_norm_w0": "layers.0.ada_rms_norm_t_cond.0.weight"}[k]
    assert np.array_equal(captured[k], component_weight(name)), k
out = os.path.join(HERE, "ref_python_components.npz")
np.savez_compressed(out, seed=np.int64(COMPONENT_SEED), **{k: v.astype(np.float32) for k, v in keep.items()})
print("wrote", out, {k: v.shape for k, v in keep.items()})
