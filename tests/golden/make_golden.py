#!/usr/bin/env python3
"""Generate tests/golden/ref_python_golden.npz by IMPORTING the reference's own PyTorch restatement
(/root/reference/scripts/test_proper_inference.py: rms_norm :34-37, rope_freqs/apply_rope :39-54,
time_embedding :56-60, run_encoder :101-191, run_decoder_step :193-274) and running it on synthetic
weights in the real tensor layout.

Runs ONLY in the build container (the reference tree does not exist on the GPU box); the fixtures it
writes are committed and are what the tests read.  `mistral_common` (tokenizer / audio I/O, not on the
hot path) is not installed here, so it is stubbed before import -- none of the stubbed names is called.

The script hard-codes 32 encoder layers x 32 heads x 64, the 5120-wide reshape, 26 decoder layers and
32/8 heads x 128; widths it does not hard-code (FFN, decoder d_model, vocab) are reduced so the fixture
inputs regenerate in seconds:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/scripts/test_proper_inference.py"

GOLDEN_DIMS = dict(enc_layers=32, enc_dim=1280, enc_heads=32, enc_ffn=256, dec_layers=26, dec_dim=256,
                   dec_heads=32, dec_kv_heads=8, dec_ffn=512, vocab=512)
GOLDEN_SEED = 20260924


def import_reference():
    from transformers.audio_utils import mel_filter_bank as hf_bank      # before the stubs below (transformers probes for mistral_common)
    for name in ["mistral_common", "mistral_common.tokens", "mistral_common.tokens.tokenizers",
                 "mistral_common.tokens.tokenizers.mistral", "mistral_common.protocol",
                 "mistral_common.protocol.transcription", "mistral_common.protocol.transcription.request",
                 "mistral_common.protocol.instruct", "mistral_common.protocol.instruct.chunk", "mistral_common.audio"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["mistral_common.tokens.tokenizers.mistral"].MistralTokenizer = object
    sys.modules["mistral_common.protocol.transcription.request"].TranscriptionRequest = object
    sys.modules["mistral_common.protocol.transcription.request"].StreamingMode = object
    sys.modules["mistral_common.protocol.instruct.chunk"].RawAudio = object
    sys.modules["mistral_common.audio"].Audio = object
    # compute_mel (:62-98) imports mistral_common.audio.mel_filter_bank inside the function: the Slaney-scale, Slaney-normalised bank.
    # mistral-common is not installed; transformers.audio_utils.mel_filter_bank (installed) is the implementation mistral-common's was
    # taken from -- same formula, selected by norm="slaney", mel_scale="slaney"; the reference's own Rust bank follows it (mel.rs:260-339)
    sys.modules["mistral_common.audio"].mel_filter_bank = lambda num_frequency_bins, num_mel_bins, min_frequency, max_frequency, sampling_rate: \
        hf_bank(num_frequency_bins=num_frequency_bins, num_mel_filters=num_mel_bins, min_frequency=min_frequency, max_frequency=max_frequency,
                sampling_rate=sampling_rate, norm="slaney", mel_scale="slaney")
    spec = importlib.util.spec_from_file_location("ref_proper_inference", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class TensorFile:
    """Stands in for safetensors.safe_open: get_tensor(name) -> torch tensor (dequantised weights)."""

    def __init__(self, dense):
        self.dense = dense

    def get_tensor(self, name):
        return torch.from_numpy(np.ascontiguousarray(self.dense[name]))


def golden_inputs():
    rng = np.random.default_rng(GOLDEN_SEED)
    mel = (0.6 * rng.standard_normal((128, 160)) + 0.3).astype(np.float32)      # log-mel-like, T=160 -> S_enc=40 -> 10 tokens
    dec_x = (0.5 * rng.standard_normal((12, 256))).astype(np.float32)           # decoder inputs (audio+text embeds), 12 positions
    norm_x = rng.standard_normal((5, 96)).astype(np.float32); norm_w = (1 + 0.1 * rng.standard_normal(96)).astype(np.float32)
    rope_x = rng.standard_normal((1, 7, 3, 64)).astype(np.float32)
    # audio for the log-mel fixture: 1.3 s of the SURVEY 8(d) synthetic clip recipe (two tones + noise, fades), peak 0.95, already padded
    # the way the pipeline pads (32 left-pad tokens of silence here: the python script uses mistral-common's 32, not the Rust 76)
    n = 20800; tt = np.arange(n) / 16000.0
    a = 0.3 * np.sin(2 * np.pi * 220.0 * tt) + 0.2 * np.sin(2 * np.pi * (440.0 + 30.0 * tt) * tt) + 0.05 * rng.standard_normal(n)
    a[:800] *= np.linspace(0, 1, 800); a[-800:] *= np.linspace(1, 0, 800); a *= 0.95 / np.abs(a).max()
    mel_audio = np.concatenate([np.zeros(32 * 1280), a, np.zeros(1280 - n % 1280 + 17 * 1280)]).astype(np.float32)
    return dict(mel=mel, dec_x=dec_x, norm_x=norm_x, norm_w=norm_w, rope_x=rope_x, mel_audio=mel_audio)


def main():
    from __graft_entry__ import load_package
    S = load_package().synth
    ref = import_reference()
    torch.manual_seed(0); torch.set_num_threads(8)
    tmp = "/tmp/vox_golden.gguf"
    S.write_synthetic_gguf(tmp, S.ModelDims(**GOLDEN_DIMS), seed=GOLDEN_SEED)
    f = TensorFile(S.gguf_dense_f32(tmp))
    inp = golden_inputs(); out = {}
    with torch.no_grad():
        out["rms_norm"] = ref.rms_norm(torch.from_numpy(inp["norm_x"]), torch.from_numpy(inp["norm_w"])).numpy()
        cos, sin = ref.rope_freqs(64, 7)
        out["rope"] = ref.apply_rope(torch.from_numpy(inp["rope_x"]), cos, sin).numpy()
        out["time_embedding_6"] = ref.time_embedding(torch.tensor([6.0]), dim=256).numpy()[0]
        out["time_embedding_full"] = ref.time_embedding(torch.tensor([6.0]), dim=3072).numpy()[0]
        out["log_mel"] = ref.compute_mel(inp["mel_audio"]).numpy()                                     # [128, T] (:62-98)
        out["encoder_out"] = ref.run_encoder(torch.from_numpy(inp["mel"]), f).numpy()[0]              # [10, 256]
        x = torch.from_numpy(inp["dec_x"]).unsqueeze(0)
        t_embed = ref.time_embedding(torch.tensor([6.0]), dim=256)                                     # [1, 256]
        logits, hidden = ref.run_decoder_step(x, torch.zeros_like(x), t_embed, f, position=0)
        out["decoder_logits"] = logits.numpy()[0]; out["decoder_hidden"] = hidden.numpy()[0]
    meta = dict(dims=np.array(list(GOLDEN_DIMS.values()), dtype=np.int64), dims_keys=np.array(list(GOLDEN_DIMS.keys())),
                seed=np.int64(GOLDEN_SEED))
    np.savez_compressed(os.path.join(HERE, "ref_python_golden.npz"), **{"in_" + k: v for k, v in inp.items()},
                        **{"out_" + k: v for k, v in out.items()}, **meta)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()))
    os.remove(tmp)


if __name__ == "__main__":
    main()
