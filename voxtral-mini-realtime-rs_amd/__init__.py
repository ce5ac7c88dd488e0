"""MI355X-native Voxtral-Mini realtime ASR hot path (gfx950 HIP kernels behind a C ABI).

Python here is a thin ctypes mirror of the reference's Rust surface (src/audio, src/gguf,
src/models); all compute lives in csrc/ -> libvoxtral_hip.so.  There is no CPU fallback:
`lib()` raises if the HIP library is missing.
"""
from . import synth  # noqa: F401  (numpy-only tooling: synthetic GGUF / audio)
from . import build  # noqa: F401
from ._lib import lib, VoxError  # noqa: F401
from .audio import (PadConfig, ChunkConfig, MelSpectrogram, pad_audio, chunk_audio, chunk_plan, needs_chunking, resample_len,  # noqa: F401
                    peak_normalize, resample, resample_to_16k, resample_plan, TimeEmbedding)
from .tokenizer import VoxtralTokenizer  # noqa: F401
from . import wer  # noqa: F401  (WER / CER harness, scripts/eval_wer.py)
from . import export  # noqa: F401  (SafeTensors -> Q4_0 GGUF)
from . import shard  # noqa: F401  (multi-GPU: LPT sharding of independent utterances; torch imported lazily)
from .models import VoxtralModel, VoxtralModelLoader  # noqa: F401
from .gguf import (Context, device_count, GgufReader, Q4Tensor, Q4Linear, q4_matmul, Q4ModelLoader, Q4VoxtralModel, tensor_add_dev, argmax_rows_dev, linear_forward, conv_downsample)  # noqa: F401
