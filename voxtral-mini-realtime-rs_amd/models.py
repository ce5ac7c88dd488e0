"""Host-side mirror of the reference's f32 model surface (`src/models`): VoxtralModelLoader (models/loader.rs:29-78)
-> VoxtralModel with the same forward entry points as the Q4 model (models/voxtral.rs:70-447, bin/transcribe.rs:362-438)."""
from __future__ import annotations

import ctypes as C

from ._lib import check, lib
from .gguf import Context, Q4VoxtralModel


class VoxtralModel(Q4VoxtralModel):
    """f32 SafeTensors model: encode_audio / transcribe_streaming (= transcribe_f32_with_model) / decoder() are inherited."""


class VoxtralModelLoader:
    def __init__(self, path):
        self.path = str(path)

    @classmethod
    def from_file(cls, path):
        return cls(path)

    def load(self, ctx: Context) -> VoxtralModel:
        h = C.c_void_p()
        check(lib().vox_f32_model_load(ctx.h, self.path.encode(), C.byref(h)))
        return VoxtralModel(ctx, h)
