"""Tekken tokenizer, decode-only (reference: src/tokenizer/mod.rs:70-232).  ids >= 1000 map to vocab[id - 1000]
bytes (base64 `token_bytes`, else `token_str`); control entries are skipped; bytes are joined and decoded as lossy UTF-8.
CPU post-processing, needed for text / WER, not part of the accelerated path."""
from __future__ import annotations

import base64
import json

TEXT_TOKEN_OFFSET = 1000


class VoxtralTokenizer:
    def __init__(self, tekken: dict):
        self._vocab_size = int(tekken["config"]["default_vocab_size"])
        vocab = tekken["vocab"]
        self.vocab_bytes = [None] * len(vocab)
        self.special_tokens = {}
        for idx, e in enumerate(vocab):
            if e.get("is_control", False):                       # mod.rs:82-87
                if e.get("token_str") is not None:
                    self.special_tokens[int(e["rank"])] = e["token_str"]
                continue
            b64 = e.get("token_bytes")
            if b64 is not None:
                try:
                    self.vocab_bytes[idx] = base64.b64decode(b64, validate=True); continue
                except Exception:
                    pass
            if e.get("token_str") is not None:
                self.vocab_bytes[idx] = e["token_str"].encode()

    @classmethod
    def from_file(cls, path):
        with open(path, "rb") as f:
            return cls(json.load(f))

    @classmethod
    def from_model_dir(cls, d):
        import os
        return cls.from_file(os.path.join(d, "tekken.json"))

    @classmethod
    def from_json(cls, s: str):
        return cls(json.loads(s))

    def decode(self, ids) -> str:
        """mod.rs:170-194"""
        out = bytearray()
        for i in ids:
            i = int(i)
            if i < TEXT_TOKEN_OFFSET:
                continue
            v = i - TEXT_TOKEN_OFFSET
            if v < len(self.vocab_bytes) and self.vocab_bytes[v] is not None:
                out += self.vocab_bytes[v]
        return out.decode("utf-8", errors="replace")

    def decode_token(self, i: int):
        i = int(i)
        if i < TEXT_TOKEN_OFFSET:
            return self.special_tokens.get(i)
        v = i - TEXT_TOKEN_OFFSET
        if v < len(self.vocab_bytes) and self.vocab_bytes[v] is not None:
            return self.vocab_bytes[v].decode("utf-8", errors="replace")
        return None

    def vocab_size(self) -> int:
        return self._vocab_size
