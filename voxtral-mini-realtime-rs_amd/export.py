"""SafeTensors -> Q4_0 GGUF exporter.  The reference ships no exporter: its `voxtral-q4.gguf` is published
externally (CLAUDE.md:75-76) and only a test-grade quantiser exists (src/gguf/tests.rs:24-57).  This tool removes that dependency: it reads
the published `consolidated.safetensors` (BF16 / F16 / F32; names of models/weights.rs:219-397), quantises every tensor the Q4 loader requires
as Q4_0 (gguf/loader.rs:385-441: all encoder / adapter / decoder linears, the Ada t_cond projections and tok_embeddings) and writes the
rest (norms, biases, conv weights) as F32, in the GGUF v3 layout the reader expects (gguf/reader.rs:105-188: reversed dims, 32-byte alignment).

Quantisers (both produce the 18-byte block {f16 d; 16 nibble bytes}, element i <-> low nibble of byte i, i + 16 <-> high nibble):
  * "reference" (default): src/gguf/tests.rs:24-57 -- d = amax / 7, q = min(15, trunc(v / d + 8.5));
  * "ggml": llama.cpp's quantize_row_q4_0_ref -- d = (the value of largest magnitude) / -8, q = min(15, trunc(v / d + 8.5)): uses all 16 levels.
Host-side numpy tooling (a one-off, minutes for the 4.4 G-parameter model); nothing here runs on the accelerated path.

    python -m voxtral_mini_realtime_rs_amd.export --safetensors models/voxtral/consolidated.safetensors --out models/voxtral-q4.gguf
"""
from __future__ import annotations

import json
import struct
import sys

import numpy as np

from . import synth


def read_safetensors(path: str):
    """{name: (shape, dtype_str, memmap view of the raw bytes)} -- the layout safetensors / models/weights.rs:170-205 read."""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    if buf.size < 8:
        raise ValueError("SafeTensors file too small")
    hlen = struct.unpack_from("<Q", buf, 0)[0]
    if hlen > buf.size - 8:
        raise ValueError("SafeTensors header length out of range")
    hdr = json.loads(bytes(buf[8:8 + hlen]).decode("utf-8"))
    out = {}
    for name, e in hdr.items():
        if name == "__metadata__":
            continue
        a, b = e["data_offsets"]
        ne = int(np.prod(e["shape"])) if e["shape"] else 1
        width = {"F32": 4, "F16": 2, "BF16": 2}.get(e["dtype"])
        if width is None:
            raise ValueError(f"Unsupported dtype {e['dtype']} for tensor '{name}'")          # weights.rs:60-64
        if b - a != ne * width or 8 + hlen + b > buf.size:
            raise ValueError(f"tensor '{name}' byte size does not match its shape")
        out[name] = (tuple(int(s) for s in e["shape"]), e["dtype"], buf[8 + hlen + a:8 + hlen + b])
    return out


def to_f32(raw: np.ndarray, dtype: str) -> np.ndarray:
    """weights.rs:16-66 load_tensor: F32 / F16 / BF16 -> f32."""
    if dtype == "F32":
        return np.asarray(raw).view(np.float32).copy()
    if dtype == "F16":
        return np.asarray(raw).view(np.float16).astype(np.float32)
    return synth.bf16_bits_to_f32(np.asarray(raw).view(np.uint16))


def quantize_q4_0_ggml(data: np.ndarray) -> np.ndarray:
    """llama.cpp quantize_row_q4_0_ref: per block of 32, max = the element of largest magnitude (signed), d = max / -8, id = 1/d,
    q = min(15, (int8)(v * id + 8.5)); element i -> low nibble of byte i, i + 16 -> high nibble."""
    x = np.ascontiguousarray(data, dtype=np.float32).reshape(-1, 32)
    idx = np.abs(x).argmax(axis=1)
    mx = x[np.arange(x.shape[0]), idx]
    d = (mx / np.float32(-8.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    q = np.clip(np.trunc(x * inv[:, None] + np.float32(8.5)), 0, 15).astype(np.uint8)
    out = np.empty((x.shape[0], 18), dtype=np.uint8)
    out[:, 0:2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
    return out.reshape(-1)


def is_q4_tensor(name: str, shape) -> bool:
    """What Q4ModelLoader loads through load_q4_linear (gguf/loader.rs:215-260, 305-383): every 2-D `.weight` that is not a norm."""
    return len(shape) == 2 and name.endswith(".weight") and "norm.weight" not in name and (shape[0] * shape[1]) % 32 == 0 and shape[1] % 32 == 0


def export_q4_gguf(st_path: str, out_path: str, scheme: str = "reference", log=None) -> dict:
    quant = {"reference": synth.quantize_q4_0, "ggml": quantize_q4_0_ggml}[scheme]
    tensors = read_safetensors(st_path)
    stats = {"q4": 0, "f32": 0, "q4_bytes": 0, "f32_bytes": 0}

    def make(name, shape, dtype, raw, q4):
        def gen():
            v = to_f32(raw, dtype)
            if log:
                log(f"  {name} {list(shape)} {dtype} -> {'Q4_0' if q4 else 'F32'}")
            return quant(v) if q4 else v
        return gen

    entries = []
    for name, (shape, dtype, raw) in tensors.items():
        q4 = is_q4_tensor(name, shape)
        ne = int(np.prod(shape))
        stats["q4" if q4 else "f32"] += 1; stats["q4_bytes" if q4 else "f32_bytes"] += ne // 32 * 18 if q4 else ne * 4
        entries.append((name, shape, synth.GGML_Q4_0 if q4 else synth.GGML_F32, make(name, shape, dtype, raw, q4)))
    synth.write_gguf(out_path, entries)
    return stats


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Quantise a Voxtral SafeTensors checkpoint to the Q4_0 GGUF the Q4 loader reads")
    ap.add_argument("--safetensors", required=True); ap.add_argument("--out", required=True)
    ap.add_argument("--scheme", choices=["reference", "ggml"], default="reference"); ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args(argv)
    st = export_q4_gguf(a.safetensors, a.out, a.scheme, log=(lambda s: print(s, file=sys.stderr)) if a.verbose else None)
    print(f"wrote {a.out}: {st['q4']} Q4_0 tensors ({st['q4_bytes'] / 1e9:.2f} GB), {st['f32']} F32 tensors ({st['f32_bytes'] / 1e6:.1f} MB)", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
