#!/usr/bin/env python3
"""Checks the multi-GPU start-up plumbing on one GPU: torch imported first (single HIP runtime in-process), the packed
weight arena copied into a torch tensor and back into a layout-only model, which must then transcribe identically."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
from model_fixtures import tiny_gguf, fake_mel
pkg = load_package(); torch.cuda.set_device(0); ctx = pkg.Context(0)
path, _ = tiny_gguf()
a = pkg.Q4ModelLoader.from_file(path).load(ctx)
b = pkg.Q4ModelLoader.from_file(path).load(ctx, layout_only=True)
pa, na = a.arena(); pb, nb = b.arena(); assert na == nb
stage = torch.empty(na, dtype=torch.uint8, device="cuda:0")
ctx.copy(stage.data_ptr(), pa, na); torch.cuda.synchronize()
chk = int(stage.to(torch.int64).sum().item())
ctx.copy(pb, stage.data_ptr(), nb)
b.arena_finalize()
t = pkg.TimeEmbedding(256).embed(6.0); mel = fake_mel(900, seed=4)
ia = a.transcribe_streaming(mel[None], t); ib = b.transcribe_streaming(mel[None], t)
assert len(ia) > 0 and (ia == ib).all(), (ia, ib)
print("torch_arena_check ok: arena", na, "bytes, checksum", chk, "ids", len(ia), "torch", torch.__version__, "hip", torch.version.hip)
