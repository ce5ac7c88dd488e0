"""Worker of tests/test_gpu_fullsize.py::test_full_load_replicated_rccl_world1: the product's multi-GPU start-up (shard.load_replicated) with a REAL RCCL process group of
one rank on one GPU -- torch imported first (one HIP runtime per process), `nccl` backend initialised, the broadcast issued on the arena memory itself -- then the same
arena pushed through a second, layout-only model the way a rank > 0 receives it (broadcast into ITS arena tensor from a clone of rank 0's bytes, then arena_finalize).
Prints one JSON line: ids of both models for the 16 s clip + broadcast stats."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2])
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pkg = load_package(); shard = pkg.shard
    ctx = pkg.Context(0)
    st = {}
    m = shard.load_replicated(pkg, ctx, path, 0, 1, local=0, group=dist.group.WORLD, stats=st)      # rank 0's side: full load + the (one-rank) RCCL broadcast on the arena
    # a receiver's side on the same GPU: layout-only model, its arena tensor filled by a broadcast (src = this rank: the payload is first copied in from rank 0's arena
    # tensor, as the wire would deliver it), then the derived copies rebuilt
    b = pkg.Q4ModelLoader.from_file(path).load(ctx, layout_only=True)
    pa, na = m.arena(); pb, nb = b.arena()
    ta = shard._arena_tensor(pa, na, 0); tb = shard._arena_tensor(pb, nb, 0)
    assert ta.data_ptr() == pa and tb.data_ptr() == pb and ta.numel() == na      # zero-copy views of the library's allocations
    tb.copy_(ta); torch.cuda.synchronize()
    dist.broadcast(tb, src=0); torch.cuda.synchronize()
    b.arena_finalize()
    x = pkg.synth.synth_audio(16.0, seed=1234); t = pkg.TimeEmbedding(m.config.dec_dim).embed(6.0)
    ids_a = m.transcribe_audio(x, t); ids_b = b.transcribe_audio(x, t)
    chk = int(ta[:: 4096].to(torch.int64).sum().item())
    print(json.dumps({"ids_a": [int(v) for v in ids_a], "ids_b": [int(v) for v in ids_b], "stats": st, "arena_bytes": int(na), "checksum": chk,
                      "backend": dist.get_backend(), "nccl_version": list(torch.cuda.nccl.version())}), flush=True)
    b.close(); m.close(); ctx.close()
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
