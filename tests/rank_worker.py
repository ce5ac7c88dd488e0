"""Worker script for tests/test_shard_gloo.py::test_spawn_ranks_world2_gloo: launched as N ranks by shard.spawn_ranks (torch.distributed.run,
the launch the driver uses for `bench.py --gpus N`); CPU only (gloo).  Runs the bench-style flow on fake work: init from the environment,
LPT-sharded work, barrier, max-over-ranks timing, rank 0 writes one JSON line to the file named on the command line."""
import importlib.util, json, os, sys, time

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo")
spec = importlib.util.spec_from_file_location("shard", os.path.join(sys.argv[2], "shard.py"))
shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
durs = shard.fleurs_like_durations(37, seed=7)
parts = shard.lpt_partition(durs, world)
dist.barrier(); t0 = time.perf_counter()
res = shard.run_sharded(list(range(37)), durs, None, rank, world, batch=4, batch_work=lambda idx: [(i, rank) for i in idx])
dt = torch.tensor([time.perf_counter() - t0 + rank], dtype=torch.float64)
dist.all_reduce(dt, op=dist.ReduceOp.MAX)
if rank == 0:
    with open(sys.argv[1], "w") as f:
        json.dump({"world": world, "n": len(res), "order_ok": [r[0] for r in res] == list(range(37)), "ranks_used": sorted({r[1] for r in res}),
                   "imbalance": shard.imbalance(durs, parts), "t_max": float(dt.item())}, f)
dist.barrier(); dist.destroy_process_group()
