"""Multi-process (world_size 2, gloo, CPU) test of the utterance-sharding layer used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest


def test_lpt_partition(pkg):
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    rng = np.random.default_rng(7)
    dur = np.clip(rng.lognormal(np.log(10.0), 0.5, 647), 3, 30)      # FLEURS-like durations
    parts = shard.lpt_partition(dur, 8)
    assert sorted(i for p in parts for i in p) == list(range(647))
    assert shard.imbalance(dur, parts) < 0.02                         # < 2 % LPT imbalance over 8 ranks
    assert shard.lpt_partition([], 4) == [[], [], [], []]
    assert shard.lpt_partition([5.0], 2) == [[0], []]
    assert shard.lpt_partition(dur, 8) == parts                       # deterministic
    out = shard.run_sharded(list(range(5)), [1] * 5, lambda x: x * x, 0, 1)
    assert out == [0, 1, 4, 9, 16]
    # length-bucketed batches (vox_transcribe_batch groups): every index once, batches <= 16, neighbours in length together
    b = shard.length_buckets(parts[0], dur, 16)
    assert sorted(i for g in b for i in g) == parts[0] and max(len(g) for g in b) <= 16
    assert all(min(dur[i] for i in b[k]) >= max(dur[i] for i in b[k + 1]) for k in range(len(b) - 1))
    calls = []
    out = shard.run_sharded(list(range(7)), [3, 1, 2, 9, 5, 4, 8], None, 0, 1, batch=3, batch_work=lambda xs: (calls.append(list(xs)), [x * 10 for x in xs])[1])
    assert out == [0, 10, 20, 30, 40, 50, 60] and calls == [[3, 6, 4], [5, 0], [2, 1]]           # balanced: 3 + 2 + 2, not 3 + 3 + 1


def _worker(rank, world, port, pkg_dir, q):
    import importlib.util, sys
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("shard", os.path.join(pkg_dir, "shard.py"))
    shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
    items = [f"utt{i}" for i in range(11)]
    costs = [3.0 + (i * 7) % 11 for i in range(11)]
    seen = []

    def work(name):
        seen.append(name)
        return f"{name}:rank{rank}"

    out = shard.run_sharded(items, costs, work, rank, world)
    outb = shard.run_sharded(items, costs, None, rank, world, batch=4, batch_work=lambda names: [f"{n}:rank{rank}" for n in names])
    assert (outb is None) == (out is None) and (out is None or [o.split(":")[0] for o in outb] == items)
    # bench.py-style timing reduction: barrier, max over ranks
    import torch
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, out, seen, float(t.item())))
    dist.barrier(); dist.destroy_process_group()


def test_run_sharded_world2_gloo(pkg):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    pkg_dir = os.path.dirname(pkg.__file__)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pkg_dir, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, seen, tmax = q.get(timeout=180)
        res[rank] = (out, seen, tmax)
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    out0, seen0, t0 = res[0]; out1, seen1, t1 = res[1]
    assert out1 is None and len(out0) == 11
    assert [o.split(":")[0] for o in out0] == [f"utt{i}" for i in range(11)]        # input order preserved
    assert sorted(seen0 + seen1) == sorted(f"utt{i}" for i in range(11)) and not set(seen0) & set(seen1)
    assert t0 == t1 == 2.0                                                            # max over ranks
    for o in out0:
        name, rk = o.split(":rank")
        assert name in (seen0 if rk == "0" else seen1)


def test_spawn_ranks_world2_gloo(pkg, tmp_path):
    """The rank launcher behind `bench.py --gpus N` (shard.spawn_ranks -> torch.distributed.run, rendezvous on 127.0.0.1), world 2 on CPU."""
    pytest.importorskip("torch")
    import json, sys
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    out = tmp_path / "r0.json"
    here = os.path.dirname(os.path.abspath(__file__))
    rc = shard.spawn_ranks(2, os.path.join(here, "rank_worker.py"), [str(out), os.path.dirname(pkg.__file__)])
    assert rc == 0
    r = json.loads(out.read_text())
    assert r["world"] == 2 and r["n"] == 37 and r["order_ok"] and r["ranks_used"] == [0, 1] and r["imbalance"] < 0.1
    assert r["t_max"] >= 1.0                                   # max over ranks (rank 1 adds 1 s to its time)
    d = shard.fleurs_like_durations()
    assert len(d) == 647 and min(d) >= 3.0 and max(d) <= 30.0 and 9.0 < sorted(d)[323] < 11.0 and d == shard.fleurs_like_durations()


def test_bench_gpus_flag_fails_loudly_without_enough_gpus():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs must refuse (exit code 2, message), not run one rank silently."""
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() >= 2:
        pytest.skip("2+ GPUs present")
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "refusing to run" in p.stderr and p.stdout.strip() == ""
    env["WORLD_SIZE"] = "1"; env["RANK"] = "0"; env["LOCAL_RANK"] = "0"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "does not match --gpus" in p.stderr


def _cli_worker(rank, world, port, pkg_dir, tmp, q):
    import importlib.util, os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(pkg_dir))
    from __graft_entry__ import load_package
    pkg = load_package()
    cli = importlib.import_module(pkg.__name__ + ".cli")
    paths = [os.path.join(tmp, f"f{i}.bin") for i in range(11)]
    done = []
    def one(i):                                  # the fake model: the line names the file and the rank that "transcribed" it
        done.append(i); return f"file{i}@rank{rank}"
    lines = cli.sharded_lines(pkg, paths, one, rank, world)
    q.put((rank, lines, done))


def test_cli_sharded_lines_world2_gloo(pkg, tmp_path):
    """`voxtral-transcribe --gpus N` data path on CPU (gloo, world 2, fake model): every file transcribed exactly once, by the rank the longest-first
    partition of the file sizes names, and rank 0 gets one line per input IN INPUT ORDER (the reference's stdout contract, bin/transcribe.rs:112-126)."""
    import multiprocessing as mp, os
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    sizes = [500, 100, 900, 300, 300, 800, 50, 700, 200, 600, 400]
    for i, n in enumerate(sizes):
        (tmp_path / f"f{i}.bin").write_bytes(b"x" * n)
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = shard.free_port()
    ps = [ctx.Process(target=_cli_worker, args=(r, 2, port, os.path.dirname(os.path.abspath(pkg.__file__)), str(tmp_path), q)) for r in range(2)]
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        r, lines, done = q.get(timeout=120); res[r] = (lines, done)
    [p.join(60) for p in ps]
    parts = shard.lpt_partition([float(n) for n in sizes], 2)
    assert res[1][0] is None and sorted(res[0][1]) == parts[0] and sorted(res[1][1]) == parts[1]
    owner = {i: r for r in range(2) for i in parts[r]}
    assert res[0][0] == [f"file{i}@rank{owner[i]}" for i in range(11)]


def _cli_units_worker(rank, world, port, pkg_dir, q):
    import importlib, os, sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(pkg_dir))
    from __graft_entry__ import load_package
    pkg = load_package()
    cli = importlib.import_module(pkg.__name__ + ".cli")
    # the unit table every rank derives from the headers: file 0 = three chunks, file 1 = one unit, file 2 = two chunks, file 3 = one unit
    units = [(0, 0, 0, 192000), (0, 1, 192000, 384000), (0, 2, 384000, 480000), (1, 0, 0, 80000), (2, 0, 0, 192000), (2, 1, 192000, 200000), (3, 0, 0, 150000)]
    calls = []
    def runner(us):                              # the fake model: a unit's text names the unit and the rank; the tail chunk of file 2 decodes to nothing; file 3 fails
        calls.append(list(us))
        return [None if i == 3 else ("" if (i, k) == (2, 1) else f"f{i}c{k}@{rank}") for i, k, _, _ in us]
    texts = cli.sharded_units(pkg, units, runner, 1024, rank, world)
    lines = cli.join_units(4, units, texts) if texts is not None else None
    q.put((rank, lines, calls))


def test_cli_sharded_units_world2_gloo(pkg):
    """`voxtral-transcribe --batch N --gpus 2` data path on CPU (gloo, world 2, fake model): CHUNKS are the units of work (bin/transcribe.rs:210-265) -- partitioned
    longest-first by their length over the ranks (a file's chunks may sit on different ranks), every rank hands its share to the runner in one call, rank 0 re-joins
    every file's non-empty chunk texts with one space (:261-275); a file with a failed unit is left to the one-by-one path."""
    import multiprocessing as mp, os
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = shard.free_port()
    ps = [ctx.Process(target=_cli_units_worker, args=(r, 2, port, os.path.dirname(os.path.abspath(pkg.__file__)), q)) for r in range(2)]
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        r, lines, calls = q.get(timeout=120); res[r] = (lines, calls)
    [p.join(60) for p in ps]
    costs = [192000.0, 192000.0, 96000.0, 80000.0, 192000.0, 8000.0, 150000.0]
    parts = shard.lpt_partition(costs, 2)
    assert res[1][0] is None and all(len(res[r][1]) == 1 and len(res[r][1][0]) == len(parts[r]) for r in range(2))      # ONE runner call per rank with its whole share
    own = {u: r for r in range(2) for u in parts[r]}
    assert res[0][0] == {0: f"f0c0@{own[0]} f0c1@{own[1]} f0c2@{own[2]}", 1: f"f1c0@{own[3]}", 2: f"f2c0@{own[4]}"}      # file 3 (failed unit) absent -> one-by-one path
    assert len({own[0], own[1], own[2]}) == 2                                                                             # file 0's chunks really were split over both ranks


def test_length_buckets_are_balanced(pkg):
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    costs = [float(i % 17) for i in range(81)]
    b = shard.length_buckets(list(range(81)), costs, 64)
    assert [len(x) for x in b] == [41, 40] and sorted(sum(b, [])) == list(range(81))          # not 64 + 17
    assert all(costs[b[0][k]] >= costs[b[0][k + 1]] for k in range(40))                      # still sorted by length
    assert shard.length_buckets([], costs, 64) == [] and [len(x) for x in shard.length_buckets(list(range(64)), costs, 64)] == [64]
    assert [len(x) for x in shard.length_buckets(list(range(130)), [1.0] * 130, 64)] == [44, 43, 43]


def _replicated_worker(rank, world, port, pkg_dir, q):
    """Fake loader / model / context with a HOST arena: the start-up protocol of shard.load_replicated without a GPU."""
    import importlib.util
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("shard", os.path.join(pkg_dir, "shard.py"))
    shard = importlib.util.module_from_spec(spec); spec.loader.exec_module(shard)
    events = []

    class FakeModel:
        def __init__(self, layout_only):
            self.buf = np.zeros(4096, np.uint8) if layout_only else (np.arange(4096) % 251).astype(np.uint8)      # rank 0 "parsed the file"
            self.finalized = False

        def arena(self):
            return self.buf.ctypes.data, self.buf.nbytes

        def arena_finalize(self):
            events.append("finalize"); self.finalized = True

    class FakeLoader:
        def load(self, ctx, layout_only=False):
            events.append("layout_only" if layout_only else "full_load"); self.m = FakeModel(layout_only); return self.m

    class FakeCtx:
        def synchronize(self):
            events.append("sync")

    ld = FakeLoader(); st = {}
    m = shard.load_replicated(None, FakeCtx(), "unused.gguf", rank, world, loader=ld, stats=st,
                              as_tensor=lambda p, n: torch.from_numpy(np.ctypeslib.as_array((__import__("ctypes").c_uint8 * n).from_address(p))))
    q.put((rank, events, int(m.buf.astype(np.int64).sum()), m.finalized, st))
    dist.barrier(); dist.destroy_process_group()


def test_load_replicated_protocol_world2_gloo(pkg):
    """shard.load_replicated (the multi-GPU start-up of cli.py / wer.py / bench.py --gpus N) on CPU, world 2, gloo, with a fake loader whose arena is host memory:
    rank 0 loads the file, rank 1 only lays the arena out; ONE broadcast issued on the arena memory itself (no staging buffer) makes rank 1's arena equal to rank 0's;
    only the receiver rebuilds the derived copies; the library stream is synchronised before the collective.  Without a process group (world 1) it is a plain load."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    shard = __import__("importlib").import_module(pkg.__name__ + ".shard")
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = shard.free_port()
    ps = [ctx.Process(target=_replicated_worker, args=(r, 2, port, os.path.dirname(pkg.__file__), q)) for r in range(2)]
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        rank, events, chk, fin, st = q.get(timeout=180); res[rank] = (events, chk, fin, st)
    for p in ps:
        p.join(timeout=60); assert p.exitcode == 0
    want = int((np.arange(4096) % 251).sum())
    assert res[0][0] == ["full_load", "sync"] and res[1][0] == ["layout_only", "sync", "finalize"]
    assert res[0][1] == want and res[1][1] == want and res[1][2] and not res[0][2]
    assert res[0][3]["bytes"] == 4096 and res[0][3]["broadcast"] and res[1][3]["broadcast"]

    class L:
        def load(self, ctx, layout_only=False):
            assert not layout_only; return "model"
    st = {}
    assert shard.load_replicated(None, None, "x", 0, 1, loader=L(), stats=st) == "model" and st == {"bytes": 0, "seconds": 0.0, "broadcast": False}
    with pytest.raises(RuntimeError):
        shard.load_replicated(None, None, "x", 0, 2, loader=L())


def test_session_pool_split_and_threads_cpu(pkg):
    """shard.SessionPool without a GPU: fake models (ids = the unit's length) on fake contexts.  Units sharing a normalisation group stay in one session (the group's peak
    is reduced over the units of ONE call), the split is balanced, results come back in input order, the pool's contexts are marked shared while it lives and the caller's context is un-marked
    by close(), a failing session raises on the calling thread."""
    import importlib, threading
    shard = importlib.import_module(pkg.__name__ + ".shard")
    calls = []

    class Ctx:
        device = 0
        shared = False
        def set_shared(self, on=True): self.shared = bool(on)
        def synchronize(self): pass
        def close(self): pass

    class Model:
        def __init__(self, fail=False): self.eng, self.fail, self.closed = True, fail, False
        def set_batch_engine(self, on=None):
            if on is not None: self.eng = bool(on)
            return self.eng, 0
        def replicate(self, c): return Model(self.fail)
        def close(self): self.closed = True
        def transcribe_batch(self, xs, t, norm_group=None):
            if self.fail and threading.current_thread() is not threading.main_thread(): raise RuntimeError("boom")
            calls.append((threading.current_thread().name, len(xs), None if norm_group is None else list(norm_group)))
            return [np.full(2, len(x), np.int32) for x in xs]

    class Pkg:
        Context = staticmethod(lambda dev: Ctx())

    m = Model(); c0 = Ctx(); pool = shard.SessionPool(Pkg, c0, m, 2)
    assert len(pool.models) == 2 and all(c.shared for c in pool.ctxs)      # every context of the pool is marked shared (vox_ctx_set_shared: no batched engines, planner on the table)
    assert pool.MIN_UNITS_PER_SESSION >= 64      # small shares run as one session (measured: two sessions only pay off from ~250 units on)
    pool.MIN_UNITS_PER_SESSION = 2
    xs = [np.zeros(n, np.float32) for n in (50, 30, 90, 10, 40, 40, 70, 20)]
    grp = [0, 0, 1, -1, 2, 2, 3, -1]
    parts = pool.split([float(x.size) for x in xs], grp)
    assert sorted(parts[0] + parts[1]) == list(range(8))
    for g in (0, 2):
        owners = {k for k in range(2) for u in parts[k] if grp[u] == g}
        assert len(owners) == 1
    loads = [sum(xs[u].size for u in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 40
    out = pool.transcribe_batch(xs, None, norm_group=grp)
    assert [int(o[0]) for o in out] == [x.size for x in xs]
    assert len({c[0] for c in calls}) == 2 and sorted(c[1] for c in calls) == sorted(len(p) for p in parts)       # two host threads, one call each
    assert all(c[2] is not None and len(c[2]) == c[1] for c in calls)
    calls.clear(); out = pool.transcribe_batch(xs[:3], None)      # fewer than 2 units per session: one plain call
    assert len(calls) == 1 and calls[0][1] == 3 and [int(o[0]) for o in out] == [50, 30, 90]
    rep = pool.models[1]; pool.close()
    assert c0.shared is False and rep.closed and len(pool.models) == 1
    bad = shard.SessionPool(Pkg, Ctx(), Model(fail=True), 2); bad.MIN_UNITS_PER_SESSION = 2
    with pytest.raises(RuntimeError, match="boom"):
        bad.transcribe_batch(xs, None)
    bad.close()
