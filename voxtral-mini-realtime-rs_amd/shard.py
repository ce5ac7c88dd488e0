"""Multi-GPU sharding of independent utterances: replicas only, one process per GPU,
no collective inside encode/decode.  Units of work = utterances (or CLI chunks, which are independent too,
bin/transcribe.rs:231-265).  Longest-processing-time-first assignment by audio duration (decode cost is
proportional to S-38 ~ 6.25 * seconds because the reference has no early stop), results gathered by input
index so the stdout order of `voxtral-transcribe` (one line per input, transcribe.rs:112-126) is preserved.

torch.distributed is imported lazily so the single-GPU path never needs it."""
from __future__ import annotations

from typing import Callable, Sequence


def lpt_partition(costs: Sequence[float], world: int) -> list[list[int]]:
    """Greedy LPT: sort by cost descending, give each item to the least-loaded rank. Deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world
    parts: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        parts[r].append(i); loads[r] += float(costs[i])
    for p in parts:
        p.sort()
    return parts


def imbalance(costs: Sequence[float], parts: list[list[int]]) -> float:
    loads = [sum(float(costs[i]) for i in p) for p in parts]
    mean = sum(loads) / max(len(loads), 1)
    return (max(loads) / mean - 1.0) if mean > 0 else 0.0


def length_buckets(idx: Sequence[int], costs: Sequence[float], batch: int) -> list[list[int]]:
    """Group a rank's share into batches of <= `batch` utterances of similar length (sorted by cost): the stacked encoder and
    the batched decode loop of vox_transcribe_batch pay for the longest member of a batch, so neighbours in length go together."""
    order = sorted(idx, key=lambda i: (-float(costs[i]), i))
    n, b = len(order), max(int(batch), 1)
    if n == 0:
        return []
    k = (n + b - 1) // b                       # number of batches; sizes differ by at most one (81 clips, batch 64 -> 41 + 40, not 64 + 17:
    base, extra = divmod(n, k)                 # a 17-clip tail batch runs at a fraction of a 64-clip batch's throughput)
    out, pos = [], 0
    for j in range(k):
        sz = base + (1 if j < extra else 0)
        out.append(order[pos:pos + sz]); pos += sz
    return out


def run_sharded(items: Sequence, costs: Sequence[float], work: Callable, rank: int, world: int, group=None,
                batch: int = 1, batch_work: Callable | None = None):
    """Every rank calls this with the same `items`/`costs`; rank r runs `work(item)` on its LPT share -- or, with `batch > 1`,
    `batch_work(list_of_items) -> list_of_results` on length-bucketed groups of its share (one vox_transcribe_batch call each).
    Returns the full, index-ordered result list on rank 0 (None elsewhere)."""
    parts = lpt_partition(costs, world)
    if batch > 1 and batch_work is not None:
        mine = []
        for grp in length_buckets(parts[rank], costs, batch):
            res = batch_work([items[i] for i in grp])
            if len(res) != len(grp):
                raise ValueError("batch_work must return one result per item")
            mine.extend(zip(grp, res))
    else:
        mine = [(i, work(items[i])) for i in parts[rank]]
    if world == 1:
        out = [None] * len(items)
        for i, r in mine:
            out[i] = r
        return out
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * len(items)
    for part in gathered:
        for i, r in part:
            out[i] = r
    return out


class _DevSpan:
    """A raw device allocation as a __cuda_array_interface__ object, so torch can wrap it without a copy (`torch.as_tensor(span, device=...)`)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _arena_tensor(ptr: int, nbytes: int, local: int):
    import torch
    return torch.as_tensor(_DevSpan(ptr, nbytes), device=f"cuda:{local}")


def load_replicated(pkg, ctx, path, rank: int, world: int, local: int | None = None, group=None, loader=None, as_tensor=None, stats: dict | None = None):
    """Multi-GPU start-up of the replicas (replaces N independent `Q4ModelLoader::from_file(..).load()` calls, bin/transcribe.rs:88-100):
    rank 0 parses + repacks the GGUF; every other rank only lays the device arena out (VOX_LOAD_LAYOUT_ONLY: header parse, no tensor data read, no upload); the
    PRIMARY part of the arena (everything parsed from the file: 2.5 GB for the Q4 model) travels with ONE `torch.distributed.broadcast` -- RCCL over xGMI on GPUs --
    issued directly on the arena memory (wrapped as a torch tensor through __cuda_array_interface__, no staging copy); each receiver then rebuilds the derived copies on
    its own GPU (vox_model_arena_finalize; the decode engines' weight stream is packed at the first decode step on every rank).  No collective ever enters the data path.
    world == 1 with an initialised process group still issues the broadcast (a one-rank RCCL broadcast: the single-GPU test of this plumbing).
    `loader` / `as_tensor` are injection points for the CPU (gloo) test of the protocol.  Returns the model; `stats` receives bytes / seconds of the broadcast."""
    import time
    loader = loader or pkg.Q4ModelLoader.from_file(path)
    dist = None
    if world > 1 or group is not None:
        import torch.distributed as dist_
        dist = dist_ if dist_.is_initialized() else None
        if world > 1 and dist is None:
            raise RuntimeError("load_replicated: world > 1 needs an initialised torch.distributed process group")
    if dist is None:
        if stats is not None:
            stats.update(bytes=0, seconds=0.0, broadcast=False)
        return loader.load(ctx)
    local = rank if local is None else local
    model = loader.load(ctx, layout_only=(rank != 0))
    ptr, nbytes = model.arena()
    t = (as_tensor or (lambda p, n: _arena_tensor(p, n, local)))(ptr, nbytes)
    # rank 0's uploads / repack kernels ran on the LIBRARY's stream, the collective runs on torch's current stream: order the two on the device (torch's stream waits
    # for an event recorded on the library's stream, wrapped as an ExternalStream) instead of blocking the host on the whole context; a host synchronisation remains
    # only where there is no device stream to order against (the CPU / gloo test of this protocol)
    ordered = False
    if getattr(t, "is_cuda", False) and hasattr(ctx, "stream"):
        try:
            import torch
            ext = torch.cuda.ExternalStream(int(ctx.stream()), device=t.device)
            torch.cuda.current_stream(t.device).wait_stream(ext); ordered = True
        except Exception:
            ordered = False
    if not ordered:
        ctx.synchronize()
    t0 = time.time()
    dist.broadcast(t, src=0, group=group)
    if t.is_cuda:
        import torch
        torch.cuda.synchronize(t.device)
    dt = time.time() - t0
    if rank != 0:
        model.arena_finalize()
    if stats is not None:
        stats.update(bytes=int(nbytes), seconds=dt, broadcast=True, ordered_by_event=ordered, world_size=dist.get_world_size(group), backend=str(dist.get_backend(group)))
    return model


def free_port() -> int:
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def spawn_ranks(n: int, script: str, argv: Sequence[str], env: dict | None = None, port: int | None = None, capture: bool = False):
    """Run `script argv` as `n` ranks of ONE node through torch.distributed.run (rendezvous on 127.0.0.1: the container hostname may
    not resolve) -- the launch the driver uses for `bench.py --gpus N`.  Returns the launcher's exit code (with `capture`: (exit code, stdout)).  The script reads
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port()), script, *argv]
    if capture:                                # (rc, stdout text): callers that re-emit rank 0's stdout through their own sys.stdout
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
        return r.returncode, r.stdout
    return subprocess.call(cmd, env=env)


def fleurs_like_durations(n: int = 647, seed: int = 7):
    """SURVEY.md section 8(d) config 5 stand-in for FLEURS-en test (647 utterances): log-normal durations, median 10 s, clipped to
    3..30 s, numpy default_rng(seed).  Returns a list of seconds (rounded to 10 ms so sample counts are exact)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return [round(float(d), 2) for d in np.clip(rng.lognormal(np.log(10.0), 0.5, n), 3.0, 30.0)]


class SessionPool:
    """Overlap INSIDE a rank's share: S concurrent sessions on ONE GPU -- S contexts (own stream, workspaces, graphs), S model replicas
    (vox_model_replicate: 2.5 GB each, device-to-device, no file), S host threads -- each running vox_transcribe_batch[_ex] over its part of the units.  One session
    leaves the GPU idle wherever its launch-bound decode steps wait (a 64-slot step streams 2.1 GB of weights in 3.1 ms); a second session's encoder GEMMs, prefill
    and steps fill those gaps: the 647-clip FLEURS-like corpus takes 3.73 s with two sessions against 4.33 s with one (profiles/r06_bench_n1.json; tools/two_sessions_probe.py, same ids).
    What a Rust host would do with one thread per (context, model) pair; results are per unit and do not depend on the split (every row of a batch is computed
    independently, tests/test_gpu_fullsize.py::test_full_two_sessions_one_gpu_same_ids).

    Every context of a pool with S > 1 is marked SHARED (vox_ctx_set_shared): the batch entry points stay off the batched decode engines -- their 256 persistent
    workgroups need the GPU to themselves, a second session makes their bounded hand-off waits expire and every strike is a session run twice -- and the slot planner
    prices its steps with the scaled table instead of per-form measurements that scatter under contention (one bad figure planned 48 slots instead of 64: 4.7 s
    instead of 3.8 s)."""

    # Fewer units than this per session and the call runs as ONE session: a small share is bound by its longest clip's step count, and a second session only slows
    # every step (same box, two sessions against one: 81 clips x0.82, 162 x0.86, 324 x1.16, 480 x1.20, 647 x1.20 -- tools/two_sessions_probe.py)
    MIN_UNITS_PER_SESSION = 128

    def __init__(self, pkg, ctx, model, sessions: int = 2):
        self.pkg, self.sessions = pkg, max(1, int(sessions))
        self.ctxs, self.models, self._own = [ctx], [model], []
        if self.sessions > 1:
            for _ in range(self.sessions - 1):
                c = pkg.Context(ctx.device); r = model.replicate(c)
                self.ctxs.append(c); self.models.append(r); self._own.append((r, c))
            for c in self.ctxs:
                c.set_shared(True)

    def split(self, weights: Sequence[float], groups: Sequence[int] | None = None, sessions: int | None = None) -> list[list[int]]:
        """Unit indices per session: LPT over the units -- or over whole normalisation groups (units that share a group id >= 0 stay in one session: the group's
        peak is reduced on the device over the units of ONE call)."""
        S = self.sessions if sessions is None else sessions
        if groups is None or all(g < 0 for g in groups):
            return [sorted(p) for p in lpt_partition(list(weights), S)]
        keys, members = {}, []
        for u, g in enumerate(groups):
            k = ("g", g) if g >= 0 else ("u", u)
            if k not in keys:
                keys[k] = len(members); members.append([])
            members[keys[k]].append(u)
        parts = lpt_partition([sum(weights[u] for u in mem) for mem in members], S)
        return [sorted(u for gi in p for u in members[gi]) for p in parts]

    def transcribe_batch(self, samples_list, t_embed, norm_group=None):
        """Same contract as Q4VoxtralModel.transcribe_batch: ids per unit, in input order."""
        n = len(samples_list)
        S = max(1, min(self.sessions, n // max(1, self.MIN_UNITS_PER_SESSION)))
        if S == 1:
            return self.models[0].transcribe_batch(samples_list, t_embed, norm_group=norm_group)
        import threading
        parts = self.split([float(len(x)) for x in samples_list], norm_group, S)
        res, errs = [None] * S, []

        def work(k):
            try:
                idx = parts[k]
                if idx:
                    res[k] = self.models[k].transcribe_batch([samples_list[i] for i in idx], t_embed, norm_group=None if norm_group is None else [norm_group[i] for i in idx])
                    self.ctxs[k].synchronize()
            except Exception as e:      # noqa: BLE001 -- re-raised on the calling thread
                errs.append(e)
        th = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        out = [None] * n
        for k in range(S):
            for i, o in zip(parts[k], res[k] or []):
                out[i] = o
        return out

    def close(self):
        for r, c in self._own:
            r.close(); c.close()
        if self._own:
            self.ctxs[0].set_shared(False)
        self._own = []
        self.ctxs, self.models = self.ctxs[:1], self.models[:1]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
