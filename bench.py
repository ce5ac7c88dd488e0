#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Voxtral-Mini realtime hot path on MI355X.

Metric (BASELINE.json): RTF + decode tok/s on 16 s / 16 kHz clips, Q4 path.
A "step" = one whole pass of the hot path over one synthetic 16 s clip that is already resident in HBM:
peak-normalise -> pad -> log-mel -> 32-layer encoder -> adapter -> 38-token prefill -> 107 greedy decode
steps -> 108 token ids read back (the reference's un-chunked `e2e-bench` pipeline, bin/e2e_bench.rs:138-254;
workload = BASELINE.json configs[2] "Single 16 s WAV, Q4_0 GGUF on 1xMI355X").  Weights are synthetic
random Q4_0 in the real GGUF layout (no checkpoint is available offline); token count is a pure function
of audio length (no EOS), so throughput does not depend on weight values.

    python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU.  Launched by the driver through torch.distributed.run (RANK / WORLD_SIZE in the environment), or -- when
WORLD_SIZE is not set -- bench.py re-executes ITSELF through the same launcher (shard.spawn_ranks) after checking that the box has N
GPUs; it fails loudly otherwise.  Independent utterances per rank ("weak" scaling: every rank transcribes its own clip each step),
rank 0 parses the GGUF and the packed weight arena reaches the other ranks through ONE RCCL broadcast; no collective in the data path.
Rank 0 prints ONE JSON line.  Extras in the same line (never `value`):
  `batch`       BASELINE configs[3]: 16 clips through vox_transcribe_batch (N = 1)
  `f32`         BASELINE configs[1]: the same clip through the f32 SafeTensors path (dense bf16 weights, N = 1)
  `fleurs_like` BASELINE configs[4] stand-in: 647 clips with FLEURS-like durations sharded LPT over the ranks, each rank's share in ONE vox_transcribe_batch call (continuous batching; --fleurs-batch)
                (replaces bin/transcribe.rs:112-126's serial loop); aggregate RTF, tok/s, LPT imbalance (every N)
  `fleurs_like_cli`  the same corpus on the reference CLI's semantics (bin/transcribe.rs:207-265, --max-mel-frames 1200): file normalised once, 1200-frame chunks as units (every N)
  `streaming_encoder`  per-chunk latency of the streaming encoder (gguf/model.rs:437-459) for 1 s and 12 s chunks, N = 1
  `piecewise`   the reference's metric loop (bin/e2e_bench.rs:179-224) call for call through the C ABI from C (tools/e2e_piecewise.c), N = 1
  `roofline`    dominant decode kernel, HIP events on the library stream + committed PMC traffic;  `cpu_baseline`  CPU oracle, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PREFIX_LEN = 38
REF_TOK_S = 19.4   # reference Q4 native decode tok/s (DGX Spark GB10), README.md:14
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def full_gguf_path(pkg, seed, rank, barrier):
    path = os.path.join(os.environ.get("VOX_BENCH_DIR", "/tmp"), f"vox_bench_full_q4_seed{seed}.gguf")
    if rank == 0 and not os.path.exists(path):
        t0 = time.time()
        tmp = path + ".tmp"
        pkg.synth.write_synthetic_gguf(tmp, pkg.synth.ModelDims(), seed=seed)
        os.replace(tmp, path)
        log(f"[bench] wrote synthetic full-size Q4_0 GGUF {path} ({os.path.getsize(path) / 1e9:.2f} GB) in {time.time() - t0:.1f}s")
    barrier()
    return path


def cpu_baseline(pkg, gguf_path, seconds):
    """CPU restatement of the reference path (oracle, kind 'port') on this box's host cores, on a bounded
    sample: the full pipeline on a `seconds`-long clip (same pad/mel/encoder/decoder, fewer positions)."""
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 8, 64)))   # threads actually used are reported as `cores`
    import oracle_lib as orc
    orc.build()
    m = orc.Model(gguf_path)
    x = pkg.synth.synth_audio(seconds, seed=1234)
    t0 = time.time()
    xn = x.copy(); orc.lib().orc_peak_normalize(xn, xn.size, 0.95)
    mel = np.ascontiguousarray(orc.mel_compute_log(orc.pad_audio(xn)).T)
    pre = time.time() - t0
    t_embed = orc.time_embedding(6.0, m.cfg.dec_dim)
    ids = m.transcribe_streaming(mel, t_embed)
    total = time.time() - t0
    enc_ms, dec_ms = m.timings()
    m.close()
    n = int(len(ids))
    return {"value": round(n / (dec_ms / 1e3), 3) if dec_ms > 0 else None, "unit": "tok/s", "cores": orc.lib().orc_num_threads(),
            "kind": "port",
            "sample": f"{seconds:g} s synthetic clip, full pipeline (mel {pre * 1e3:.0f} ms, encode {enc_ms:.0f} ms, "
                      f"prefill+{max(n - 1, 0)} decode steps {dec_ms:.0f} ms, {n} ids); value = ids / decode_s as bin/e2e_bench.rs:236-240",
            "rtf": round(total / seconds, 3), "total_s": round(total, 2)}


def f32_extra(pkg, ctx, t_embed, seconds, reps=3):
    """BASELINE configs[1]: the bench clip through the f32 SafeTensors path (VoxtralModelLoader; bf16 checkpoint exact on device,
    f32 activations and accumulation).  Parity of this exact model + clip: tests/test_gpu_fullsize.py::test_full_16s_clip_f32_vs_oracle_golden."""
    from model_fixtures import full_dense_safetensors
    t0 = time.time(); st = full_dense_safetensors(7); t_gen = time.time() - t0
    t0 = time.time(); m = pkg.VoxtralModelLoader.from_file(st).load(ctx); t_load = time.time() - t0
    x = pkg.synth.synth_audio(seconds, seed=1234); dx = ctx.upload(x)
    ids = m.transcribe_audio(None, t_embed, device_ptr=dx, n_samples=x.size)      # warm-up (graph capture)
    ctx.synchronize(); tb = time.perf_counter(); st_ms = {"preprocess_ms": 0.0, "encode_ms": 0.0, "decode_ms": 0.0}
    for _ in range(reps):
        ids = m.transcribe_audio(None, t_embed, device_ptr=dx, n_samples=x.size)
        tm = m.timings()
        for k in st_ms:
            st_ms[k] += tm[k] / reps
    ctx.synchronize(); dt = (time.perf_counter() - tb) / reps
    wb = m.weight_bytes(); cf = m.config; ctx.free(dx); m.close()
    n = len(ids)
    # bytes one decode step streams: the decoder's linears + the tied lm_head as bf16 (NOT the whole arena: the encoder's weights are not read by a decode step)
    qd, kd = cf.dec_heads * cf.dec_head_dim, cf.dec_kv_heads * cf.dec_head_dim
    dec_bytes = 2 * (cf.dec_layers * ((qd + 2 * kd) * cf.dec_dim + cf.dec_dim * qd + 3 * cf.dec_ffn * cf.dec_dim) + cf.vocab * cf.dec_dim)
    step_s = (st_ms["decode_ms"] / 1e3 / max(n, 1)) if st_ms["decode_ms"] > 0 else None
    return {"workload": f"single {seconds:g} s clip, f32 SafeTensors path (BASELINE configs[1]): synthetic BF16 checkpoint, dense bf16 weights on device, f32 arithmetic",
            "tok_per_s": round(n / dt, 1), "ms_per_clip": round(dt * 1e3, 2), "rtf": round(dt / seconds, 5), "ids_per_clip": n,
            "decode_tok_per_s_ref_def": round(n / (st_ms["decode_ms"] / 1e3), 1), "stage_ms": {k: round(v, 3) for k, v in st_ms.items()},
            "weight_bytes": wb, "decode_step_bytes": dec_bytes, "decode_step_weight_GBps": round(dec_bytes / 1e9 / step_s, 1) if step_s else None,
            "decode_step_frac_of_hbm_peak": round(dec_bytes / 1e9 / step_s / HBM_PEAK_GBS, 4) if step_s else None,
            "checkpoint_write_s": round(t_gen, 1), "load_s": round(t_load, 1)}


def streaming_encoder_extra(pkg, ctx, model, chunk_frames=(100, 1200), seconds=120.0):
    """SURVEY section 8(f2) -- the "realtime" in the reference's name: Q4VoxtralModel::encode_audio_with_cache (gguf/model.rs:437-459,791-799) fed chunk by chunk.
    Per-chunk latency of vox_encode_audio_with_cache (host mel in -> host audio embeddings out, synchronous: what a streaming caller waits for) for 100-frame (1 s of
    audio) and 1200-frame (12 s, the CLI's chunk) chunks of a `seconds`-long stream -- long enough that the 750-row sliding window evicts (steady state)."""
    out = {}
    rng = np.random.default_rng(11)
    for cf in chunk_frames:
        n_chunks = max(int(seconds * 100 // cf), 6)
        mel = (0.6 * rng.standard_normal((128, cf * n_chunks)) + 0.3).astype(np.float32)
        cache = model.create_encoder_cache()
        lat = []
        for i in range(n_chunks):
            ch = np.ascontiguousarray(mel[:, i * cf:(i + 1) * cf])
            t0 = time.perf_counter(); emb = model.encode_audio_with_cache(ch[None], cache); lat.append((time.perf_counter() - t0) * 1e3)
        warm = lat[2:] if len(lat) > 4 else lat
        out[f"chunk_{cf}_frames"] = {"chunk_audio_s": cf / 100.0, "chunks": n_chunks, "rows_per_chunk": int(emb.shape[1]), "latency_ms_mean": round(float(np.mean(warm)), 3),
                                     "latency_ms_p50": round(float(np.median(warm)), 3), "latency_ms_max": round(float(np.max(warm)), 3), "first_chunk_ms": round(lat[0], 3),
                                     "rtf": round(float(np.mean(warm)) / 1e3 / (cf / 100.0), 5), "cache_rows_at_end": int(cache.seq_len()), "stream_rows": int(cache.abs_pos())}
        cache.close() if hasattr(cache, "close") else None
    out["workload"] = (f"{seconds:g} s stream through vox_encode_audio_with_cache, chunk by chunk (conv stem -> 32 layers against the cached K / V, window 750, evicting -> adapter); "
                       "latency = one synchronous call, host mel in, host embeddings out")
    return out


def piecewise_extra(pkg, gguf_path, x, ref_ids, reps=3):
    """The reference's OWN metric loop (bin/e2e_bench.rs:179-224: embed_tokens_from_ids -> + audio row -> forward_hidden_with_cache -> lm_head -> argmax -> scalar read-back,
    per token) replayed call for call through the C ABI by a plain C11 program (tools/e2e_piecewise.c; its own process and model load, gcc-built here): what a drop-in
    `e2e-bench` over this library reports -- next to `value`, which is the fused vox_transcribe_audio path."""
    import subprocess, tempfile
    d = tempfile.mkdtemp(prefix="vox_pw_"); exe = os.path.join(d, "e2e_piecewise"); wav = os.path.join(d, "clip.f32")
    pkg_dir = os.path.dirname(pkg.build.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "e2e_piecewise.c"), "-o", exe,
                           "-L" + pkg_dir, "-lvoxtral_hip", "-lm", "-Wl,-rpath," + pkg_dir])
    np.asarray(x, dtype=np.float32).tofile(wav)
    r = subprocess.run([exe, gguf_path, wav, str(reps)], capture_output=True, text=True, timeout=900)
    if r.returncode not in (0, 3):
        return {"error": (r.stderr or r.stdout)[-400:]}
    res = json.loads(r.stdout.strip().splitlines()[-1])
    ids = res.pop("ids")
    a, b = res["lm_head+argmax"], res["lm_head_argmax"]
    return {"workload": "bin/e2e_bench.rs:138-254 call for call over include/voxtral_hip.h from C11 (tools/e2e_piecewise.c): host preprocess, encode_audio, then per token "
                        "embed_tokens_from_ids -> tensor add -> forward_hidden_with_cache (one decode-engine launch) -> lm_head -> argmax + read-back, on device pointers",
            "tok_per_s": a["tok_per_s"], "decode_ms": a["decode_ms"], "encode_ms": a["encode_ms"], "preprocess_ms": res["preprocess_ms"], "rtf": a["rtf"], "decode_tokens": a["decode_tokens"],
            "tok_per_s_with_lm_head_argmax": b["tok_per_s"], "decode_engine": res["decode_engine"], "reps": res["reps"],
            "ids_equal_fused_path": bool(a["ids_equal_transcribe_audio"] and b["ids_equal_transcribe_audio"] and list(map(int, ref_ids)) == ids),
            "note": "tok_per_s = ids / decode-stage seconds as bin/e2e_bench.rs:236-240 (prefill included), the reference's definition"}


def fleurs_like_extra(pkg, ctx, model, t_embed, rank, world, dist, n_clips, batch, simulate_world=0, bcast_bytes=0, chunk_frames=0, sessions=1):
    """BASELINE configs[4] stand-in (no FLEURS offline): `n_clips` synthetic clips with FLEURS-like durations, LPT-sharded over the ranks
    (shard.run_sharded), each rank handing its share to vox_transcribe_batch in calls of <= `batch` clips (default: the whole share in one call -- continuous batching over
    16 .. 128 decode slots; 64: the round-4 form, length-bucketed lock-step batches); results gathered in input order.
    Wall time = barrier .. barrier, max over ranks.  Replaces the reference's serial per-file loop (bin/transcribe.rs:112-126)."""
    import importlib
    shard = importlib.import_module(pkg.__name__ + ".shard")
    durs = shard.fleurs_like_durations(n_clips, seed=7)
    parts = shard.lpt_partition(durs, world)
    need = set(parts[rank])
    if world == 1 and simulate_world > 1:
        need = set(range(n_clips))
    clips = {i: pkg.synth.synth_audio(durs[i], seed=9000 + i) for i in sorted(need)}      # every rank synthesises only its share (host, untimed)
    plans = None
    if chunk_frames > 0:
        # the reference CLI's pipeline (bin/transcribe.rs:207-265; what scripts/eval_wer.py:182-200 runs, i.e. what the published WER is defined on): the FILE is
        # peak-normalised once, split at --max-mel-frames (default 1200 = 192 000 samples), every chunk an independent unit -- here ALL chunks of a rank's files in
        # one vox_transcribe_batch_ex call, the file peaks reduced on the device (norm_group = file)
        cc = pkg.ChunkConfig.voxtral().with_max_frames(chunk_frames)
        plans = {i: (pkg.chunk_plan(clips[i].size, cc) if pkg.needs_chunking(clips[i].size, cc) else [(0, clips[i].size)]) for i in clips}

    # sessions > 1: the rank's share as `sessions` concurrent sessions on its GPU -- vox_model_set_sessions: the library's own hidden contexts + model replicas + threads
    # behind the SAME vox_transcribe_batch call (shard.SessionPool is the host-threads twin of it)
    if sessions > 1 and len(parts[rank]) < sessions * shard.SessionPool.MIN_UNITS_PER_SESSION:
        sessions = 1      # a share this small runs as one session anyway (128 units per session): no replica is made for it
    if sessions > 1:
        model.set_sessions(sessions)
    runner = model

    def batch_work(idx_list):
        if plans is None:
            outs = runner.transcribe_batch([clips[i] for i in idx_list], t_embed)
            return [len(o) for o in outs]
        units, grp, owner = [], [], []
        for i in idx_list:
            for a, b in plans[i]:
                units.append(clips[i][a:b]); grp.append(i); owner.append(i)
        outs = runner.transcribe_batch(units, t_embed, norm_group=grp)
        per = {i: 0 for i in idx_list}
        for i, o in zip(owner, outs):
            per[i] += len(o)
        return [per[i] for i in idx_list]

    runner.transcribe_batch([clips[i] for i in parts[rank][:min(batch, len(parts[rank]))]], t_embed)      # warm-up: workspaces + kernels
    # a rank that throws inside the sharded section would leave the others in a collective until the NCCL timeout; so every rank runs its local
    # work under try/except, the ranks agree on success (one all_reduce) BEFORE any gather, and a failure anywhere skips the extra on every rank
    ok = 1; err = None; res = None; dt = 0.0
    if world > 1:
        dist.barrier()
    eng_n0 = model.set_batch_engine()[1]
    ctx.synchronize(); t0 = time.perf_counter()
    try:
        mine = []
        for grp in shard.length_buckets(parts[rank], durs, batch):
            mine.extend(zip(grp, batch_work(grp)))
    except Exception as e:
        ok = 0; err = str(e)
    ctx.synchronize(); dt = time.perf_counter() - t0
    tm_last = model.timings()      # rank 0's LAST call inside the timed region (the whole share when it is one call): stage times, decode steps
    if sessions > 1:
        model.set_sessions(1)
    eng_corpus = model.set_batch_engine()[1] - eng_n0      # rank 0's engine launches inside the timed region (steps with <= 2 active slot groups: one launch each)
    if world > 1:
        import torch
        tt = torch.tensor([dt, float(ok)], dtype=torch.float64, device=f"cuda:{torch.cuda.current_device()}")
        mx = tt.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX); mn = tt.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dt = float(mx[0].item())
        if float(mn[1].item()) < 1.0:
            return {"error": err or "a rank failed inside the sharded section"} if rank == 0 else None
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank == 0:
            res = [None] * n_clips
            for part in gathered:
                for i, r in part:
                    res[i] = r
    else:
        if not ok:
            return {"error": err}
        res = [None] * n_clips
        for i, r in mine:
            res[i] = r
    if rank != 0:
        return None
    total_s = float(sum(durs)); ntok = int(sum(res))
    sim = None
    if world == 1 and simulate_world > 1:
        # Replicas share nothing in the data path, so the N-GPU wall time of this corpus = the slowest rank's share run alone.  On ONE GPU: run every rank's
        # longest-first share serially, time each; predicted_scaling = T(all clips on one rank) / max_r T(share r).  The start-up broadcast (primary arena over
        # xGMI) is reported beside it, not folded in: it is paid once per process, not per corpus.
        sparts = shard.lpt_partition(durs, simulate_world); per = []
        for grp in shard.length_buckets(sparts[0], durs, batch):      # untimed warm-up at a share's size (workspace pool; the real N-GPU run warms up the same way above)
            batch_work(grp)
        eng_share = []
        for r in range(simulate_world):
            e0 = model.set_batch_engine()[1]
            ctx.synchronize(); t1 = time.perf_counter()
            for grp in shard.length_buckets(sparts[r], durs, batch):
                batch_work(grp)
            ctx.synchronize(); per.append(time.perf_counter() - t1); eng_share.append(model.set_batch_engine()[1] - e0)
        sim = {"world": simulate_world, "engine_launches_per_rank": eng_share, "per_rank_s": [round(v, 3) for v in per], "clips_per_rank": [len(q) for q in sparts],
               "batches_per_rank": [[len(g) for g in shard.length_buckets(q, durs, batch)] for q in sparts][:2],
               "predicted_wall_s": round(max(per), 3), "predicted_scaling": round(dt / max(per), 2), "predicted_efficiency": round(dt / max(per) / simulate_world, 3),
               "lpt_imbalance": round(shard.imbalance(durs, sparts), 4), "weight_broadcast_bytes": int(bcast_bytes),
               "weight_broadcast_s_estimate": round(bcast_bytes / 100e9, 3),
               "note": "one-GPU bound on the N-GPU curve of this corpus (replicas only: no data-path collective); broadcast estimate at 100 GB/s per xGMI ring hop"}
    n_units = sum(len(plans[i]) for i in range(n_clips)) if (plans is not None and len(plans) == n_clips) else (None if plans is not None else n_clips)
    pipe = (f"reference CLI pipeline (bin/transcribe.rs:207-265, --max-mel-frames {chunk_frames}): file peak-normalised once, split into {chunk_frames}-frame chunks, every chunk a unit "
            f"of the batch (vox_transcribe_batch_ex, norm_group = file)") if chunk_frames > 0 else "un-chunked e2e-bench pipeline (bin/e2e_bench.rs:98-135)"
    return {"simulated_world": sim, "pipeline": pipe, "units": n_units, "chunk_frames": int(chunk_frames), "sessions_per_gpu": int(sessions),
            "workload": f"{n_clips} synthetic clips, FLEURS-like log-normal durations (median 10 s, 3..30 s, rng 7), host samples -> ids; LPT shards over {world} rank(s), "
                        f"{'one vox_transcribe_batch call per rank (continuous batching)' if batch >= n_clips else f'{batch}-clip length-bucketed calls'} (BASELINE configs[4] stand-in; no FLEURS / WER offline)",
            "clips": n_clips, "audio_s": round(total_s, 1), "wall_s": round(dt, 3), "rtf": round(dt / total_s, 6), "tok_per_s": round(ntok / dt, 1),
            "ids": ntok, "lpt_imbalance": round(shard.imbalance(durs, parts), 4), "batch": batch, "engine_launches": int(eng_corpus), "batch_engine_active": bool(model.set_batch_engine()[0]),
            # rank 0's last call: front-end / encoder / (prefill + decode steps) wall times and the number of decode steps (graph replays); with sessions: the longest session's stages, all replays
            "last_call_stage_ms": {k: round(tm_last[k], 1) for k in ("preprocess_ms", "encode_ms", "decode_ms")}, "last_call_decode_steps": int(tm_last["graph_replays"]),
            "last_call_ms_per_step_incl_prefill": round(tm_last["decode_ms"] / max(int(tm_last["graph_replays"]), 1), 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=16.0)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=float(os.environ.get("VOX_CPU_BASELINE_S", "16.0")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=16, help="also report BASELINE configs[3] (B utterances through vox_transcribe_batch) at N=1; 0 = skip")
    ap.add_argument("--no-piecewise", action="store_true", help="skip the piecewise extra (the reference's e2e-bench decode loop call for call through the C ABI, tools/e2e_piecewise.c, N = 1)")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32 SafeTensors extra (BASELINE configs[1], N = 1)")
    ap.add_argument("--fleurs-clips", type=int, default=647, help="clips of the FLEURS-like sharded extra (BASELINE configs[4] stand-in); 0 = skip")
    ap.add_argument("--simulate-world", type=int, default=8, help="N = 1 only: also run each of W ranks' share of the FLEURS-like corpus serially on this GPU and report the predicted 1 -> W scaling (0 = skip)")
    ap.add_argument("--fleurs-batch", type=int, default=0, help="clips per vox_transcribe_batch call of the FLEURS-like extra; 0 (default) = a rank's whole share in ONE call (continuous batching over slots, round 5); 64 = the round-4 length-bucketed lock-step batches")
    ap.add_argument("--cli-chunk-frames", type=int, default=1200, help="also run the FLEURS-like corpus on the reference CLI's pipeline: files split into chunks of this many mel frames "
                    "(bin/transcribe.rs:55-57 default 1200), every chunk a unit of the batch (`fleurs_like_cli`); 0 = skip")
    ap.add_argument("--corpus-sessions", type=int, default=2, help="also run the FLEURS-like corpus with this many concurrent sessions per GPU (vox_model_set_sessions: hidden contexts + model replicas + "
                    "library threads behind the same vox_transcribe_batch call; `fleurs_like_sessions`); <= 1 = skip")
    ap.add_argument("--gemv-iters", type=int, default=260)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by hand: spawn the N ranks ourselves, exactly the way the driver does
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            log(f"[bench] --gpus {args.gpus} requested but this box has {have} GPU(s): refusing to run (no silent single-rank fallback)")
            sys.exit(2)
        from __graft_entry__ import load_package
        import importlib
        shard = importlib.import_module(load_package().__name__ + ".shard")
        sys.exit(shard.spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"[bench] WORLD_SIZE={world} does not match --gpus {args.gpus}: refusing to run"); sys.exit(2)
    # torch first: it bundles its own libamdhip64; importing it before libvoxtral_hip.so keeps ONE HIP runtime in-process.
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))

    def barrier():
        if world > 1:
            dist.barrier()

    from __graft_entry__ import load_package
    pkg = load_package()
    pkg.lib()   # fails loudly if the HIP library is missing
    ctx = pkg.Context(local)
    path = full_gguf_path(pkg, args.seed, rank, barrier)

    t0 = time.time()
    # N > 1: rank 0 parses + repacks; the PRIMARY part of the device arena (the Q4 row planes + f32 tensors: 2.5 GB) reaches the other ranks by ONE RCCL broadcast over
    # xGMI issued on the arena memory itself; every rank derives the rest on its own GPU (shard.load_replicated -- the same start-up cli.py / wer.py --gpus N use)
    import importlib
    shard_mod = importlib.import_module(pkg.__name__ + ".shard")
    bst = {}
    model = shard_mod.load_replicated(pkg, ctx, path, rank, world, local=local, stats=bst)
    bcast_s = float(bst.get("seconds", 0.0)); bcast_bytes = int(bst.get("bytes", 0))
    load_s = time.time() - t0
    cfg = model.config
    t_embed = pkg.TimeEmbedding(cfg.dec_dim).embed(6.0)

    # inputs resident in HBM before the timed region (each rank its own utterance: independent units)
    x = pkg.synth.synth_audio(args.seconds, seed=1234 + rank)
    d_x = ctx.upload(x)
    ids = None
    for _ in range(max(args.warmup, 0)):
        ids = model.transcribe_audio(None, t_embed, device_ptr=d_x, n_samples=x.size)
    barrier(); torch.cuda.synchronize(); ctx.synchronize()
    stage_ms = {"preprocess_ms": 0.0, "encode_ms": 0.0, "decode_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = model.transcribe_audio(None, t_embed, device_ptr=d_x, n_samples=x.size)
        tm = model.timings()
        for k in stage_ms:
            stage_ms[k] += tm[k]
    ctx.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed_local = time.perf_counter() - t0
    elapsed = elapsed_local
    per_rank_ms = [elapsed_local * 1e3 / max(args.steps, 1)]; per_rank_bcast_s = [bcast_s]
    if world > 1:
        tt = torch.tensor([elapsed, bcast_s], dtype=torch.float64, device=f"cuda:{local}")
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank_ms = [float(v[0].item()) * 1e3 / max(args.steps, 1) for v in allt]
        per_rank_bcast_s = [float(v[1].item()) for v in allt]
        elapsed = max(float(v[0].item()) for v in allt)
    # what the collective layer actually saw (a line that says how many ranks RCCL had, not only how many were asked for)
    try:
        rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        rccl = None
    dist_info = {"world_size_seen": dist.get_world_size() if world > 1 else 1, "backend": str(dist.get_backend()) if world > 1 else None, "rccl_version": rccl,
                 "weight_broadcast_s_per_rank": [round(v, 3) for v in per_rank_bcast_s], "broadcast_ordered_by_event": bool(bst.get("ordered_by_event", False)),
                 "gpus_visible": torch.cuda.device_count()}
    n_ids = int(len(ids))
    steps = max(args.steps, 1)
    for k in stage_ms:
        stage_ms[k] /= steps
    ms_per_step = elapsed * 1e3 / steps
    value = world * steps * n_ids / elapsed

    fleurs = None
    if args.fleurs_clips > 0:
        try:
            fleurs = fleurs_like_extra(pkg, ctx, model, t_embed, rank, world, dist, args.fleurs_clips, max(1, min(4096, args.fleurs_batch)) if args.fleurs_batch > 0 else 4096,
                                       simulate_world=args.simulate_world, bcast_bytes=model.arena()[1])
        except Exception as e:     # an extra never costs the headline line
            fleurs = {"error": str(e)} if rank == 0 else None
    fleurs_cli = None
    if args.fleurs_clips > 0 and args.cli_chunk_frames > 0:
        try:      # the same corpus on the reference CLI's semantics (1200-frame chunks as units): the pipeline wer.py / `voxtral-transcribe --batch` run
            fleurs_cli = fleurs_like_extra(pkg, ctx, model, t_embed, rank, world, dist, args.fleurs_clips, 4096, simulate_world=args.simulate_world, bcast_bytes=model.arena()[1],
                                           chunk_frames=args.cli_chunk_frames)
        except Exception as e:
            fleurs_cli = {"error": str(e)} if rank == 0 else None

    fleurs_s = None
    if args.fleurs_clips > 0 and args.corpus_sessions > 1:
        try:      # the same corpus (un-chunked), every rank's share as `corpus_sessions` concurrent sessions on its GPU (overlap inside a rank's share)
            fleurs_s = fleurs_like_extra(pkg, ctx, model, t_embed, rank, world, dist, args.fleurs_clips, 4096, simulate_world=0, bcast_bytes=model.arena()[1], sessions=args.corpus_sessions)
        except Exception as e:
            fleurs_s = {"error": str(e)} if rank == 0 else None

    out = None
    if rank == 0:
        out = {
            "metric": "decode_tok_per_s", "value": round(value, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / REF_TOK_S, 2), "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"single {args.seconds:g} s 16 kHz clip per GPU, Q4_0 GGUF (BASELINE configs[2]); un-chunked e2e-bench pipeline: "
                                   f"mel -> 32-layer encoder -> adapter -> 38-token prefill + {n_ids - 1} decode steps",
                       "weights": "synthetic Q4_0, real Voxtral-Mini-4B-Realtime shapes (711 tensors, 2.5 GB)",
                       "clip_s": args.seconds, "ids_per_clip": n_ids, "batch": 1, "parallelism": f"replicas x{world}",
                       "arithmetic": "f32 activations and accumulation; Q4_0 weights exact.  Decode-engine dot products: Q4 nibbles x per-block fixed-point activations (three 7-bit digits + a "
                                     "signed top digit, 2^-26 of the block maximum) as exact int32 sums on v_mfma_i32_16x16x64_i8, f32 from the block scale on; prefill / encoder / batched "
                                     "GEMMs: v_mfma_f32_16x16x32_bf16 on exact integer weights and bf16 hi + lo activations (2^-17 relative), f32 accumulation -- no path narrower than the "
                                     "reference's f32 (engine vs per-operator logits 1.5e-6 of the largest)"},
            "rtf": round(ms_per_step / 1e3 / args.seconds, 5),
            "decode_tok_per_s_ref_def": round(n_ids / (stage_ms["decode_ms"] / 1e3), 2) if stage_ms["decode_ms"] > 0 else None,
            "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()},
            "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
            "distributed": dist_info, "load_s": round(load_s, 2), "weight_broadcast_s": round(bcast_s, 3), "weight_broadcast_bytes": bcast_bytes, "weight_bytes": model.weight_bytes(), "device_memory": model.memory(),
            "note": "value = ids emitted by all ranks / max-over-ranks wall time of the whole pipeline; decode_tok_per_s_ref_def follows "
                    "bin/e2e_bench.rs:236-240 (ids / decode-stage time); vs_baseline divides by the reference's 19.4 tok/s measured on a DGX Spark GB10",
        }
        if fleurs_cli is not None:
            out["fleurs_like_cli"] = fleurs_cli
            if fleurs is not None and "error" not in fleurs and "error" not in fleurs_cli:
                out["fleurs_like_cli"]["tok_per_s_vs_unchunked"] = round(fleurs_cli["tok_per_s"] / fleurs["tok_per_s"], 3)
        if fleurs_s is not None:
            out["fleurs_like_sessions"] = fleurs_s
            if fleurs is not None and "error" not in fleurs and "error" not in fleurs_s:
                out["fleurs_like_sessions"]["tok_per_s_vs_one_session"] = round(fleurs_s["tok_per_s"] / fleurs["tok_per_s"], 3)
                out["fleurs_like_sessions"]["same_ids_as_one_session"] = bool(fleurs_s["ids"] == fleurs["ids"])
        if fleurs is not None:
            out["fleurs_like"] = fleurs
            if "error" not in fleurs:
                # STRONG scaling, first-class next to the weak-scaling `value`: the fixed 647-clip FLEURS-like corpus (BASELINE configs[4] stand-in) sharded over the N ranks --
                # total work fixed as N grows; the driver's per-N lines give the configs[4] curve directly (tok/s at N / tok/s at 1)
                out["strong_scaling"] = {"metric": "fleurs_like_corpus_tok_per_s", "value": fleurs["tok_per_s"], "unit": "tok/s", "n_gpus": world, "scaling": "strong",
                                         "wall_s": fleurs["wall_s"], "rtf": fleurs["rtf"], "clips": fleurs["clips"], "audio_s": fleurs["audio_s"],
                                         "predicted_8gpu_scaling_from_one_gpu": (fleurs.get("simulated_world") or {}).get("predicted_scaling")}
        # ---- roofline of the dominant kernel: the fused gate/up Q4 GEMV (w1|w3, 26 launches per token), HIP events on our stream
        names = ["qkv", "wo", "w1w3", "w2", "lm_head"]; per = {}
        per_step_us = 0.0; per_step_bytes = 0.0
        for which, nm in enumerate(names):
            us, nbytes, kname = model.bench_decode_gemv(which, args.gemv_iters if which != 4 else 40)
            per[nm] = {"avg_us": round(us, 3), "bytes": int(nbytes), "GBps": round(nbytes / us / 1e3, 1), "kernel": kname}
            mult = 1 if which == 4 else cfg.dec_layers
            per_step_us += us * mult; per_step_bytes += nbytes * mult
        dom = per["w1w3"]
        engine = None
        if model.set_decode_engine(True):      # the product's decode step is ONE launch of the persistent engine: that launch is the dominant kernel (92 % of the clip's wall time)
            # three passes of 200 launches (graph replays of 50 per position quarter), the MEDIAN pass reported
            passes = [model.bench_decode_gemv(5, 200) for _ in range(3)]      # (50 launches per position quarter: with 10 the first launch behind each quarter's event -- a clock ramp of ~250 us on some boxes -- read as +25 us per launch)
            us, nbytes, kname = sorted(passes, key=lambda p: p[0])[1]
            engine = {"avg_us": round(us, 2), "bytes": int(nbytes), "GBps": round(nbytes / us / 1e3, 1), "kernel": kname, "avg_us_passes": [round(p[0], 2) for p in passes]}
            per["decode_engine"] = engine; dom = engine
        # HBM traffic per launch from the PMC counters: collected offline with rocprofv3 (separate --pmc passes, gfx950 FETCH_SIZE x2
        # correction) and committed under profiles/; a live run cannot read PMCs, so this is the committed measurement or null.
        traffic, traffic_src = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            import re
            key = re.sub(r"[A-Z]+=", "", dom["kernel"]).replace(" ", "")       # "q4_gemv_kernel<P=3,R=2,...>" -> the demangled template name
            for k, v in pm["hbm_bytes_per_launch"].items():
                if k.replace(" ", "") == key:
                    traffic, traffic_src = int(v), pm["source"]
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(dom["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": dom["bytes"], "avg_launch_us": dom["avg_us"],
                           "all_decode_gemvs": per,
                           "all_decode_gemvs_note": "decode_engine = the whole step (26 layers + lm_head) as the ONE launch the product replays per token -- and the only one: a replayed launch "
                                                    "forms its own input from the previous launch's argmax partials; the five GEMV rows are the per-operator launches it replaces "
                                                    "(stand-alone, 26 layers cycled), kept for comparison",
                           "decode_step_launches": 1 if engine else 4 * cfg.dec_layers + 2,
                           "decode_step_gemv_GBps": round(per_step_bytes / per_step_us / 1e3, 1),
                           "decode_step_algorithmic_bytes": int(per_step_bytes),
                           "decode_step_measured_ms": round(stage_ms["decode_ms"] / max(n_ids, 1), 4),
                           "decode_step_frac_of_hbm_peak": round(per_step_bytes / (stage_ms["decode_ms"] / 1e3 / max(n_ids, 1)) / 1e9 / HBM_PEAK_GBS, 4)}
        if world == 1 and args.batch > 1:
            # BASELINE configs[3]: B x 16 s utterances on one GPU -- stacked encoder (every GEMM once over all frames), stacked prefill,
            # one batched decode step per position (weights streamed once per step for the whole batch).  An extra, not `value`.
            clips = [pkg.synth.synth_audio(args.seconds, seed=4321 + i) for i in range(args.batch)]
            ptrs = [ctx.upload(c) for c in clips]; lens = [c.size for c in clips]
            model.transcribe_batch(None, t_embed, device_ptrs=ptrs, n_samples=lens)                       # warm-up
            reps = 3; ctx.synchronize(); tb = time.perf_counter()
            eng_on, eng_n0 = model.set_batch_engine()
            for _ in range(reps):
                outs = model.transcribe_batch(None, t_embed, device_ptrs=ptrs, n_samples=lens)
            ctx.synchronize(); bdt = (time.perf_counter() - tb) / reps
            eng_on, eng_n1 = model.set_batch_engine()
            tmb = model.timings(); ntok = sum(len(o) for o in outs)
            step_ms = tmb["decode_ms"] / max(ntok // args.batch, 1)
            out["batch"] = {"workload": f"{args.batch} x {args.seconds:g} s clips, Q4_0 (BASELINE configs[3])", "batch": args.batch,
                            "tok_per_s": round(ntok / bdt, 1), "ms_per_batch": round(bdt * 1e3, 2), "rtf": round(bdt / (args.seconds * args.batch), 5),
                            "stage_ms": {k: round(tmb[k], 2) for k in ("preprocess_ms", "encode_ms", "decode_ms")},
                            "decode_step_ms": round(step_ms, 4),
                            "decode_step_weight_GBps": round(per_step_bytes / 1e9 / (step_ms / 1e3), 1),
                            "decode_step_frac_of_hbm_peak": round(per_step_bytes / 1e9 / (step_ms / 1e3) / HBM_PEAK_GBS, 4),
                            "decode_layer_engine": bool(eng_on) and (eng_n1 - eng_n0) > 0, "engine_launches_per_batch": (eng_n1 - eng_n0) // reps}
            # roofline of the batched step: ALGORITHMIC bytes per step = the decoder's Q4 blocks + the tied lm_head once + the K / V rows the step's
            # attention reads (f32, 212 992 B per position and row at the real geometry; positions 38 .. S - 1, mean taken), divided by the measured step time; `traffic` =
            # the committed PMC figure of the engine launch (fabric bytes, Infinity-Cache hits included; profiles/r06_pmc_batch_engines.txt) + the lm_head's algorithmic bytes
            n_steps = max(ntok // args.batch - 1, 1); mean_pos = PREFIX_LEN + n_steps / 2.0
            kv_bytes = args.batch * mean_pos * 2.0 * cfg.dec_layers * cfg.dec_kv_heads * cfg.dec_head_dim * 4.0
            alg = per_step_bytes + kv_bytes
            eng_traffic = None
            try:
                eng_traffic = int(json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["hbm_bytes_per_launch"]["decode_engine_b16_kernel<1, false>"])
            except Exception:
                pass
            out["batch"]["roofline"] = {"bound": "hbm", "kernel": "decode_engine_b16_kernel<1, false> + lm_head + argmax (one batched decode step)", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                        "algorithmic_bytes_per_step": int(alg), "of_which_kv_bytes": int(kv_bytes), "achieved": round(alg / 1e9 / (step_ms / 1e3), 1),
                                        "frac": round(alg / 1e9 / (step_ms / 1e3) / HBM_PEAK_GBS, 4),
                                        "traffic": (eng_traffic + int(per["lm_head"]["bytes"])) if eng_traffic else None,
                                        "traffic_source": "profiles/r06_pmc_batch_engines.txt (engine launch, FETCH_SIZE x 2) + the lm_head's algorithmic bytes"}
            for pp in ptrs:
                ctx.free(pp)
        if world == 1:
            try:
                out["streaming_encoder"] = streaming_encoder_extra(pkg, ctx, model)
            except Exception as e:
                out["streaming_encoder"] = {"error": str(e)}
        if world == 1 and not args.no_piecewise:
            try:
                out["piecewise"] = piecewise_extra(pkg, path, x, ids)
            except Exception as e:
                out["piecewise"] = {"error": str(e)}
        if world == 1 and not args.no_f32:
            try:
                out["f32"] = f32_extra(pkg, ctx, t_embed, args.seconds)
            except Exception as e:
                out["f32"] = {"error": str(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(pkg, path, args.cpu_baseline_seconds)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "tok/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    ctx.free(d_x)
    model.close(); ctx.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
