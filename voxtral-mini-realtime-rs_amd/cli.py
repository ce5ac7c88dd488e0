"""`voxtral-transcribe`-compatible driver (reference: src/bin/transcribe.rs:27-318): same flags, one line of text per
input on stdout, logs on stderr.  WAV ingest / resampling / chunking / token decode stay on the CPU (caller side in the
reference too); pad -> log-mel -> encoder -> decoder run on the GPU through the C ABI.

    python -m ... cli --gguf model.gguf --tokenizer tekken.json --audio a.wav [--audio b.wav] [--delay 6] [--max-mel-frames 1200]
"""
from __future__ import annotations

import argparse
import sys
import time
import wave

import numpy as np


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def load_wav(path):
    """audio/io.rs:90-131: PCM int (8/16/24/32 bit, hound's i32 samples) mixed to mono as (sum of the channels as f32 / channels) / 2^(bits-1);
    IEEE float: mean of the channels."""
    with wave.open(path, "rb") as w:
        ch, sw, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if sw == 2:
        v = np.frombuffer(raw, dtype="<i2").astype(np.int64)
    elif sw == 4:
        v = np.frombuffer(raw, dtype="<i4").astype(np.int64)
    elif sw == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16); v = np.where(v & 0x800000, v - (1 << 24), v)
    elif sw == 1:
        v = np.frombuffer(raw, dtype=np.uint8).astype(np.int64) - 128            # hound: 8-bit WAV is unsigned, read as signed
    else:
        raise ValueError(f"unsupported WAV sample width {sw}")
    s = v.reshape(-1, ch).sum(axis=1) if ch > 1 else v                            # i32 sum of the frame's channels (io.rs:110-113)
    x = (s.astype(np.float32) / np.float32(ch)) / np.float32(1 << (8 * sw - 1))
    return x.astype(np.float32), sr


def resample_to_16k(x, sr, ctx=None, pkg=None):
    """audio/resample.rs:10-52.  vox_resample on the GPU: the algorithm of rubato's synchronous FFT resampler, which is what the reference calls
    (tests/test_resample.py pins it against the CPU oracle's independent restatement; the crate itself is outside the reference's tree, so parity with ITS
    output is unpinned and recorded as such in DESIGN.md).  There is no host fallback: without the HIP library this CLI cannot resample."""
    if ctx is None or pkg is None:
        raise RuntimeError("resample_to_16k needs a GPU context (the product path has no CPU fallback)")
    return pkg.resample_to_16k(ctx, x, sr)


def transcribe_one(pkg, path, model, tokenizer, mel, pad_cfg, chunk_cfg, t_embed):
    """bin/transcribe.rs:187-276"""
    x, sr = load_wav(path)
    if sr != 16000:
        log(f"  resampling {sr} Hz -> 16 kHz"); x = resample_to_16k(x, sr, mel.ctx, pkg)
    x = pkg.peak_normalize(x, 0.95)                                              # :207
    chunks = pkg.chunk_audio(x, chunk_cfg) if pkg.needs_chunking(x.size, chunk_cfg) else [None]
    texts = []
    for i, ch in enumerate(chunks):
        samples = x if ch is None else ch.samples
        if len(chunks) > 1:
            log(f"  chunk {i + 1}/{len(chunks)}: {ch.start_sample / 16000:.2f}-{ch.end_sample / 16000:.2f} s")
        m = mel.compute_log(pkg.pad_audio(samples, pad_cfg))                     # :279-306
        if m.shape[0] == 0:
            raise RuntimeError("Audio too short to produce mel frames")
        ids = model.transcribe_streaming(np.ascontiguousarray(m.T)[None], t_embed)
        text = tokenizer.decode([t for t in ids if t >= 1000]).strip()           # :309-318
        if text:
            texts.append(text)
    return " ".join(texts)


def file_costs(paths):
    """Decode cost of a file ~ its duration (the reference has no early stop): the file size is the proxy that needs no decode."""
    import os
    return [float(os.path.getsize(p)) if os.path.exists(p) else 0.0 for p in paths]


def unit_table(pkg, paths, chunk_cfg):
    """The units of work of a file list, from the WAV headers alone (so every rank derives the same table without decoding anything): every file is split exactly like
    bin/transcribe.rs:210-226 -- one unit if it fits `max_mel_frames`, its chunks otherwise -- at the length it will have at 16 kHz.
    -> [(file index, chunk index, start_sample, end_sample)], file-major.  Unreadable / empty files contribute nothing: their line (and their error) comes from the
    one-by-one path."""
    units = []
    for i, p in enumerate(paths):
        try:
            with wave.open(p, "rb") as w:
                n, sr = w.getnframes(), w.getframerate()
            n16 = pkg.resample_len(n, sr) if sr != 16000 else n
            if n16 <= 0:
                continue
            plan = pkg.chunk_plan(n16, chunk_cfg) if pkg.needs_chunking(n16, chunk_cfg) else [(0, n16)]
            units.extend((i, k, a, b) for k, (a, b) in enumerate(plan) if b > a)
        except Exception:
            pass
    return units


class UnitRunner:
    """Transcribes units (file, chunk, start, end) through vox_transcribe_batch_ex.  A file is loaded, resampled and peak-normalised ONCE, as a file
    (bin/transcribe.rs:197-207); its chunks are views into it handed over as already normalised (norm_group < 0), so a file's chunks may sit in different calls or
    on different ranks and still carry the file's scale.  Returns the text of every unit (None = failed: the file falls back to the one-by-one path)."""

    def __init__(self, pkg, ctx, model, tokenizer, paths, t_embed):
        self.pkg, self.ctx, self.model, self.tok, self.paths, self.t_embed = pkg, ctx, model, tokenizer, paths, t_embed
        self.files = {}

    def samples(self, i):
        if i not in self.files:
            try:
                x, sr = load_wav(self.paths[i])
                if sr != 16000:
                    x = resample_to_16k(x, sr, self.ctx, self.pkg)
                self.files[i] = self.pkg.peak_normalize(x, 0.95)
            except Exception as e:
                log(f"  (batched path) cannot read {self.paths[i]}: {e}"); self.files[i] = None
        return self.files[i]

    def __call__(self, units):
        views, keep = [], []
        for u, (i, k, a, b) in enumerate(units):
            x = self.samples(i)
            if x is not None and b <= x.size:      # (a header that lied about the length: leave the file to the one-by-one path)
                views.append(x[a:b]); keep.append(u)
        out = [None] * len(units)
        if not views:
            return out
        try:
            t1 = time.time(); ids = self.model.transcribe_batch(views, self.t_embed, norm_group=[-1] * len(views))
            log(f"batch of {len(views)}: {time.time() - t1:.3f}s")
            for u, r in zip(keep, ids):
                out[u] = self.tok.decode([t for t in r if t >= 1000]).strip()           # :309-318, per chunk
        except Exception as e:
            log(f"batched path failed ({e}); falling back to one by one")
        for i in {units[u][0] for u in keep}:      # a file whose units are all done is dropped from memory
            self.files.pop(i, None)
        return out


def join_units(n_files, units, unit_texts):
    """bin/transcribe.rs:261-275: a file's line = its non-empty chunk texts joined by one space.  -> {file index: line} for the files whose units ALL came back."""
    per = {}
    for (i, k, _, _), t in zip(units, unit_texts):
        per.setdefault(i, []).append((k, t))
    lines = {}
    for i, lst in per.items():
        if all(t is not None for _, t in lst):
            lines[i] = " ".join(t for _, t in sorted(lst) if t)
    return lines


def _gloo_gather(fn):
    """Run `fn(group)` -- which ends in ONE gather of python objects -- over a GLOO group whatever the default group is: with `--gguf --gpus N` the default group is
    RCCL (the start-up weight broadcast), whose object collectives stage through GPU buffers and fall under the NCCL watchdog's 10-minute collective timeout -- a rank
    that is still transcribing a long share must not be killed by the ranks that wait for it.  Every rank calls this, so every rank takes part in new_group."""
    import datetime
    import torch.distributed as dist
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", timeout=datetime.timedelta(hours=12))
    sub = None
    try:
        group = None
        if not own and dist.get_backend() != "gloo":
            sub = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=12)); group = sub
        return fn(group)
    finally:
        if sub is not None:
            dist.destroy_process_group(sub)
        if own:
            dist.barrier(); dist.destroy_process_group()


def sharded_lines(pkg, paths, one, rank, world, group=None):
    """N > 1 ranks: every rank transcribes its longest-first share of `paths` with `one(index) -> text`; rank 0 returns every line in input order
    (None elsewhere).  The only collective is the final gather of the text lines (over gloo: _gloo_gather)."""
    if group is not None:
        return pkg.shard.run_sharded(list(range(len(paths))), file_costs(paths), one, rank, world, group=group)
    return _gloo_gather(lambda g: pkg.shard.run_sharded(list(range(len(paths))), file_costs(paths), one, rank, world, group=g))


def sharded_units(pkg, units, runner, batch, rank, world, group=None):
    """N > 1 ranks, chunks as units of work: the units are LPT-partitioned by their length (decode cost ~ samples), every rank hands its share to `runner` in calls of
    <= `batch` units; rank 0 returns every unit's text in table order (None elsewhere)."""
    costs = [float(b - a) for _, _, a, b in units]
    run = lambda g: pkg.shard.run_sharded(units, costs, None, rank, world, group=g, batch=max(batch, 2), batch_work=runner)
    return run(group) if (group is not None or world == 1) else _gloo_gather(run)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="voxtral-transcribe", description="Transcribe audio using Voxtral Mini 4B Realtime (MI355X HIP path)")
    ap.add_argument("-a", "--audio", action="append", default=[])
    ap.add_argument("--audio-list")
    ap.add_argument("-m", "--model", default="models/voxtral")
    ap.add_argument("--gguf")
    ap.add_argument("--tokenizer")
    ap.add_argument("-d", "--delay", type=int, default=6)
    ap.add_argument("--max-mel-frames", type=int, default=1200)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1, help="extension: shard the input files over N GPUs of this node (one process per GPU, longest-first "
                    "assignment, no collective in the data path; stdout keeps one line per input in input order)")
    ap.add_argument("--batch", type=int, default=1, help="extension: every file is split into its --max-mel-frames chunks and up to N chunks go into one "
                    "vox_transcribe_batch_ex call (continuous batching; same ids per chunk and the same lines as one by one; output order unchanged)")
    ap.add_argument("--sessions-per-gpu", type=int, default=1, help="extension (with --batch, Q4 GGUF): this many concurrent sessions on every GPU (own context + model replica + "
                    "library thread each: vox_model_set_sessions): one session's launch-bound decode steps leave gaps a second session fills (647 FLEURS-like clips: 3.7 s against 4.3 s; "
                    "calls with fewer than 128 units per session stay on one); same lines")
    a = ap.parse_args(argv)
    if a.audio_list and a.audio:
        ap.error("--audio-list conflicts with --audio")
    if a.gguf and not a.tokenizer:
        ap.error("--gguf requires --tokenizer")
    paths = a.audio
    if a.audio_list:
        paths = [l.strip() for l in open(a.audio_list) if l.strip()]
    if not paths:
        ap.error("No audio files specified")
    if a.max_mel_frames <= 0:
        ap.error("--max-mel-frames must be greater than 0")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from __graft_entry__ import load_package
    pkg = load_package()
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if a.gpus > 1 and world == 1:
        # the reference transcribes its files one after the other (bin/transcribe.rs:112-126); here they are independent units of work for N replicas:
        # re-launch this command as N ranks (the driver's own launch for bench.py --gpus N); rank 0 prints every line, in input order
        n_dev = pkg.device_count()
        if n_dev < a.gpus:
            log(f"Error: --gpus {a.gpus} requested but {n_dev} GPU(s) visible"); return 2
        rc, out = pkg.shard.spawn_ranks(a.gpus, os.path.abspath(__file__), list(argv if argv is not None else sys.argv[1:]), capture=True)
        sys.stdout.write(out); sys.stdout.flush()
        return rc
    own_group = False
    if world > 1:
        a.device = int(os.environ.get("LOCAL_RANK", rank))
        if a.gguf:
            # torch BEFORE the first call into libvoxtral_hip.so (it bundles its own libamdhip64: one HIP runtime per process), then RCCL for the start-up broadcast
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(a.device)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", a.device)); own_group = True
    tok_path = a.tokenizer or os.path.join(a.model, "tekken.json")
    if not os.path.exists(tok_path):
        log(f"Error: Tokenizer not found at {tok_path}"); return 1
    tokenizer = pkg.VoxtralTokenizer.from_file(tok_path)
    ctx = pkg.Context(a.device)
    t0 = time.time()
    if a.gguf and world > 1:
        # N replicas: rank 0 reads the file, the packed weights reach the other GPUs with ONE RCCL broadcast over xGMI (shard.load_replicated) instead of N file reads
        st = {}
        log(f"[rank {rank}] Loading Q4 GGUF model from {a.gguf} ({'file' if rank == 0 else 'layout only + broadcast'})")
        model = pkg.shard.load_replicated(pkg, ctx, a.gguf, rank, world, local=a.device, stats=st)
        log(f"[rank {rank}] weight broadcast: {st.get('bytes', 0) / 1e9:.2f} GB in {st.get('seconds', 0.0):.3f}s")
    elif a.gguf:
        log(f"Loading Q4 GGUF model from {a.gguf}"); model = pkg.Q4ModelLoader.from_file(a.gguf).load(ctx)
    else:
        st = os.path.join(a.model, "consolidated.safetensors")
        log(f"Loading f32 model from {st}"); model = pkg.VoxtralModelLoader.from_file(st).load(ctx)
    log(f"Model loaded in {time.time() - t0:.2f}s")
    t_embed = pkg.TimeEmbedding(model.config.dec_dim).embed(float(a.delay))
    mel = pkg.MelSpectrogram.voxtral(ctx); pad_cfg = pkg.PadConfig.voxtral()
    chunk_cfg = pkg.ChunkConfig.voxtral().with_max_frames(a.max_mel_frames)
    rc = 0
    texts = {}
    if a.batch > 1:
        # every file -> its chunks (transcribe.rs:210-226); ALL units of this rank's share go to vox_transcribe_batch_ex in calls of <= --batch units (continuous batching
        # over decode slots); a file's line = its chunk texts joined by " " (:261-275).  Files with a failed unit fall back to the one-by-one path below.
        units = unit_table(pkg, paths, chunk_cfg)
        if a.sessions_per_gpu > 1 and a.gguf:      # vox_model_set_sessions: calls with >= 128 units per session run as that many concurrent sessions on this GPU
            model.set_sessions(a.sessions_per_gpu)
            log(f"{a.sessions_per_gpu} sessions on this GPU (model replicated device to device)")
        runner = UnitRunner(pkg, ctx, model, tokenizer, paths, t_embed)
        try:
            unit_texts = sharded_units(pkg, units, runner, a.batch, rank, world) if units else []
        finally:
            if a.sessions_per_gpu > 1 and a.gguf:
                model.set_sessions(1)
        if unit_texts is not None:      # (rank 0, or the only rank)
            texts = join_units(len(paths), units, unit_texts)
    def one(i):
        nonlocal rc
        if i in texts:
            return texts[i]
        p = paths[i]
        try:
            t1 = time.time(); text = transcribe_one(pkg, p, model, tokenizer, mel, pad_cfg, chunk_cfg, t_embed)
            log(f"{p}: {time.time() - t1:.3f}s")
        except Exception as e:      # per-utterance failure isolates to that line (empty), eval_wer.py:211-223 tolerates it
            log(f"Error transcribing {p}: {e}"); text = ""; rc = 1
        return text
    if world > 1 and a.batch > 1:
        if rank == 0:      # the units were sharded above; what is left (unreadable files, failed units) is rank 0's
            for i in range(len(paths)):
                print(one(i), flush=True)
    elif world > 1:
        lines = sharded_lines(pkg, paths, one, rank, world)
        if rank == 0:
            for text in lines:
                print(text, flush=True)
    else:
        for i in range(len(paths)):
            print(one(i), flush=True)
    if own_group:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
