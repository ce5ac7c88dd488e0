"""WER / CER evaluation harness (reference: scripts/eval_wer.py:56-326) without the `jiwer` / `datasets` dependencies -- CPU-side tooling around
the accelerated path.  The reference writes the corpus to WAVs, runs `voxtral-transcribe --audio-list` ONCE
(model loads once, one stdout line per file) and scores the lines; here the corpus is a manifest (TSV `wav<TAB>reference` or JSONL
{"audio": .., "text": ..}; a `datasets`-style iterable of dicts works too) and the transcriptions come from `cli.main`'s in-process twin.

jiwer semantics restated (jiwer 3.x, the version the reference's script resolves):
  * normalize_text (eval_wer.py:93-97): RemovePunctuation (every character whose Unicode category starts with "P"), ToLowerCase, Strip,
    RemoveMultipleSpaces (runs of >= 2 white-space characters -> one space);
  * wer(refs, hyps): sentences -> words on white space; (S + D + I) over ALL sentences / total reference words (not the mean of sentence WERs);
  * cer(refs, hyps): the same on characters of the stripped sentences, spaces included;
  * an empty reference sentence is an error in jiwer; eval_wer.py scores it 0 / 1 per utterance (:249-252) and would raise in the aggregate --
    `aggregate()` therefore takes skip_empty (default False = raise like jiwer).
"""
from __future__ import annotations

import json
import re
import sys
import time
import unicodedata
from dataclasses import dataclass, field, asdict


def remove_punctuation(text: str) -> str:
    return "".join(c for c in text if not unicodedata.category(c).startswith("P"))


def normalize_text(text: str) -> str:
    """eval_wer.py:93-97."""
    return re.sub(r"\s\s+", " ", remove_punctuation(text).lower().strip())


def edit_counts(ref: list, hyp: list):
    """Levenshtein alignment counts (hits, substitutions, deletions, insertions) between two token lists (unit costs; among optimal
    alignments the counts of S + D + I are what WER needs -- their sum is the distance)."""
    n, m = len(ref), len(hyp)
    if n == 0:
        return 0, 0, 0, m
    if m == 0:
        return 0, 0, n, 0
    # dp over (cost, subs, dels, ins) with tuple comparison on cost first: deterministic
    prev = [(j, 0, 0, j) for j in range(m + 1)]
    for i in range(1, n + 1):
        cur = [(i, 0, i, 0)] + [None] * m
        ri = ref[i - 1]
        for j in range(1, m + 1):
            if ri == hyp[j - 1]:
                best = prev[j - 1]
            else:
                c = prev[j - 1]; best = (c[0] + 1, c[1] + 1, c[2], c[3])
            d = prev[j]; cand = (d[0] + 1, d[1], d[2] + 1, d[3])
            if cand[0] < best[0]:
                best = cand
            ins = cur[j - 1]; cand = (ins[0] + 1, ins[1], ins[2], ins[3] + 1)
            if cand[0] < best[0]:
                best = cand
            cur[j] = best
        prev = cur
    cost, s, d, i_ = prev[m]
    return n - s - d, s, d, i_


def _as_list(x):
    return [x] if isinstance(x, str) else list(x)


def _rate(refs, hyps, tokenize, skip_empty):
    refs, hyps = _as_list(refs), _as_list(hyps)
    if len(refs) != len(hyps):
        raise ValueError(f"After applying the transforms on the reference and hypothesis sentences, their lengths must match: {len(refs)} vs {len(hyps)}")
    errs = total = 0
    for r, h in zip(refs, hyps):
        rt, ht = tokenize(r), tokenize(h)
        if not rt:
            if skip_empty:
                continue
            raise ValueError("one or more references are empty strings")
        _, s, d, i = edit_counts(rt, ht)
        errs += s + d + i; total += len(rt)
    return errs / total if total else 0.0


def wer(refs, hyps, skip_empty: bool = False) -> float:
    """jiwer.wer with its default transform (RemoveMultipleSpaces, Strip, split on white space)."""
    return _rate(refs, hyps, lambda s: re.sub(r"\s\s+", " ", s).strip().split(), skip_empty)


def cer(refs, hyps, skip_empty: bool = False) -> float:
    """jiwer.cer with its default transform (Strip, characters)."""
    return _rate(refs, hyps, lambda s: list(s.strip()), skip_empty)


@dataclass
class UtteranceResult:            # eval_wer.py:56-62
    id: str
    reference: str
    hypothesis: str
    wer: float
    audio_duration_secs: float


@dataclass
class EvalReport:                 # eval_wer.py:65-77
    dataset: str
    total_utterances: int
    successful: int
    failed: int
    aggregate_wer: float
    aggregate_cer: float
    total_audio_secs: float
    total_wall_secs: float
    rtf: float
    delay_tokens: int
    utterances: list = field(default_factory=list)


def score(ids, references, hypotheses, durations, dataset: str, wall_secs: float, delay: int, skip_empty: bool = True, log=None) -> EvalReport:
    """eval_wer.py:232-293 (phase 5 + aggregation).  `hypotheses` may be shorter than `references` (missing lines count as empty)."""
    results, refs_n, hyps_n = [], [], []
    n = len(references)
    for i in range(n):
        hyp = hypotheses[i].strip() if i < len(hypotheses) else ""
        rn, hn = normalize_text(references[i]), normalize_text(hyp)
        u = wer(rn, hn) if rn else (0.0 if not hn else 1.0)                                   # :249-252
        results.append(UtteranceResult(str(ids[i]), references[i], hyp, u, float(durations[i])))
        refs_n.append(rn); hyps_n.append(hn)
        if log:
            log(f"  [{i + 1}/{n}] {ids[i]} ({durations[i]:.1f}s) " + (f"WER={u:.0%}" if u > 0 else "OK"))
    total_audio = float(sum(durations))
    return EvalReport(dataset, n, len(results), 0, wer(refs_n, hyps_n, skip_empty) if refs_n else 0.0, cer(refs_n, hyps_n, skip_empty) if refs_n else 0.0,
                      total_audio, wall_secs, wall_secs / total_audio if total_audio > 0 else 0.0, delay, results)


def format_report(r: EvalReport) -> str:          # eval_wer.py:296-313
    bar = "=" * 60
    return "\n".join([f"\n{bar}", f"WER Evaluation Report: {r.dataset}", bar,
                      f"  Utterances:  {r.successful}/{r.total_utterances} ({r.failed} failed)", f"  WER:         {r.aggregate_wer:.2%}",
                      f"  CER:         {r.aggregate_cer:.2%}", f"  Audio:       {r.total_audio_secs:.1f}s ({r.total_audio_secs / 60:.1f} min)",
                      f"  Wall time:   {r.total_wall_secs:.1f}s ({r.total_wall_secs / 60:.1f} min)", f"  RTF:         {r.rtf:.2f}x",
                      f"  Delay:       {r.delay_tokens} tokens ({r.delay_tokens * 80}ms)", bar])


def save_report(r: EvalReport, path: str):        # eval_wer.py:316-342
    d = asdict(r)
    with open(path, "w") as f:
        json.dump(d, f, indent=2)


def load_manifest(path: str):
    """[(id, wav_path, reference_text)] from a TSV (`wav<TAB>text`, optional leading id column) or JSONL ({"audio"|"path"|"wav": .., "text"|"transcription": ..})."""
    items = []
    with open(path, encoding="utf-8") as f:
        for k, line in enumerate(f):
            line = line.rstrip("\n")
            if not line.strip():
                continue
            if line.lstrip().startswith("{"):
                o = json.loads(line)
                wav = o.get("audio") or o.get("path") or o.get("wav"); txt = o.get("text") if o.get("text") is not None else o.get("transcription", "")
                items.append((str(o.get("id", f"utt_{k}")), wav, txt))
            else:
                parts = line.split("\t")
                if len(parts) == 2:
                    items.append((f"utt_{k}", parts[0], parts[1]))
                elif len(parts) >= 3:
                    items.append((parts[0], parts[1], parts[2]))
                else:
                    raise ValueError(f"{path}:{k + 1}: expected `wav<TAB>reference`")
    return items


def main(argv=None):
    """`python -m ... wer --manifest corpus.tsv --gguf model.gguf --tokenizer tekken.json [--delay 6] [--batch 1024] [--sessions-per-gpu 2]`: the eval_wer.py flow
    (:345-399) with the model loaded once in-process; prints the report, writes wer_<dataset>.json."""
    import argparse, io, contextlib, os, wave
    ap = argparse.ArgumentParser(description="WER evaluation for Voxtral (MI355X HIP path)")
    ap.add_argument("--manifest", required=True); ap.add_argument("--dataset", default="manifest")
    ap.add_argument("--gguf"); ap.add_argument("--model"); ap.add_argument("--tokenizer")
    ap.add_argument("--delay", type=int, default=6); ap.add_argument("--output"); ap.add_argument("--limit", type=int)
    ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1, help="shard the corpus over N GPUs (cli --gpus N: one process per GPU, transcriptions gathered in corpus order)")
    ap.add_argument("--sessions-per-gpu", type=int, default=1, help="concurrent sessions per GPU (cli --sessions-per-gpu: context + model replica + host thread each; same transcriptions)")
    a = ap.parse_args(argv)
    if not a.gguf and not a.model:
        ap.error("Either --gguf or --model is required")
    items = load_manifest(a.manifest)
    if a.limit:
        items = items[:a.limit]
    durations = []
    for _, wav, _ in items:
        with wave.open(wav, "rb") as w:
            durations.append(w.getnframes() / float(w.getframerate()))
    from . import cli
    args = ["--delay", str(a.delay), "--device", str(a.device), "--batch", str(a.batch)] + (["--gpus", str(a.gpus)] if a.gpus > 1 else [])
    args += ["--sessions-per-gpu", str(a.sessions_per_gpu)] if a.sessions_per_gpu > 1 else []
    args += (["--gguf", a.gguf] if a.gguf else ["--model", a.model]) + (["--tokenizer", a.tokenizer] if a.tokenizer else [])
    for _, wav, _ in items:
        args += ["--audio", wav]
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        rc = cli.main(args)
    wall = time.time() - t0
    out = buf.getvalue(); out = out[:-1] if out.endswith("\n") else out
    hyps = out.split("\n") if out else []                       # one line per input, empty lines kept (eval_wer.py:214-222)
    if len(hyps) != len(items):
        print(f"WARNING: Expected {len(items)} transcriptions, got {len(hyps)}", file=sys.stderr)
    rep = score([i for i, _, _ in items], [t for _, _, t in items], hyps, durations, a.dataset, wall, a.delay, log=lambda s: print(s, file=sys.stderr))
    print(format_report(rep))
    save_report(rep, a.output or f"wer_{a.dataset}.json")
    return rc


if __name__ == "__main__":
    sys.exit(main())
