"""Tekken decode (tokenizer/mod.rs:170-194) and the voxtral-transcribe-compatible driver (bin/transcribe.rs)."""
import base64
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tekken(n=1200):
    vocab = [{"rank": i, "token_bytes": base64.b64encode(f" w{i}".encode()).decode(), "token_str": f" w{i}"} for i in range(n)]
    vocab[3] = {"rank": 3, "token_bytes": None, "token_str": "plain"}
    vocab[4] = {"rank": 4, "token_bytes": base64.b64encode(b"\xff\xfe").decode(), "token_str": None}      # invalid UTF-8 -> lossy
    vocab[5] = {"rank": 5, "token_str": "<ctl>", "is_control": True}
    return {"config": {"pattern": "", "num_vocab_tokens": n, "default_vocab_size": 131072, "default_num_special_tokens": 1000, "version": "v7"}, "vocab": vocab}


def test_tokenizer_decode(pkg):
    t = pkg.VoxtralTokenizer.from_json(json.dumps(_tekken()))
    assert t.vocab_size() == 131072
    assert t.decode([1, 32, 33, 999]) == ""                                   # control / streaming ids < 1000 are dropped
    assert t.decode([1000, 1001, 32, 1002]) == " w0 w1 w2"                    # id - 1000 = vocab index
    assert t.decode([1003]) == "plain" and t.decode([1004]) == "��"  # token_str fallback; lossy UTF-8
    assert t.decode([1005]) == "" and t.decode([1000 + 5000]) == ""           # control entry / out of range skipped
    assert t.decode_token(1001) == " w1" and t.decode_token(5) == "<ctl>" and t.decode_token(7) is None


def _write_wav(path, x, sr=16000, ch=1):
    with wave.open(path, "wb") as w:
        w.setnchannels(ch); w.setsampwidth(2); w.setframerate(sr)
        d = np.clip(x, -1, 1); d = (d * 32767).astype("<i2")
        if ch > 1:
            d = np.repeat(d[:, None], ch, axis=1).reshape(-1)
        w.writeframes(d.tobytes())


def test_wav_loader(pkg, tmp_path):
    cli = __import__("importlib").import_module(pkg.__name__ + ".cli")
    x = pkg.synth.synth_audio(0.5, seed=1)
    _write_wav(str(tmp_path / "m.wav"), x); _write_wav(str(tmp_path / "s.wav"), x, ch=2); _write_wav(str(tmp_path / "r.wav"), x[::2], sr=8000)
    a, sr = cli.load_wav(str(tmp_path / "m.wav")); assert sr == 16000 and a.size == x.size and np.abs(a - x).max() < 1e-4 + 1 / 32767
    b, _ = cli.load_wav(str(tmp_path / "s.wav")); assert np.abs(b - a).max() < 1e-6                 # stereo averaged to mono
    r, sr = cli.load_wav(str(tmp_path / "r.wav")); assert sr == 8000
    with pytest.raises(RuntimeError):                       # no host fallback: resampling is the HIP library's (tests/test_resample.py covers it on the GPU)
        cli.resample_to_16k(r, 8000)


def test_unit_table_splits_files_like_the_reference_cli(pkg, tmp_path):
    """cli.unit_table: the units of work of a file list from the WAV headers alone -- a file that fits --max-mel-frames is one unit, a longer one is its chunks
    (bin/transcribe.rs:210-226, audio/chunk.rs:125-157: 1200 frames = 192 000 samples, no overlap, tail chunk kept), at the length the file has at 16 kHz."""
    cli = __import__("importlib").import_module(pkg.__name__ + ".cli")
    z = np.zeros(1, np.float32)
    for name, n, sr in (("a", 80000, 16000), ("b", 480000, 16000), ("c", 192000, 16000), ("d", 192001, 16000), ("e", 120000, 8000)):
        _write_wav(str(tmp_path / f"{name}.wav"), np.resize(z, n), sr=sr)
    (tmp_path / "empty.wav").write_bytes(b"")
    paths = [str(tmp_path / f"{n}.wav") for n in ("a", "b", "missing", "c", "d", "e", "empty")]
    cfg = pkg.ChunkConfig.voxtral().with_max_frames(1200)
    u = cli.unit_table(pkg, paths, cfg)
    assert u == [(0, 0, 0, 80000), (1, 0, 0, 192000), (1, 1, 192000, 384000), (1, 2, 384000, 480000), (3, 0, 0, 192000),
                 (4, 0, 0, 192000), (4, 1, 192000, 192001), (5, 0, 0, 192000), (5, 1, 192000, 240000)]      # e: 120 000 samples at 8 kHz = 240 000 at 16 kHz
    lines = cli.join_units(7, u, ["x", "p", "", "q", "c", "d", None, "e1", "e2"])
    assert lines == {0: "x", 1: "p q", 3: "c", 5: "e1 e2"}                                                      # empty chunk texts are dropped (transcribe.rs:263-265)
    assert 4 not in lines                                                                                        # a failed unit leaves its file to the one-by-one path


@pytest.mark.gpu
def test_cli_end_to_end(pkg, tmp_path):
    """One line per input on stdout, logs on stderr (transcribe.rs:61-64,125); chunked long input; missing file -> empty line."""
    S = pkg.synth
    gguf = str(tmp_path / "m.gguf"); S.write_synthetic_gguf(gguf, S.tiny_dims(vocab=2048), seed=5)
    tok = str(tmp_path / "tekken.json"); json.dump(_tekken(1200), open(tok, "w"))
    _write_wav(str(tmp_path / "a.wav"), S.synth_audio(3.0, seed=2)); _write_wav(str(tmp_path / "b.wav"), S.synth_audio(9.0, seed=3))
    lst = tmp_path / "list.txt"; lst.write_text(f"{tmp_path / 'a.wav'}\n\n{tmp_path / 'b.wav'}\n{tmp_path / 'missing.wav'}\n")
    cmd = [sys.executable, os.path.join(ROOT, "voxtral-mini-realtime-rs_amd", "cli.py"), "--gguf", gguf, "--tokenizer", tok,
           "--audio-list", str(lst), "--max-mel-frames", "600"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    lines = r.stdout.split("\n")
    assert lines[-1] == "" and len(lines) == 4, r.stdout + r.stderr            # exactly one line per input
    assert r.returncode == 1 and "missing.wav" in r.stderr and lines[2] == ""
    assert "chunk 2/2" in r.stderr                                              # 9 s > 600 frames = 6 s -> two chunks
    assert all(w.startswith("w") for w in lines[0].split()) and all(w.startswith("w") for w in lines[1].split())
    r2 = subprocess.run(cmd[:-4] + ["--audio", str(tmp_path / "a.wav")], capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and r2.stdout.split("\n")[0] == lines[0]           # deterministic, same text
    # --batch (extension): every file's chunks are units of ONE vox_transcribe_batch_ex call (a.wav + the two chunks of b.wav); same lines in the same order
    r3 = subprocess.run(cmd + ["--batch", "4"], capture_output=True, text=True, timeout=300)
    assert r3.returncode == 1 and r3.stdout == r.stdout and "batch of 3" in r3.stderr


@pytest.mark.gpu
def test_cli_batch_makes_chunks_units_of_work_and_keeps_the_serial_lines(pkg, tmp_path):
    """with the reference's default --max-mel-frames 1200 every file longer than 12 s is split (bin/transcribe.rs:55-57,210-226); `--batch` must hand
    ALL chunks of all files to vox_transcribe_batch_ex as units (the file peak-normalised once, :207) and print exactly the one-by-one path's lines -- here 30 s, 25 s and
    13 s files among short ones, more units than one 16-row group (continuous batching), and with --batch 8 a file's chunks split over several calls."""
    import contextlib, io
    S = pkg.synth
    cli = __import__("importlib").import_module(pkg.__name__ + ".cli")
    gguf = str(tmp_path / "m.gguf"); S.write_synthetic_gguf(gguf, S.tiny_dims(vocab=2048), seed=5)
    tok = str(tmp_path / "tekken.json"); json.dump(_tekken(1200), open(tok, "w"))
    secs = [30.0, 2.0, 25.0, 13.0, 12.0, 12.01] + [1.0 + 0.45 * ((7 * i) % 11) for i in range(14)]
    wavs = []
    for i, sec in enumerate(secs):
        p = str(tmp_path / f"w{i}.wav"); _write_wav(p, (0.2 + 0.03 * (i % 5)) * S.synth_audio(sec, seed=800 + i)); wavs.append(p)      # different peaks per file
    args = ["--gguf", gguf, "--tokenizer", tok] + sum((["--audio", w] for w in wavs), [])
    n_units = len(cli.unit_table(pkg, wavs, pkg.ChunkConfig.voxtral().with_max_frames(1200)))
    assert n_units == len(secs) + 2 + 2 + 1 + 1                                     # 30 s -> 3, 25 s -> 3, 13 s -> 2, 12.01 s -> 2 units

    def run(extra):
        buf = io.StringIO(); err = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):
            rc = cli.main(args + extra)
        return rc, buf.getvalue(), err.getvalue()

    rc1, out1, err1 = run([])
    assert rc1 == 0 and "chunk 3/3" in err1 and len(out1.split("\n")) == len(secs) + 1 and all(out1.split("\n")[:-1])
    rc2, out2, err2 = run(["--batch", "1024"])
    assert rc2 == 0 and out2 == out1 and f"batch of {n_units}" in err2 and "chunk " not in err2      # no file went through the serial loop
    rc3, out3, err3 = run(["--batch", "8"])
    assert rc3 == 0 and out3 == out1 and err3.count("batch of ") == (n_units + 7) // 8
    # two concurrent sessions on the GPU (shard.SessionPool: a second context + model replica + host thread): the same lines
    rc4, out4, err4 = run(["--batch", "1024", "--sessions-per-gpu", "2"])      # (28 units: the call itself stays on one session -- 128 units per session -- the replica is made and freed)
    assert rc4 == 0 and out4 == out1 and "2 sessions on this GPU" in err4 and "chunk " not in err4


@pytest.mark.gpu
def test_cli_wide_batch_is_continuous_batching_and_keeps_the_lines(pkg, tmp_path):
    """`voxtral-transcribe --batch N` with more un-chunked files than one 16-row group: one vox_transcribe_batch call = continuous batching over decode slots (round 5);
    stdout must be, line for line and in input order, what the one-by-one run prints (transcribe.rs:112-126's contract)."""
    import contextlib, io
    S = pkg.synth
    cli = __import__("importlib").import_module(pkg.__name__ + ".cli")
    gguf = str(tmp_path / "m.gguf"); S.write_synthetic_gguf(gguf, S.tiny_dims(vocab=2048), seed=5)
    tok = str(tmp_path / "tekken.json"); json.dump(_tekken(1200), open(tok, "w"))
    wavs = []
    for i in range(37):
        p = str(tmp_path / f"w{i}.wav"); _write_wav(p, S.synth_audio(0.5 + 0.21 * ((5 * i) % 13), seed=300 + i)); wavs.append(p)
    args = ["--gguf", gguf, "--tokenizer", tok] + sum((["--audio", w] for w in wavs), [])

    def run(extra):
        buf = io.StringIO(); err = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(err):
            rc = cli.main(args + extra)
        return rc, buf.getvalue(), err.getvalue()

    rc1, out1, _ = run([])
    rc2, out2, err2 = run(["--batch", "1024"])
    assert rc1 == 0 and rc2 == 0 and out1 == out2 and len(out1.split("\n")) == 38
    assert "batch of 37" in err2


@pytest.mark.gpu
def test_wer_harness_end_to_end(pkg, tmp_path, capsys):
    """wer.main (scripts/eval_wer.py flow: manifest -> one in-process `voxtral-transcribe` run -> normalise -> WER / CER -> JSON report)."""
    S = pkg.synth
    cli = __import__("importlib").import_module(pkg.__name__ + ".cli")
    gguf = str(tmp_path / "m.gguf"); S.write_synthetic_gguf(gguf, S.tiny_dims(vocab=2048), seed=5)
    tok = str(tmp_path / "tekken.json"); json.dump(_tekken(1200), open(tok, "w"))
    wavs = []
    for i, sec in enumerate((3.0, 2.0, 4.0)):
        p = str(tmp_path / f"u{i}.wav"); _write_wav(p, S.synth_audio(sec, seed=20 + i)); wavs.append(p)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        assert cli.main(["--gguf", gguf, "--tokenizer", tok] + sum((["--audio", w] for w in wavs), [])) == 0
    texts = buf.getvalue().split("\n")[:3]
    assert all(t for t in texts)
    words = texts[1].split()
    refs = [texts[0].upper() + "!", " ".join(words[:-1] + ["zzz"]), texts[2]]              # punctuation / case only; one substituted word; exact
    man = tmp_path / "corpus.tsv"; man.write_text("".join(f"{w}\t{r}\n" for w, r in zip(wavs, refs)))
    out = tmp_path / "wer.json"
    assert pkg.wer.main(["--manifest", str(man), "--gguf", gguf, "--tokenizer", tok, "--output", str(out), "--dataset", "toy", "--batch", "2"]) == 0
    rep = json.loads(out.read_text())
    n_words = sum(len(pkg.wer.normalize_text(r).split()) for r in refs)
    assert rep["total_utterances"] == 3 and rep["aggregate_wer"] == pytest.approx(1 / n_words) and rep["utterances"][0]["wer"] == 0.0 and rep["utterances"][2]["wer"] == 0.0
    assert rep["total_audio_secs"] == pytest.approx(9.0, abs=0.01) and rep["rtf"] > 0 and "WER Evaluation Report: toy" in capsys.readouterr().out
