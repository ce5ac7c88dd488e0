"""WER / CER harness (voxtral wer.py, restating scripts/eval_wer.py + the jiwer semantics it relies on): known-answer tests."""
import json

import pytest


def test_normalize_text(pkg):
    W = pkg.wer
    assert W.normalize_text("Hello,   World! It's 3 o'clock -- fine.") == "hello world its 3 oclock fine"      # eval_wer.py:93-97
    assert W.normalize_text("  ¿Qué?  «ok»…  ") == "qué ok"                                                     # every Unicode P* category
    assert W.normalize_text("a\t\nb") == "a b" and W.normalize_text("") == ""


def test_wer_cer_known_answers(pkg):
    W = pkg.wer
    assert W.wer("the cat sat", "the cat sat") == 0.0
    assert W.wer("the cat sat on the mat", "the cat sit on mat") == pytest.approx(2 / 6)        # 1 substitution + 1 deletion
    assert W.wer("a b c", "a x b c y") == pytest.approx(2 / 3)                                  # 2 insertions: WER can exceed intuition (> 0.5)
    assert W.wer("a", "b c d") == pytest.approx(3.0)                                            # 1 sub + 2 ins over 1 word
    assert sum(W.edit_counts("a b c d".split(), "a c d e".split())[1:]) == 2 and W.edit_counts([], ["x"]) == (0, 0, 0, 1)
    # aggregate over sentences = pooled edits / pooled reference words, NOT the mean of sentence rates (jiwer)
    assert W.wer(["a b c d", "e"], ["a b c d", "x"]) == pytest.approx(1 / 5)
    assert W.cer("abc def", "abd def") == pytest.approx(1 / 7)                                  # spaces count as characters
    assert W.cer(["ab", "cd"], ["ab", "c"]) == pytest.approx(1 / 4)
    with pytest.raises(ValueError):
        W.wer(["a", ""], ["a", "b"])                                                            # jiwer: empty reference is an error
    assert W.wer(["a", ""], ["a", "b"], skip_empty=True) == 0.0
    with pytest.raises(ValueError):
        W.wer(["a"], ["a", "b"])


def test_score_report_and_manifest(pkg, tmp_path):
    W = pkg.wer
    refs = ["Mary had a little lamb.", "Its fleece was white as snow!", ""]
    hyps = ["mary had a little lamb", "its fleece was white as snow", ""]
    r = W.score(["u0", "u1", "u2"], refs, hyps, [2.0, 3.0, 1.0], "toy", wall_secs=0.6, delay=6)
    assert r.aggregate_wer == 0.0 and r.aggregate_cer == 0.0 and r.total_utterances == 3 and r.rtf == pytest.approx(0.1)
    r2 = W.score(["u0", "u1"], refs[:2], ["mary had a little lamp"], [2.0, 3.0], "toy", 1.0, 6)          # second line missing -> empty hypothesis
    assert r2.utterances[0].wer == pytest.approx(1 / 5) and r2.utterances[1].wer == 1.0
    assert r2.aggregate_wer == pytest.approx((1 + 6) / 11)
    assert "WER:         63.64%" in W.format_report(r2) and "Delay:       6 tokens (480ms)" in W.format_report(r2)
    p = tmp_path / "r.json"; W.save_report(r2, str(p))
    d = json.loads(p.read_text()); assert d["dataset"] == "toy" and len(d["utterances"]) == 2 and set(d["utterances"][0]) == {"id", "reference", "hypothesis", "wer", "audio_duration_secs"}
    m = tmp_path / "m.tsv"; m.write_text("a.wav\tHello there\nid7\tb.wav\tSecond line\n\n")
    assert W.load_manifest(str(m)) == [("utt_0", "a.wav", "Hello there"), ("id7", "b.wav", "Second line")]
    j = tmp_path / "m.jsonl"; j.write_text('{"id": "x", "audio": "c.wav", "transcription": "T"}\n')
    assert W.load_manifest(str(j)) == [("x", "c.wav", "T")]
