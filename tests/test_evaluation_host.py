"""CPU: the evaluation hook — CER arithmetic pinned to the reference's
utils.py golden, batching / output logic with a stand-in model."""
import json
import os

import numpy as np
import pytest

from reazonspeech_amd import evaluation as E

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_eval.json")


def test_normalize_and_cer_match_reference():
    with open(GOLD, encoding="utf-8") as fp:
        gold = json.load(fp)
    for c in gold["cases"]:
        assert [E.normalize(c["reference"]), E.normalize(c["prediction"])] == c["normalized"]
        got = E.calculate_cer(c["reference"], c["prediction"])
        assert got == c["cer"]


def test_edit_distance_properties():
    assert E.edit_distance("", "abc") == 3 and E.edit_distance("abc", "") == 3
    assert E.edit_distance("kitten", "sitting") == 3
    assert E.edit_distance("同じ", "同じ") == 0
    assert E.edit_distance("ab", "ba") == 2


class _StubEvaluator(E.RSAmdEvaluator):
    """predicts the reference text of every other example, nothing for the rest"""

    def __init__(self, **kw):
        super().__init__(model=object(), **kw)
        self.batches = []

    def _evaluate(self, example, **kw):
        return {"prediction": example["hint"]}

    def _evaluate_batch(self, batch, **kw):
        self.batches.append(len(batch["audio"]))
        return {"predictions": [a["hint"] for a in batch["audio"]]}


def _rows(n):
    return [{"audio": {"array": np.zeros(8, np.float32), "sampling_rate": 16000, "hint": "あいう" if i % 2 else "あ"},
             "hint": "あいう" if i % 2 else "あ", "text": "あいう"} for i in range(n)]


def test_evaluate_batches_and_reports(tmp_path, capsys):
    ev = _StubEvaluator(batch_size=4, text_column="text", output_file=str(tmp_path / "o.jsonl"))
    rows = ev.evaluate(_rows(10))
    assert ev.batches == [4, 4, 2]
    assert len(rows) == 10 and all("prediction" in r and "distance" in r for r in rows)
    assert sum(r["distance"] for r in rows) == 5 * 2 and sum(r["length"] for r in rows) == 30
    assert "CER: 33.33%" in capsys.readouterr().out
    lines = open(tmp_path / "o.jsonl", encoding="utf-8").read().strip().split("\n")
    assert len(lines) == 10 and "audio" not in json.loads(lines[0])
    # unbatched path
    ev2 = _StubEvaluator(batch_size=None)
    assert len(ev2.evaluate(_rows(3))) == 3 and ev2.batches == []
    with pytest.raises(ValueError):
        E.RSAmdEvaluator(model=object()).evaluate()
