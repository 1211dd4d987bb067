"""Generate tests/golden/reference_host.json — run in the BUILD container only.

Imports the reference's own in-tree files for the host-side parts of the hot path
(/root/reference/pkg/nemo-asr/src/{decode,interface,writer}.py — pure Python, no NeMo needed)
and records their outputs on seeded inputs.  These are the only pieces of the path whose
behaviour the reference pins in-tree, so they are pinned exactly.

    python tests/golden/make_reference_golden.py
"""
import importlib.util
import io
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/pkg/nemo-asr/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_host.json")


def load_reference():
    pkg = types.ModuleType("refasr")
    pkg.__path__ = [REF]
    sys.modules["refasr"] = pkg
    mods = {}
    for name in ("interface", "decode", "writer"):
        spec = importlib.util.spec_from_file_location(f"refasr.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refasr.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


class FakeTokenizer:
    """stands in for model.tokenizer (NeMo SentencePiece wrapper): ids -> text by table"""

    def __init__(self, pieces):
        self.pieces = pieces

    def ids_to_text(self, ids):
        text = "".join(self.pieces[i] for i in ids).replace("▁", " ")
        return text[1:] if text.startswith(" ") else text


class FakeModel:
    def __init__(self, pieces):
        self.tokenizer = FakeTokenizer(pieces)


class FakeSeq(list):
    def tolist(self):
        return list(self)


class FakeHyp:
    def __init__(self, y, ts):
        self.y_sequence = FakeSeq(y)
        self.timestamp = ts


def main():
    ref = load_reference()
    pieces = ["▁", "。", "、", "?", "!", ",", "今日", "は", "良い", "天気", "です", "ね", "▁明日", "も",
              "晴れ", "る", "でしょう", "か", "はい", "そう", "思い", "ます", "東京", "大阪", "行き", "たい"]
    rng = np.random.default_rng(20260925)
    cases = []
    for case in range(24):
        n = int(rng.integers(0, 40))
        ids = [int(x) for x in rng.integers(0, len(pieces), size=n)]
        # ALSD-style steps: frame + emitted-so-far + 1, frames non-decreasing with random gaps
        gaps = rng.choice([0, 0, 1, 1, 2, 3, 9, 14], size=n)
        frames = np.cumsum(gaps).tolist()
        steps = [int(f + i + 1) for i, f in enumerate(frames)]
        hyp = FakeHyp([len(pieces)] + ids, steps)
        res = ref["decode"].decode_hypothesis(FakeModel(pieces), hyp)
        writers = {}
        for ext in ("vtt", "srt", "ass", "json", "tsv", None):
            fp = io.StringIO()
            w = ref["writer"].get_writer(fp, ext)
            w.write_header()
            for seg in res.segments:
                w.write(seg)
            writers[str(ext)] = fp.getvalue()
        cases.append({
            "ids": ids, "steps": steps, "frames": [int(f) for f in frames], "blank": len(pieces),
            "text": res.text,
            "subwords": [[sw.seconds, sw.token_id, sw.token] for sw in res.subwords],
            "segments": [[s.start_seconds, s.end_seconds, s.text] for s in res.segments],
            "writers": writers,
        })
    consts = {k: getattr(ref["decode"], k) for k in ("PAD_SECONDS", "SECONDS_PER_STEP", "SUBWORDS_PER_SEGMENTS",
                                                     "PHONEMIC_BREAK")}
    consts["TOKEN_EOS"] = sorted(ref["decode"].TOKEN_EOS)
    consts["TOKEN_COMMA"] = sorted(ref["decode"].TOKEN_COMMA)
    # get_writer quirk (writer.py:160-166): splitext keeps the dot, so "x.vtt" without --to -> TextWriter
    class Named(io.StringIO):
        name = "x.vtt"
    quirk = type(ref["writer"].get_writer(Named())).__name__
    cfg = ref["interface"].TranscribeConfig()
    with open(OUT, "w", encoding="utf-8") as fp:
        json.dump({"pieces": pieces, "cases": cases, "consts": consts, "writer_for_x_vtt": quirk,
                   "config_defaults": {"verbose": cfg.verbose, "raw_hypothesis": cfg.raw_hypothesis}},
                  fp, ensure_ascii=False, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
