"""Generate tests/golden/reference_eval.json from the reference's own
/root/reference/pkg/evaluation/src/utils.py (run in the BUILD container only).  `editdistance` and
`num2words` are not installed, so the import is satisfied with a textbook Levenshtein and a
`num2words` that raises (the sample strings contain no digits); what gets pinned is the reference's
normalisation tables and the CER arithmetic."""
import importlib.util
import json
import os
import sys
import types

REF = "/root/reference/pkg/evaluation/src/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_eval.json")


def lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def main():
    ed = types.ModuleType("editdistance")
    ed.eval = lev
    nw = types.ModuleType("num2words")
    nw.num2words = lambda *a, **k: (_ for _ in ()).throw(OverflowError())
    sys.modules["editdistance"], sys.modules["num2words"] = ed, nw
    spec = importlib.util.spec_from_file_location("ref_eval_utils", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    pairs = [("今日は、良い天気ですね。", "今日は良い天気ですね"), ("「ＡＢＣ」と言った！", "abcと言った"),
             ("東京タワーに行きたい", "東京タワーへ行きたいです"), ("はい？", "はい"), ("ｘｙｚ　テスト", "xyz テスト"),
             ("完全に違う文", "まったく別の文章です"), ("同じ", "同じ")]
    cases = [{"reference": r, "prediction": p, "normalized": [m.normalize(r), m.normalize(p)],
              "cer": m.calculate_cer(r, p)} for r, p in pairs]
    with open(OUT, "w", encoding="utf-8") as fp:
        json.dump({"cases": cases}, fp, ensure_ascii=False, indent=0)
    print("wrote", OUT, len(cases))


if __name__ == "__main__":
    main()
