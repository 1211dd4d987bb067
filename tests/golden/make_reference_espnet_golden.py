"""Generate tests/golden/reference_espnet.json — run in the BUILD container only.

Imports the REFERENCE's own espnet host files (/root/reference/pkg/espnet-asr/src/{interface,audio,ctc,transcribe}.py) and
runs them on the deterministic fake model of tests/espnet_fake.py: the 20 s windowing / cut-at-the-longest-gap loop
(transcribe.py:34-82), find_blank (ctc.py:29-58), find_end_of_segment (:77-86) and split_text (:88-101) are pinned by
their outputs.  Two of the reference's imports cannot be satisfied here and are stubbed: `librosa` (not needed: the inputs
are 16 kHz mono already) and `ctc_segmentation` — the stub is THIS repo's restatement
(reazonspeech_amd/espnet/asr/ctc_segmentation.py), so the golden pins everything around the aligner, not the aligner.

    python tests/golden/make_reference_espnet_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/pkg/espnet-asr/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_espnet.json")


def load_reference():
    from reazonspeech_amd.espnet.asr import ctc_segmentation as ours
    sys.modules["ctc_segmentation"] = ours
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    pkg = types.ModuleType("refespnet")
    pkg.__path__ = [REF]
    sys.modules["refespnet"] = pkg
    mods = {}
    for name in ("interface", "audio", "ctc", "transcribe"):
        spec = importlib.util.spec_from_file_location(f"refespnet.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refespnet.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def cases():
    import espnet_fake as fk
    yield "short_3s", fk.long_audio(3.0, 1)
    yield "one_window_19s", fk.long_audio(19.0, 2)
    yield "long_47s", fk.long_audio(47.3, 3)
    yield "long_90s", fk.long_audio(90.0, 4)


def main():
    import espnet_fake as fk
    ref = load_reference()
    out = {"cases": {}, "find_end_of_segment": [], "find_blank": []}
    for n, col in fk.blank_patterns():
        b = ref["ctc"].find_blank(fk.ColumnModel(col), np.zeros(n, np.float32))
        out["find_blank"].append([int(b.start), int(b.end)])
    for name, wav in cases():
        model = fk.FakeEspnetModel()
        audio = ref["interface"].AudioData(wav, 16000)
        res = ref["transcribe"].transcribe(model, audio, ref["interface"].TranscribeConfig(verbose=False))
        blank = ref["ctc"].find_blank(model, wav[:20 * 16000])
        out["cases"][name] = {
            "text": res.text, "windows": model.calls,
            "segments": [[float(s.start_seconds), float(s.end_seconds), s.text] for s in res.segments],
            "find_blank_first_window": [int(blank.start), int(blank.end)],
        }
        print(name, len(res.text), "chars,", len(res.segments), "segments, windows", model.calls)
    rng = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng.integers(1, 60))
        text = "".join(rng.choice(list("あいうえおかきくけこ。、?!,")) for _ in range(n))
        timings = np.cumsum(rng.integers(500, 12000, size=n)).tolist()
        start = int(rng.integers(0, n))
        out["find_end_of_segment"].append({"text": text, "timings": timings, "start": start,
                                           "end": int(ref["ctc"].find_end_of_segment(text, timings, start))})
    json.dump(out, open(OUT, "w"), ensure_ascii=False, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
