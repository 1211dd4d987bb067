"""Generate tests/golden/reference_k2.json — run in the BUILD container only.

Imports the REFERENCE's own k2 host files (/root/reference/pkg/k2-asr/src/{interface,audio,huggingface,transcribe}.py) with
their absent third-party imports stubbed (sherpa_onnx, huggingface_hub, librosa, soundfile) and records
  * what `load_model(device, precision, language)` hands to sherpa_onnx.OfflineRecognizer.from_transducer for every valid
    (language, precision) pair — repository id, file names, the fixed keyword arguments (huggingface.py:16-83) — and the
    ValueError messages of invalid arguments (:37-38, :61-62);
  * `transcribe()` (transcribe.py:10-45) on the deterministic fake recogniser of tests/k2_fake.py: padded sample counts the
    recogniser sees, subwords, text, and the long-audio warning.

    python tests/golden/make_reference_k2_golden.py
"""
import importlib.util
import json
import os
import sys
import types
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/pkg/k2-asr/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_k2.json")


def load_reference(calls):
    sherpa = types.ModuleType("sherpa_onnx")

    class OfflineRecognizer:
        @staticmethod
        def from_transducer(**kw):
            calls.append(kw)
            return "recognizer"
    sherpa.OfflineRecognizer = OfflineRecognizer
    sys.modules["sherpa_onnx"] = sherpa
    hf = types.ModuleType("huggingface_hub")
    hf.utils = types.SimpleNamespace(LocalEntryNotFoundError=type("LocalEntryNotFoundError", (Exception,), {}))
    hf.snapshot_download = lambda repo, local_files_only=False: "/cache/" + repo
    sys.modules["huggingface_hub"] = hf
    sys.modules.setdefault("librosa", types.ModuleType("librosa"))
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    pkg = types.ModuleType("refk2")
    pkg.__path__ = [REF]
    sys.modules["refk2"] = pkg
    mods = {}
    for name in ("interface", "audio", "huggingface", "transcribe"):
        spec = importlib.util.spec_from_file_location(f"refk2.{name}", os.path.join(REF, name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"refk2.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def main():
    import k2_fake as fk
    calls = []
    ref = load_reference(calls)
    out = {"load_model": {}, "errors": {}, "transcribe": {}}
    for lang in ("ja", "ja-en", "ja-en-mls-5k"):
        for prec in ("fp32", "int8", "int8-fp32"):
            calls.clear()
            ref["huggingface"].load_model(device="cuda", precision=prec, language=lang)
            out["load_model"][f"{lang}|{prec}"] = calls[0]
    for kw in ({"language": "fr"}, {"precision": "fp16"}, {"language": "xx", "precision": "yy"}):
        try:
            ref["huggingface"].load_model(**kw)
        except ValueError as e:
            out["errors"][json.dumps(kw, sort_keys=True)] = str(e)
    out["constants"] = {"PAD_SECONDS": ref["transcribe"].PAD_SECONDS, "TOO_LONG_SECONDS": ref["transcribe"].TOO_LONG_SECONDS}
    for name, secs, seed in (("short", 2.0, 1), ("ten", 10.0, 2), ("long", 29.5, 3)):
        model = fk.FakeRecognizer()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res = ref["transcribe"].transcribe(model, ref["interface"].AudioData(fk.audio(secs, seed), 16000))
        out["transcribe"][name] = {"seen": model.seen, "text": res.text, "subwords": [[s.token, s.seconds] for s in res.subwords],
                                   "warnings": [str(x.message) for x in w]}
    json.dump(out, open(OUT, "w"), ensure_ascii=False, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
