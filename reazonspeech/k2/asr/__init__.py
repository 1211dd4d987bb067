"""`reazonspeech.k2.asr` — the reference's import path (pkg/k2-asr/pyproject.toml maps its `src/` to this name) served by the
MI355X implementation in `reazonspeech_amd.k2.asr`; see reazonspeech/nemo/asr/__init__.py."""
import importlib
import sys

from reazonspeech_amd.k2.asr import *                    # noqa: F401,F403
from reazonspeech_amd.k2.asr import __all__              # noqa: F401

for _sub in ("interface", "audio", "huggingface", "transcribe"):
    sys.modules[__name__ + "." + _sub] = importlib.import_module("reazonspeech_amd.k2.asr." + _sub)
del _sub
