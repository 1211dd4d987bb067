"""`reazonspeech.nemo.asr` — the reference's import path (pkg/nemo-asr/pyproject.toml:16-17 maps its `src/` to this
name; exports: pkg/nemo-asr/src/__init__.py:1-3) served by the MI355X implementation in `reazonspeech_amd.nemo.asr`.

`reazonspeech` and `reazonspeech.nemo` stay namespace packages (no `__init__.py`), exactly like the reference's
split distributions, so the sibling packages (`reazonspeech.espnet.asr`, `reazonspeech.k2.asr`, ...) of an existing
installation keep importing.  Submodules are aliased, so `reazonspeech.nemo.asr.cli:main` (the console script,
pyproject.toml:19-20), `.interface`, `.decode`, `.audio`, `.writer` and `.transcribe` resolve to the same objects.
"""
import importlib
import sys

from reazonspeech_amd.nemo.asr import *                    # noqa: F401,F403
from reazonspeech_amd.nemo.asr import __all__              # noqa: F401

for _sub in ("interface", "audio", "decode", "transcribe", "writer", "cli"):
    # sys.modules only: the attribute `transcribe` must stay the FUNCTION, as in the reference's __init__
    sys.modules[__name__ + "." + _sub] = importlib.import_module("reazonspeech_amd.nemo.asr." + _sub)
del _sub
