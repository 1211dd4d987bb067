"""`reazonspeech.espnet.asr` — the reference's import path (pkg/espnet-asr/pyproject.toml:19-20 maps its `src/` to this
name) served by the MI355X implementation in `reazonspeech_amd.espnet.asr`; see reazonspeech/nemo/asr/__init__.py."""
import importlib
import sys

from reazonspeech_amd.espnet.asr import *                    # noqa: F401,F403
from reazonspeech_amd.espnet.asr import __all__              # noqa: F401

for _sub in ("interface", "audio", "ctc", "transcribe", "writer", "cli"):
    sys.modules[__name__ + "." + _sub] = importlib.import_module("reazonspeech_amd.espnet.asr." + _sub)
del _sub
