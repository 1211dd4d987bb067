"""`reazonspeech.avsr` — the reference's import path (pkg/avsr/pyproject.toml maps its `src/` to this name) served by the MI355X
implementation in `reazonspeech_amd.avsr`; see reazonspeech/nemo/asr/__init__.py."""
from reazonspeech_amd.avsr import *                    # noqa: F401,F403
from reazonspeech_amd.avsr import __all__              # noqa: F401
