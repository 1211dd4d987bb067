"""Drop-in surface of `reazonspeech.nemo.asr`.

Reference exports (pkg/nemo-asr/src/__init__.py:1-3): TranscribeConfig, transcribe,
load_model, audio_from_numpy, audio_from_tensor, audio_from_path.  Additive (SURVEY.md
§8b): `transcribe_batch`, the batched entry point the RTFx metric is quoted on.
"""
from .interface import TranscribeConfig
from .transcribe import transcribe, transcribe_batch, load_model
from .audio import audio_from_numpy, audio_from_tensor, audio_from_path

__all__ = [
    "TranscribeConfig", "transcribe", "transcribe_batch", "load_model",
    "audio_from_numpy", "audio_from_tensor", "audio_from_path",
]
