"""Drop-in surface of `reazonspeech.k2.asr` (pkg/k2-asr/src/__init__.py:1-4): TranscribeConfig, load_model, transcribe,
audio_from_numpy, audio_from_tensor, audio_from_path.  Additive: `transcribe_batch`."""
from .interface import TranscribeConfig
from .huggingface import load_model
from .transcribe import transcribe, transcribe_batch
from .audio import audio_from_numpy, audio_from_tensor, audio_from_path

__all__ = ["TranscribeConfig", "load_model", "transcribe", "transcribe_batch", "audio_from_numpy", "audio_from_tensor", "audio_from_path"]
