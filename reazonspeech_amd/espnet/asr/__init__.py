"""Drop-in surface of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/__init__.py:1-3): TranscribeConfig, transcribe,
load_model, audio_from_numpy, audio_from_tensor, audio_from_path.  Additive: `transcribe_batch`."""
from .interface import TranscribeConfig
from .transcribe import transcribe, transcribe_batch, load_model
from .audio import audio_from_numpy, audio_from_tensor, audio_from_path

__all__ = ["TranscribeConfig", "transcribe", "transcribe_batch", "load_model", "audio_from_numpy", "audio_from_tensor",
           "audio_from_path"]
