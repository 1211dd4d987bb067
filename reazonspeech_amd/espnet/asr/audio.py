"""Audio loading / normalisation of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/audio.py:1-57).  Same functions as the
nemo package's, except that a file is decoded straight to 16 kHz (`librosa.load(path, sr=SAMPLERATE)`, :39)."""
from ...nemo.asr import audio as _na
from .interface import AudioData

SAMPLERATE = 16000


def audio_from_numpy(array, samplerate):
    return AudioData(array, samplerate)


def audio_from_tensor(tensor, samplerate):
    return audio_from_numpy(tensor.numpy(), samplerate)


def audio_from_path(path):
    """decode a file and resample it to 16 kHz (mono: librosa.load's default) — audio.py:30-40"""
    a = _na.audio_from_path(path)
    return norm_audio(AudioData(a.waveform, a.samplerate))


def norm_audio(audio):
    """16 kHz mono waveform (audio.py:42-57: resample, then downmix)"""
    a = _na.norm_audio(_na.AudioData(audio.waveform, audio.samplerate))
    return AudioData(a.waveform, SAMPLERATE)
