"""Audio helpers of `reazonspeech.k2.asr` (pkg/k2-asr/src/audio.py): constructors (:8-42), `audio_to_file` (:44-52),
`norm_audio` (resample, then down-mix: :54-68), `pad_audio` (`np.pad` with a scalar width pads BOTH ends: :70-83)."""
import numpy as np

from ...nemo.asr import audio as _na
from .interface import AudioData

SAMPLERATE = 16000


def audio_from_numpy(array, samplerate):
    return AudioData(array, samplerate)


def audio_from_tensor(tensor, samplerate):
    return audio_from_numpy(tensor.numpy(), samplerate)


def audio_from_path(path):
    """decode a file at its native rate (`librosa.load(path, sr=None)`, :41)"""
    a = _na.audio_from_path(path)
    return AudioData(a.waveform, a.samplerate)


def audio_to_file(fp, audio, format='wav'):
    """write audio to a file object (:44-52: soundfile.write); needs soundfile"""
    import soundfile
    soundfile.write(fp, audio.waveform, audio.samplerate, format=format)


def norm_audio(audio):
    a = _na.norm_audio(_na.AudioData(audio.waveform, audio.samplerate))
    return AudioData(a.waveform, SAMPLERATE)


def pad_audio(audio, seconds):
    waveform = np.pad(audio.waveform, pad_width=int(seconds * audio.samplerate), mode='constant')
    return AudioData(waveform, audio.samplerate)
