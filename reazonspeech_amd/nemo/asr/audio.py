"""Host-side audio preparation.

Behaviour restated from pkg/nemo-asr/src/audio.py:
  * constructors `audio_from_numpy/tensor/path`   (audio.py:8-42)
  * `norm_audio`: resample to 16 kHz FIRST, then down-mix (audio.py:54-68)
  * `pad_audio`: `np.pad` with a scalar width pads BOTH ends (audio.py:70-83)

librosa / soundfile are not part of this image, so resampling is done with a
polyphase filter from scipy and file decoding with soundfile-if-present, else scipy's
WAV reader.  On the GPU path `pad_audio` is never materialised: the front-end kernel
reads the raw samples with an offset and treats everything outside as zero
(`rs_frontend_logmel(..., pad_left, pad_right, ...)`), which is bit-identical to padding
first because the padding value is exactly 0.0.
"""
from fractions import Fraction

import numpy as np

from .interface import AudioData

SAMPLERATE = 16000


def audio_from_numpy(array, samplerate):
    """Wrap a numpy waveform (shape [L] or [channels, L]) — audio.py:8-18."""
    return AudioData(array, samplerate)


def audio_from_tensor(tensor, samplerate):
    """Wrap a CPU torch tensor — audio.py:20-30 (`tensor.numpy()`)."""
    return audio_from_numpy(tensor.numpy(), samplerate)


def audio_from_path(path):
    """Decode an audio file at its native sample rate — audio.py:32-42
    (`librosa.load(path, sr=None)`: float32, mono down-mix, native rate)."""
    try:
        import soundfile  # optional, not in this image
        data, samplerate = soundfile.read(path, dtype="float32", always_2d=True)
        data = data.T  # [channels, L]
    except ImportError:
        from scipy.io import wavfile
        samplerate, raw = wavfile.read(path)
        if raw.dtype.kind == "i":
            data = raw.astype(np.float32) / float(np.iinfo(raw.dtype).max + 1)
        elif raw.dtype.kind == "u":  # 8-bit PCM is unsigned
            data = (raw.astype(np.float32) - 128.0) / 128.0
        else:
            data = raw.astype(np.float32)
        data = data.T if data.ndim > 1 else data[None, :]
    # librosa.load(mono=True) averages channels
    mono = data.mean(axis=0) if data.shape[0] > 1 else data[0]
    return audio_from_numpy(np.ascontiguousarray(mono, dtype=np.float32), int(samplerate))


def audio_to_file(fp, audio, format="wav"):
    """Write a waveform (audio.py:44-52).  WAV only without soundfile."""
    try:
        import soundfile
        soundfile.write(fp, audio.waveform, audio.samplerate, format=format)
    except ImportError:
        if format != "wav":
            raise RuntimeError("only WAV output is supported without soundfile")
        from scipy.io import wavfile
        wavfile.write(fp, audio.samplerate, np.asarray(audio.waveform, dtype=np.float32))


def _resample(waveform, orig_sr, target_sr):
    """Band-limited resampling along the last axis (stand-in for `librosa.resample`,
    audio.py:64-65; librosa's default is a soxr/kaiser windowed-sinc, so sample values
    differ in the last bits — this is host pre-processing outside the parity contract)."""
    from scipy.signal import resample_poly
    ratio = Fraction(int(target_sr), int(orig_sr))
    out = resample_poly(np.asarray(waveform, dtype=np.float64), ratio.numerator,
                        ratio.denominator, axis=-1)
    return out.astype(np.float32)


def norm_audio(audio):
    """16 kHz mono float waveform; order of operations as audio.py:62-68."""
    waveform = audio.waveform
    if audio.samplerate != SAMPLERATE:
        waveform = _resample(waveform, audio.samplerate, SAMPLERATE)
    if len(waveform.shape) > 1:
        waveform = np.mean(waveform, axis=0)  # librosa.to_mono
    return AudioData(waveform, SAMPLERATE)


def pad_audio(audio, seconds):
    """Zero-pad `int(seconds * samplerate)` samples on both sides (audio.py:80-82)."""
    width = int(seconds * audio.samplerate)
    waveform = np.pad(audio.waveform, pad_width=width, mode="constant")
    return AudioData(waveform, audio.samplerate)
