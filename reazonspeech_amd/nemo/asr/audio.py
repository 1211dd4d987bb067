"""Host-side audio preparation.

Behaviour restated from pkg/nemo-asr/src/audio.py:
  * constructors `audio_from_numpy/tensor/path`   (audio.py:8-42)
  * `norm_audio`: resample to 16 kHz FIRST, then down-mix (audio.py:54-68)
  * `pad_audio`: `np.pad` with a scalar width pads BOTH ends (audio.py:70-83)

librosa / soundfile / soxr are not part of this image.  Resampling uses soxr or librosa when they are importable
(then it IS the reference's `librosa.resample`, res_type "soxr_hq") and otherwise a polyphase windowed-sinc filter built
to soxr HQ's published specification (pass band to 0.913 of the lower Nyquist, >= 120 dB rejection from the Nyquist
up): measured against analytic tones it is within -135 dB in the pass band and below -140 dB in the stop band
(tests/test_host_reference_parity.py), i.e. it differs from soxr only inside the transition band and at the 24-bit
level elsewhere.  File decoding uses soundfile (any libsndfile format, like librosa.load) or audioread when present,
else scipy's WAV reader.  On the GPU path `pad_audio` is never materialised: the front-end kernel
reads the raw samples with an offset and treats everything outside as zero
(`rs_frontend_logmel(..., pad_left, pad_right, ...)`), which is bit-identical to padding
first because the padding value is exactly 0.0.
"""
import functools
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
    (`librosa.load(path, sr=None)`: float32, mono down-mix, native rate; librosa tries soundfile first, then
    audioread — so does this, and falls back to scipy's WAV reader when neither is installed)."""
    data = None
    try:
        import soundfile  # optional, not in this image
        try:
            data, samplerate = soundfile.read(path, dtype="float32", always_2d=True)
            data = data.T  # [channels, L]
        except (RuntimeError, OSError):
            # libsndfile cannot decode this container (mp3 / m4a ...): librosa.load catches that and retries with
            # audioread — so does this
            data = None
    except ImportError:
        pass
    if data is None:
        try:
            import audioread  # optional: mp3 / m4a through the system decoders, librosa's second choice
            with audioread.audio_open(path) as f:
                samplerate, ch = f.samplerate, f.channels
                raw = np.frombuffer(b"".join(f), dtype="<i2")
            data = (raw.astype(np.float32) / 32768.0).reshape(-1, ch).T
        except ImportError:
            pass
    if data is None:
        from scipy.io import wavfile
        try:
            samplerate, raw = wavfile.read(path)
        except ValueError as e:
            raise RuntimeError(f"{path}: not a WAV file, and neither soundfile nor audioread is installed to decode "
                               f"other containers (the reference uses librosa.load, pkg/nemo-asr/src/audio.py:41)") from e
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


SOXR_HQ_PASSBAND = 0.913        # soxr "HQ" recipe: pass band end as a fraction of the lower Nyquist frequency
SOXR_HQ_REJECTION_DB = 125.0    # 20-bit precision


@functools.lru_cache(maxsize=16)
def _hq_filter(up, down):
    """linear-phase Kaiser-windowed sinc at the internal rate (orig * up = target * down): transition band
    [0.913, 1] x the lower Nyquist, >= 125 dB down beyond it"""
    from scipy.signal import firwin, kaiserord
    nyq = 0.5 / max(up, down)                         # lower Nyquist in cycles / sample of the internal rate
    numtaps, beta = kaiserord(SOXR_HQ_REJECTION_DB, (1.0 - SOXR_HQ_PASSBAND) * nyq / 0.5)
    return firwin(numtaps | 1, (SOXR_HQ_PASSBAND + 1.0) / 2 * nyq / 0.5, window=("kaiser", beta))


def _resample(waveform, orig_sr, target_sr):
    """Band-limited resampling along the last axis: `librosa.resample(..., res_type="soxr_hq")` (audio.py:64-65) when
    soxr / librosa are importable, else the polyphase filter of `_hq_filter` (same specification; see the module
    docstring for the measured difference)."""
    x = np.asarray(waveform)
    try:
        import soxr  # what librosa.resample calls for its default res_type
        y = soxr.resample(np.ascontiguousarray(x.T, dtype=np.float32), orig_sr, target_sr, quality="HQ")
        return np.ascontiguousarray(y.T, dtype=np.float32)
    except ImportError:
        pass
    from scipy.signal import resample_poly
    ratio = Fraction(int(target_sr), int(orig_sr))
    out = resample_poly(x.astype(np.float64), ratio.numerator, ratio.denominator, axis=-1,
                        window=_hq_filter(ratio.numerator, ratio.denominator))
    n_out = int(np.ceil(x.shape[-1] * target_sr / orig_sr))     # librosa's output length
    return out[..., :n_out].astype(np.float32)


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
