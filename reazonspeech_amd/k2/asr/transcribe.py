"""`transcribe()` of `reazonspeech.k2.asr` (pkg/k2-asr/src/transcribe.py:7-45): normalise to 16 kHz mono, pad PAD_SECONDS of
silence on both sides, warn above TOO_LONG_SECONDS, decode one stream, pair tokens with timestamps."""
import warnings

from .interface import TranscribeConfig, TranscribeResult, Subword
from .audio import pad_audio, norm_audio

PAD_SECONDS = 0.9
TOO_LONG_SECONDS = 30.0


def _prepare(audio):
    audio = pad_audio(norm_audio(audio), PAD_SECONDS)
    duration = audio.waveform.shape[0] / audio.samplerate
    if duration > TOO_LONG_SECONDS:      # the reference's warning (transcribe.py:27-34): upstream's memory grows with T^2
        warnings.warn(
            f"Passing a long audio input ({duration:.1f}s) is not recommended, "
            "because K2 will require a large amount of memory. "
            "Read the upstream discussion for more details: "
            "https://github.com/k2-fsa/icefall/issues/1680"
        )
    return audio


def _result(stream):
    subwords = [Subword(token=t, seconds=s) for t, s in zip(stream.result.tokens, stream.result.timestamps)]
    return TranscribeResult(stream.result.text, subwords)


def transcribe(model, audio, config=None):
    """Inference audio data using the K2 model (transcribe.py:10-45).

    Args:
        model (K2Model): what `load_model()` returned
        audio (AudioData): Audio data to transcribe
        config (TranscribeConfig): Additional settings

    Returns:
        TranscribeResult
    """
    if config is None:
        config = TranscribeConfig()
    audio = _prepare(audio)
    stream = model.create_stream()
    stream.accept_waveform(audio.samplerate, audio.waveform)
    model.decode_stream(stream)
    return _result(stream)


def transcribe_batch(model, audios, config=None):
    """Additive: many utterances as one (or several pipelined) batches on the device; per utterance the same result as
    `transcribe` (every kernel masks by the utterance's own length)."""
    streams = []
    for a in audios:
        a = _prepare(a)
        st = model.create_stream()
        st.accept_waveform(a.samplerate, a.waveform)
        streams.append(st)
    model.decode_streams(streams)
    return [_result(st) for st in streams]
