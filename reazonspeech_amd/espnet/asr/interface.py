"""Data types of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/interface.py:1-25): AudioData, Segment,
TranscribeResult(text, segments), TranscribeConfig(verbose=True)."""
from dataclasses import dataclass

import numpy as np


@dataclass
class AudioData:
    """A container for audio waveform"""
    waveform: np.float32
    samplerate: int


@dataclass
class Segment:
    """A segment of transcription with timestamps"""
    start_seconds: float
    end_seconds: float
    text: str


@dataclass
class TranscribeResult:
    text: str
    segments: list


@dataclass
class TranscribeConfig:
    verbose: bool = True
