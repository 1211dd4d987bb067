"""Data types of `reazonspeech.k2.asr` (pkg/k2-asr/src/interface.py:1-26): AudioData, Subword(seconds, token),
TranscribeResult(text, subwords), TranscribeConfig(verbose=True)."""
from dataclasses import dataclass

import numpy as np


@dataclass
class AudioData:
    """Container for audio waveform"""
    waveform: np.float32
    samplerate: int


@dataclass
class Subword:
    """A subword with a single-point timestamp"""
    seconds: float
    token: str


@dataclass
class TranscribeResult:
    text: str
    subwords: list


@dataclass
class TranscribeConfig:
    verbose: bool = True
