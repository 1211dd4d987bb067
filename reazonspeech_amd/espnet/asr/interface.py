"""Value types of the `reazonspeech.espnet.asr` API.

Names, field order and defaults follow pkg/espnet-asr/src/interface.py:4-25 so that the reference's callers (its CLI, the
subtitle writers, evaluation notebooks) keep working unchanged.  This family reports SEGMENTS only: the window loop of
`transcribe()` produces one `Segment` per 30 s window with the window's bounds as its times (pkg/espnet-asr/src/transcribe.py:54-74).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class AudioData:
    """Mono waveform (float32 samples) and its sample rate in Hz (interface.py:4-8)."""
    waveform: np.ndarray
    samplerate: int


@dataclass
class Segment:
    """Text decoded from one window of the input, with the window's start and end in seconds of the original audio
    (interface.py:10-15)."""
    start_seconds: float
    end_seconds: float
    text: str


@dataclass
class TranscribeResult:
    """What `transcribe()` returns: the windows' texts concatenated, and the windows (interface.py:17-20)."""
    text: str
    segments: List[Segment] = field(default_factory=list)


@dataclass
class TranscribeConfig:
    """Per-call options (interface.py:22-25).  `verbose` turns the progress output of the window loop on."""
    verbose: bool = True
