"""Value types of the `reazonspeech.k2.asr` API.

Names, field order and defaults are the ones callers of the reference package construct and read
(pkg/k2-asr/src/interface.py:4-26), so that switching packages needs no change on their side.  Compared with the NeMo
package there are no segments and no token ids: the sherpa-onnx result the reference consumes has only token strings and one
time stamp per token (pkg/k2-asr/src/transcribe.py:41-45).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class AudioData:
    """Mono waveform (float32 samples in [-1, 1]) and its sample rate in Hz (interface.py:4-8)."""
    waveform: np.ndarray
    samplerate: int


@dataclass
class Subword:
    """One emitted token and the single point in time at which the transducer emitted it, in seconds of the PADDED audio the
    recogniser saw: like the reference, the 0.9 s of leading padding (transcribe.py:7,31-33) is NOT subtracted
    (interface.py:10-14; transcribe.py:43)."""
    seconds: float
    token: str


@dataclass
class TranscribeResult:
    """What `transcribe()` returns: the text (tokens joined) and the tokens with their times (interface.py:16-19)."""
    text: str
    subwords: List[Subword] = field(default_factory=list)


@dataclass
class TranscribeConfig:
    """Per-call options (interface.py:21-26).  `verbose` only controls the long-audio warning."""
    verbose: bool = True
