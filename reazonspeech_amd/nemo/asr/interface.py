"""Result / config value types of the `reazonspeech.nemo.asr` API.

Field names, order and defaults follow pkg/nemo-asr/src/interface.py:4-36 so that
callers (CLI writers, the evaluation harness, notebooks) can switch packages without
touching their code.  `Hypothesis` is new: it is the adapter object that stands in for
NeMo's `Hypothesis` at the two attributes the reference reads (`y_sequence`,
`timestamp`; pkg/nemo-asr/src/decode.py:40,44).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class AudioData:
    """A waveform plus its sample rate (interface.py:4-8)."""
    waveform: np.ndarray
    samplerate: int


@dataclass
class Subword:
    """One emitted subword and the time (seconds, single point) it was emitted at
    (interface.py:10-17)."""
    seconds: float
    token_id: int
    token: str


@dataclass
class Segment:
    """A run of subwords with start/end seconds (interface.py:19-24)."""
    start_seconds: float
    end_seconds: float
    text: str


@dataclass
class TranscribeResult:
    """What `transcribe()` returns (interface.py:26-31).  `hypothesis` is only filled
    when `TranscribeConfig.raw_hypothesis` is set (transcribe.py:57-58)."""
    text: str
    subwords: List[Subword]
    segments: List[Segment]
    hypothesis: object = None


@dataclass
class TranscribeConfig:
    """Per-call options (interface.py:33-36)."""
    verbose: bool = True
    raw_hypothesis: bool = False


#: What the beam search's `timestamp[idx]` holds relative to the alignment step i = frame + labels emitted before
#: (the index the device kernel records): `timestamp[idx] = i + ALSD_TIMESTAMP_OFFSET`.  With 1 the reference's
#: conversion `frame = step - idx - 1` (decode.py:48) returns the emission frame itself, i.e. subword time =
#: 0.08 * frame - 0.5 s — the semantics this package documents.  NeMo's own ALSD bookkeeping is [UPSTREAM] and could
#: not be checked here (a +-1 frame = 80 ms ambiguity in subword / segment times; text and ids
#: are unaffected): if a real NeMo hypothesis shows `timestamp[idx] = i`, set this to 0 — the one place to flip.
ALSD_TIMESTAMP_OFFSET = 1


class _IdSequence(list):
    """A list of ints that also answers `.tolist()` like the tensor NeMo returns
    (decode.py:40 calls `hyp.y_sequence.tolist()`)."""

    def tolist(self):
        return list(self)


def _int_list(values):
    """python ints from an integer ndarray / tensor (one C call) or any iterable"""
    out = values.tolist() if hasattr(values, "tolist") else list(values)
    return out if all(type(v) is int for v in out) else [int(v) for v in out]


@dataclass
class Hypothesis:
    """Greedy-decode result shaped like NeMo's ALSD `Hypothesis`.

    The reference post-processor was written for ALSD beam search output: it drops the
    first element of `y_sequence` (a prepended blank, decode.py:38-40) and converts
    `timestamp[idx]` (an alignment *step* = frames + symbols so far) back to a frame
    with `step - idx - 1` (decode.py:48).  The greedy kernel reports plain encoder frame
    indices, so the adapter stores `[blank] + ids` and `frame + idx + 1`; the
    reference formula then yields `seconds = max(0.08*frame - 0.5, 0)` unchanged.
    """
    y_sequence: _IdSequence
    timestamp: List[int]
    frames: List[int] = field(default_factory=list)
    score: Optional[float] = None

    @classmethod
    def from_greedy(cls, ids, frames, blank_id):
        ids, frames = _int_list(ids), _int_list(frames)
        steps = [f + idx + 1 for idx, f in enumerate(frames)]
        return cls(_IdSequence([int(blank_id)] + ids), steps, frames)

    @classmethod
    def from_alsd(cls, ids, steps, blank_id, offset=None):
        """Beam-search result: `steps[idx]` is the alignment index i = frame + labels before (what rs_rnnt_alsd
        records); the timestamps handed to the reference post-processor are `i + ALSD_TIMESTAMP_OFFSET`."""
        off = ALSD_TIMESTAMP_OFFSET if offset is None else int(offset)
        ids, steps = _int_list(ids), _int_list(steps)
        frames = [s - idx for idx, s in enumerate(steps)]
        return cls(_IdSequence([int(blank_id)] + ids), [s + off for s in steps], frames)
