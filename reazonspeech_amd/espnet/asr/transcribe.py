"""`load_model()` / `transcribe()` of `reazonspeech.espnet.asr` (pkg/espnet-asr/src/transcribe.py:12-82).

The control flow is the reference's: audio longer than WINDOW_SECONDS is cut at the midpoint of the longest non-speech
stretch the CTC head finds in the next 20 s (:59-67), each piece is recognised with (16000, 8000) samples of zero padding
(:69) and split into time-stamped segments by CTC segmentation (:72-77).  Underneath, `Speech2Text` is replaced by
`EspnetModel` (model.py): HIP front-end, Conv2dSubsampling, conformer blocks, CTC head and transducer greedy search."""
import sys

import numpy as np
import torch

from .audio import norm_audio
from .interface import TranscribeConfig, TranscribeResult, Segment
from .ctc import split_text, find_blank

# Hyper parameters (transcribe.py:9-10)
WINDOW_SECONDS = 20
PADDING = (16000, 8000)


CHECKPOINT_ENV = "REAZONSPEECH_ESPNET_CHECKPOINT"


def load_model(device=None, checkpoint=None, config=None, seed=0, beam_size=None, max_pops=0, precision="bf16", synthetic=False):
    """Load the ReazonSpeech ESPnet model onto a ROCm GPU (transcribe.py:12-32).

    Args:
      device (str): "cuda" / "cuda:N"; None picks "cuda" when available like the reference (:20-24).  There is no CPU path
        in this package: "cpu" raises.
      checkpoint (str): an ESPnet2 model directory or model-zoo `.zip` (training `config.yaml`, `*.pth`, `feats_stats.npz`);
        defaults to $REAZONSPEECH_ESPNET_CHECKPOINT.  Read without ESPnet (runtime/weights_espnet.py: read_espnet), strictly.
      config (ModelConfig): architecture (family "espnet") for synthetic weights; default: the 120M Conformer-Transducer shape.
      seed (int): seed of the synthetic weights.
      beam_size (int): transducer search width, as Speech2Text's `beam_size`: <= 1 greedy search, larger the default beam
        search.  None = 20 for a checkpoint (Speech2Text's default, which the reference keeps: :27-31) and 1 for synthetic
        weights (an untrained joint can make the default search extend one frame without end; see `max_pops`).
      max_pops (int): prediction-network evaluations the beam search may spend per frame (0 = 16 * beam_size); upstream has
        no bound.  A window that exceeds it is retried with four and sixteen times the bound, then decoded greedily with a
        warning (model.py: EspnetModel._search) — never truncated, and the windows already decoded are kept.
      precision (str): "bf16" = the throughput mode; "fp32" = the float32 parity mode (float32 weights, activations and
        arithmetic end to end — what ESPnet computes on the reference's path; include/rs_asr.h "precision_f32").

    The reference downloads `reazon-research/reazonspeech-espnet-v2` through espnet_model_zoo (:27-31), which an offline box
    cannot do: give `checkpoint=` / the environment variable.  Without a checkpoint this RAISES; seeded synthetic weights of
    the architecture (timings valid, transcripts meaningless) are loaded only on request — `config=`, `synthetic=True` or
    $REAZONSPEECH_AMD_SYNTHETIC=1 — and a warning says so."""
    from ...runtime.config import ESPNET_CONFORMER_120M
    from ...runtime.weights_espnet import synthetic_state_dict_espnet
    from .model import EspnetModel, synthetic_token_list
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if str(device).startswith("cpu"):
        raise RuntimeError("reazonspeech_amd runs on MI355X (gfx950) only; no CPU path exists (use the reference package for CPU inference)")
    import os
    from ...runtime.weights_espnet import read_espnet
    checkpoint = checkpoint or (os.environ.get(CHECKPOINT_ENV) if config is None else None)
    if checkpoint:
        if not os.path.exists(checkpoint):
            raise FileNotFoundError(f"checkpoint {checkpoint!r} does not exist")
        cfg, sd, tokens = read_espnet(checkpoint)
        return EspnetModel(cfg, sd, tokens, device=device, beam_size=20 if beam_size is None else beam_size, max_pops=max_pops,
                           precision=precision)
    cfg = config or ESPNET_CONFORMER_120M
    if config is None:
        if not (synthetic or os.environ.get("REAZONSPEECH_AMD_SYNTHETIC", "0") not in ("", "0")):
            raise FileNotFoundError(f"no ESPnet2 checkpoint: give `checkpoint=` or ${CHECKPOINT_ENV} (a model directory / model-zoo .zip of "
                                    "reazon-research/reazonspeech-espnet-v2; the reference downloads it through espnet_model_zoo, :27-31).  Seeded "
                                    "synthetic weights are loaded only on request: config=..., synthetic=True or $REAZONSPEECH_AMD_SYNTHETIC=1.")
        print("[reazonspeech_amd] WARNING: SEEDED SYNTHETIC weights of the 120M Conformer-Transducer architecture were requested "
              "(`synthetic=True` / $REAZONSPEECH_AMD_SYNTHETIC): timings are valid, transcripts are meaningless.", file=sys.stderr, flush=True)
    return EspnetModel(cfg, synthetic_state_dict_espnet(cfg, seed), synthetic_token_list(cfg.vocab_size, seed), device=device,
                       beam_size=1 if beam_size is None else beam_size, max_pops=max_pops, precision=precision)


def _windows(model, waveform, window):
    """Cut `waveform` into the pieces the recogniser sees (transcribe.py:59-67,78): everything that is left when it fits one
    window, otherwise the head of the next `window` samples up to the middle of its longest silent stretch (`find_blank`).
    Lazy: the blank finder of a piece runs after the previous piece has been recognised, like the reference's loop.
    -> (offset, samples)"""
    offset = 0
    while offset < len(waveform):
        rest = waveform[offset:]
        if len(rest) > window:
            gap = find_blank(model, rest[:window])
            rest = rest[:int((gap.start + gap.end) / 2)]
        yield offset, rest
        offset += len(rest)


def transcribe(model, audio, config=None):
    """Interface function to transcribe audio data (transcribe.py:34-82).

    Args:
      model (EspnetModel): what `load_model()` returned
      audio (AudioData): Audio to transcribe
      config (TranscribeConfig): Additional settings

    Returns:
      TranscribeResult
    """
    config = config or TranscribeConfig()
    audio = norm_audio(audio)
    rate = audio.samplerate
    total = len(audio.waveform)
    texts, segments = [], []
    for offset, samples in _windows(model, audio.waveform, int(WINDOW_SECONDS * rate)):
        text = model(np.pad(samples, PADDING, mode="constant"))[0][0]        # nbest[0] = (text, tokens, ids, hypothesis)
        texts.append(text)
        segments.extend(Segment(start_seconds=(offset + first) / rate, end_seconds=(offset + last) / rate, text=piece)
                        for first, last, piece in split_text(model, samples, text))
        if config.verbose:       # the reference draws a tqdm bar over the samples (transcribe.py:55-56,79-80)
            done = offset + len(samples)
            print(f"\rTranscribe: {done}/{total}", end="" if done < total else "\n", file=sys.stderr, flush=True)
    return TranscribeResult("".join(texts), segments)


def transcribe_batch(model, audios, config=None):
    """Additive: many SHORT utterances (each at most one 20 s window) recognised as one batch on the device, then segmented
    one by one on the host.  An utterance longer than a window goes through `transcribe` on its own."""
    if config is None:
        config = TranscribeConfig(verbose=False)
    norm = [norm_audio(a) for a in audios]
    window = int(WINDOW_SECONDS * 16000)
    short = [i for i, a in enumerate(norm) if len(a.waveform) <= window]
    out = [None] * len(norm)
    texts = model.recognize_batch([norm[i].waveform for i in short]) if short else []
    for i, asr in zip(short, texts):
        samples = norm[i].waveform
        segs = [Segment(start / 16000, end / 16000, text) for start, end, text in split_text(model, samples, asr)]
        out[i] = TranscribeResult(asr, segs)
    for i, a in enumerate(norm):
        if out[i] is None:
            out[i] = transcribe(model, a, TranscribeConfig(verbose=False))
    return out
