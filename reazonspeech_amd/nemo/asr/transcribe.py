"""`load_model()` / `transcribe()` — the drop-in pair (pkg/nemo-asr/src/transcribe.py:9-60),
plus the additive batched entry point `transcribe_batch()` the RTFx metric is quoted on.

What changes underneath: NeMo's `EncDecRNNTBPEModel` is replaced by
`reazonspeech_amd.runtime.model.AsrModel` (HIP kernels behind include/rs_asr.h); padding is
folded into the front-end kernel instead of `np.pad`; decoding is the checkpoint's strategy — batched
greedy or ALSD beam search on the device — adapted to the ALSD-shaped `Hypothesis` the reference
post-processor expects (interface.Hypothesis).
"""
import os
import sys

import torch

from .interface import TranscribeConfig, Hypothesis
from .decode import decode_hypothesis, PAD_SECONDS
from .audio import norm_audio

#: where a real checkpoint is looked for; the reference downloads
#: 'reazon-research/reazonspeech-nemo-v2' from the HF hub (transcribe.py:26-28), which an
#: offline box cannot do.
CHECKPOINT_ENV = "REAZONSPEECH_NEMO_CHECKPOINT"
HF_REPO = "reazon-research/reazonspeech-nemo-v2"      # transcribe.py:27


def resolve_checkpoint(checkpoint=None, allow_download=True):
    """Where the model's `.nemo` archive comes from, in the order a user of the reference would expect:
      1. the `checkpoint` argument, 2. $REAZONSPEECH_NEMO_CHECKPOINT,
      3. the Hugging Face cache of 'reazon-research/reazonspeech-nemo-v2' (what the reference's
         `from_pretrained` fills, transcribe.py:26-28), 4. a download of that repository when the hub is reachable
         (not with HF_HUB_OFFLINE=1).
    -> path of the archive, or None when there is none (the caller then falls back to synthetic weights, loudly)."""
    import glob
    path = checkpoint or os.environ.get(CHECKPOINT_ENV)
    if path:
        if not os.path.exists(path):
            raise FileNotFoundError(f"checkpoint {path!r} does not exist")
        return path
    try:
        from huggingface_hub import snapshot_download
    except ImportError:
        return None
    for local_only in (True, False):
        if not local_only and (not allow_download or os.environ.get("HF_HUB_OFFLINE", "0") not in ("", "0")):
            break
        try:
            root = snapshot_download(HF_REPO, local_files_only=local_only, allow_patterns=["*.nemo"], etag_timeout=5)
        except Exception as e:             # not cached / no network / authentication / disk: remembered for the caller's warning
            resolve_checkpoint.last_error = f"{'cache lookup' if local_only else 'download'}: {type(e).__name__}: {e}"
            continue
        found = sorted(glob.glob(os.path.join(root, "**", "*.nemo"), recursive=True))
        if found:
            return found[0]
    return None


SYNTHETIC_ENV = "REAZONSPEECH_AMD_SYNTHETIC"


def load_model(device=None, checkpoint=None, config=None, seed=0, pos_cap=None, decoding=None, beam_size=None,
               precision="bf16", synthetic=False):
    """Load the ReazonSpeech FastConformer-RNNT model onto a ROCm GPU.

    Args:
      device (str): "cuda" / "cuda:N" (ROCm devices report as cuda).  None picks "cuda" when
        available, like the reference (transcribe.py:18-22); there is no CPU execution path
        in this package, so "cpu" raises.
      checkpoint (str): path of a `.nemo` archive.  Defaults to $REAZONSPEECH_NEMO_CHECKPOINT, then to the Hugging Face
        cache / hub copy of 'reazon-research/reazonspeech-nemo-v2' (`resolve_checkpoint`).  When none can be found or
        downloaded this RAISES, as `from_pretrained` does in the reference (transcribe.py:26-28) — a transient hub error
        must not produce a model with made-up weights.
      config (ModelConfig), seed (int), synthetic (bool): SEEDED SYNTHETIC weights are loaded only on request — `config=`
        (an architecture; no checkpoint lookup is made), `synthetic=True` or $REAZONSPEECH_AMD_SYNTHETIC=1 (benchmarks /
        tests: timings are valid, transcripts meaningless), and a warning says so.
      decoding (str): override the checkpoint's decoding strategy: "greedy_batch", "alsd" (alignment-length
        synchronous beam search, what the reference checkpoint ships with: decode.py:29,38-41) or "beam" ([UPSTREAM] NeMo's
        `strategy: beam`, the default Graves beam search — the one ESPnet runs, csrc/k_rnnt_beam.hip).
      beam_size (int): override the beam size ("alsd": 1..8, "beam": 1..64).
      precision (str): "bf16" (default): the throughput mode — bf16 matrix-core operands, float32 accumulation, float32
        residual stream, exact float32 decode.  "fp32": the parity mode — float32 weights, activations and arithmetic end
        to end, i.e. what the reference computes (transcribe.py:26-28, :48-53 run NeMo in float32 without autocast; the
        same keyword as `reazonspeech.k2.asr.load_model(precision=...)`, pkg/k2-asr/src/huggingface.py:16); ~20x slower.
        "fp32x3": the float32 mode with every float32 PRODUCT of its GEMMs formed from three bf16 matrix-core terms (hi / lo split,
        float32 accumulation): twice the float32 mode's speed; not an IEEE chain, but every id of the 256-row float32-oracle goldens
        is reproduced (tests/test_gpu_fullsize.py).
      pos_cap (int): encoder frames (80 ms each) the resident relative-position tables cover at load time
        (default 1024, about 82 s); longer utterances grow the tables on first use.

    Returns:
      reazonspeech_amd.runtime.model.AsrModel
    """
    from ...runtime.config import FASTCONFORMER_619M
    from ...runtime.model import AsrModel
    from ...runtime.tokenizer import SentencePieceTokenizer, SyntheticTokenizer
    from ...runtime import weights as W

    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    if str(device).startswith("cpu"):
        raise RuntimeError("reazonspeech_amd runs on MI355X (gfx950) only; no CPU path exists "
                           "(use the reference package for CPU inference)")
    if config is not None and checkpoint is None:
        if os.environ.get(CHECKPOINT_ENV):
            print(f"[reazonspeech_amd] WARNING: ${CHECKPOINT_ENV} is set but ignored because an explicit `config` was passed "
                  f"(synthetic weights of that architecture are generated); pass `checkpoint=` to load the archive.", file=sys.stderr, flush=True)
    resolve_checkpoint.last_error = None
    want_synthetic = config is not None or synthetic or os.environ.get(SYNTHETIC_ENV, "0") not in ("", "0")
    checkpoint = None if want_synthetic and checkpoint is None else resolve_checkpoint(checkpoint)
    if not checkpoint and not want_synthetic:
        raise FileNotFoundError(
            f"no checkpoint: neither the `checkpoint` argument, ${CHECKPOINT_ENV}, nor a cached / downloadable copy of '{HF_REPO}' was found"
            + (f" (last lookup failure — {resolve_checkpoint.last_error})" if getattr(resolve_checkpoint, "last_error", None) else "")
            + f".  Seeded synthetic weights are loaded only on request: config=..., synthetic=True or ${SYNTHETIC_ENV}=1.")
    if checkpoint:
        cfg, sd, tok_bytes = W.read_nemo(checkpoint)
        tokenizer = SentencePieceTokenizer(tok_bytes) if tok_bytes else SyntheticTokenizer(cfg.vocab_size)
    else:
        if config is None:
            print("[reazonspeech_amd] WARNING: SEEDED SYNTHETIC weights of the 619M architecture were requested "
                  f"(`synthetic=True` / ${SYNTHETIC_ENV}): timings are valid, transcripts are meaningless.", file=sys.stderr, flush=True)
        cfg = config or FASTCONFORMER_619M
        sd = W.synthetic_state_dict(cfg, seed)
        tokenizer = SyntheticTokenizer(cfg.vocab_size, seed)
    if decoding is not None:
        cfg = cfg.with_(decoding=str(decoding))
    if beam_size is not None:
        cfg = cfg.with_(beam_size=int(beam_size))
    cfg.validate()
    kw = {} if pos_cap is None else {"pos_cap": int(pos_cap)}
    return AsrModel(cfg, sd, tokenizer, device=device, pad_seconds=PAD_SECONDS, precision=precision, **kw)


def _prepare(audio):
    """16 kHz mono float32 waveform (transcribe.py:44 minus the padding, which the kernel applies)"""
    import numpy as np
    return np.ascontiguousarray(norm_audio(audio).waveform, dtype=np.float32)


def transcribe_batch(model, audios, config=None, distributed=False):
    """Transcribe a list of AudioData in one batched pass.

    Each utterance is processed exactly as `transcribe()` would process it alone
    (per-utterance padding, masking and normalisation inside the kernels).

    With `distributed=True` and `torch.distributed` initialised (one process per GPU; the reference's
    multi-GPU mechanism, pkg/evaluation/src/base.py:194-212) every rank passes the same list, decodes a
    length-balanced shard on its own GPU and gets every result back, in the caller's order, through the
    path's one collective (an RCCL all_gather of the hypotheses).

    Returns:
      list[TranscribeResult]
    """
    if config is None:
        config = TranscribeConfig()
    waves = [_prepare(a) for a in audios]
    if config.verbose:
        # the reference forwards `verbose` to NeMo (transcribe.py:52), which draws a tqdm bar on stderr
        print(f"[reazonspeech_amd] transcribing {len(waves)} utterance(s), "
              f"{sum(len(w) for w in waves) / 16000.0:.1f} s of audio on {model.device}", file=sys.stderr, flush=True)
    results = [None] * len(waves)

    def to_results(indices, decoded):
        # ids -> text / subwords / segments (decode.py:28-66).  Called per batch while the GPU works on the next ones.
        for k, i in enumerate(indices):
            ids, frames = decoded.ids[k], decoded.frames[k]
            if decoded.scores is not None:
                # beam search: alignment steps (frame + labels before) through the adapter with the documented offset
                hyp = Hypothesis.from_alsd(ids, [f + idx for idx, f in enumerate(frames)], model.cfg.blank_id)
                hyp.score = decoded.scores[k]
            else:
                hyp = Hypothesis.from_greedy(ids, frames, model.cfg.blank_id)
            ret = decode_hypothesis(model, hyp)
            if config.raw_hypothesis:
                ret.hypothesis = hyp
            results[i] = ret

    if distributed:
        to_results(list(range(len(waves))), model.transcribe_waveforms_sharded(waves))
    else:
        model.transcribe_waveforms(waves, on_batch=to_results)
    return results


def transcribe(model, audio, config=None):
    """Inference of one utterance (transcribe.py:30-60).

    Args:
        model: what `load_model()` returned
        audio (AudioData): audio to transcribe
        config (TranscribeConfig): additional settings

    Returns:
        TranscribeResult
    """
    return transcribe_batch(model, [audio], config)[0]
