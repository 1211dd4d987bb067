"""`load_model()` of `reazonspeech.k2.asr` (pkg/k2-asr/src/huggingface.py:16-83).

The reference resolves (language, precision) to four files of a Hugging Face repository — tokens.txt and the encoder / decoder /
joiner ONNX graphs (:26-66) — looks them up in the local cache before going online (:68-71) and hands them to sherpa-onnx
(:73-83: one thread, 16 kHz, 80-dim features, greedy search).  Here the same files are read WITHOUT onnx / onnxruntime /
sherpa-onnx (runtime/onnx_lite.py) and run by the HIP kernels of csrc/k_zipformer.hip; argument checking and error messages
follow the reference."""
import os
import sys

import torch

REPOS = {                 # language -> (repository, epochs in the file names)      huggingface.py:26-35
    "ja": ("reazon-research/reazonspeech-k2-v2", 99),
    "ja-en": ("reazon-research/reazonspeech-k2-v2-ja-en", 35),
    "ja-en-mls-5k": ("reazon-research/reazonspeech-k2-v2-ja-en-mls-5k-corrected", 21),
}
CHECKPOINT_ENV = "REAZONSPEECH_K2_CHECKPOINT"


def repo_files(language, precision):
    """-> (repository id, {tokens, encoder, decoder, joiner} file names); ValueError like the reference (:37-38, :61-62)"""
    if language not in REPOS:
        raise ValueError(f"Unknown language: '{language}'")
    repo, epochs = REPOS[language]
    stem = {part: f"{part}-epoch-{epochs}-avg-1" for part in ("encoder", "decoder", "joiner")}
    files = {
        "fp32": {"tokens": "tokens.txt", **{p: s + ".onnx" for p, s in stem.items()}},
        "int8": {"tokens": "tokens.txt", **{p: s + ".int8.onnx" for p, s in stem.items()}},
        "int8-fp32": {"tokens": "tokens.txt", "encoder": stem["encoder"] + ".int8.onnx", "decoder": stem["decoder"] + ".onnx",
                      "joiner": stem["joiner"] + ".int8.onnx"},
    }
    if precision not in files:
        raise ValueError("Unknown precision: '%s'" % precision)
    return repo, files[precision]


SYNTHETIC_ENV = "REAZONSPEECH_AMD_SYNTHETIC"


def resolve_checkpoint(language, precision, checkpoint=None):
    """directory holding the four files: the argument, $REAZONSPEECH_K2_CHECKPOINT, the Hugging Face cache
    (`snapshot_download(local_files_only=True)`, :68-69), then the hub (:70-71).  Like the reference, a failed lookup RAISES
    (a missing `huggingface_hub`, no cache entry while offline, a network / authentication error, a partial download): a
    caller with a transient hub error must not get a model with made-up weights."""
    repo, files = repo_files(language, precision)
    for cand in (checkpoint, os.environ.get(CHECKPOINT_ENV)):
        if cand:
            if not os.path.isdir(cand):
                raise FileNotFoundError(f"checkpoint directory {cand!r} does not exist")
            return cand, files
    import huggingface_hub as hf
    try:
        return hf.snapshot_download(repo, local_files_only=True), files
    except hf.utils.LocalEntryNotFoundError:
        return hf.snapshot_download(repo), files


def load_model(device=None, precision="fp32", language="ja", checkpoint=None, config=None, seed=0, compute="bf16", synthetic=False):
    """Load the ReazonSpeech k2 model onto a ROCm GPU (huggingface.py:16-83).

    Args:
      device (str): "cuda" / "cuda:N"; None picks "cuda".  The reference's default is "cpu" (sherpa-onnx's provider); this
        package has no CPU path — "cpu" and "coreml" raise.
      precision (str): "fp32", "int8" or "int8-fp32": which ONNX files are read, validated like the reference (:61-62).  The HIP
        path computes bf16 x bf16 -> f32 from the float32 weights; quantized graphs are refused when they are actually read.
      language (str): "ja", "ja-en" or "ja-en-mls-5k" (:26-38)
      checkpoint (str): directory with tokens.txt and the three ONNX files (default: $REAZONSPEECH_K2_CHECKPOINT, then the
        Hugging Face cache, then the hub)
      config (ZipformerConfig), seed, synthetic: SEEDED SYNTHETIC weights are loaded only on request — `config=` (an architecture),
        `synthetic=True` or $REAZONSPEECH_AMD_SYNTHETIC=1 (benchmarks, tests: timings are valid, transcripts meaningless).  Without
        such a request a checkpoint that cannot be found or downloaded raises, as `hf.snapshot_download` does in the reference.
      compute (str): "bf16" (default): the throughput mode — bf16 matrix-core operands, float32 accumulation and residual stream,
        exact float32 decode.  "fp32": the parity mode — float32 weights, activations and arithmetic end to end, i.e. what
        onnxruntime computes from the reference's default float32 graphs; greedy ids identical to the float32 oracle, ~6x slower.
        "fp32x3": the float32 mode with its products formed from three bf16 matrix-core terms (2x faster; same 256-row golden, ids identical).
        (`precision` keeps the reference's meaning: WHICH files are read.)

    A real icefall export has never been read by runtime/k2_onnx.py (no file is reachable from the build environment): the reader
    is verified against files written in the documented export layout only, and checks itself after loading (every expected
    tensor found exactly once, parameter count == cfg.n_params(), folded softmax constants summing to 1) — treat a first real
    checkpoint as UNVERIFIED until one transcript has been compared with sherpa-onnx.

    Returns:
      K2Model (answers sherpa_onnx.OfflineRecognizer's create_stream / decode_stream)
    """
    from ...runtime.k2_config import ZIPFORMER_159M
    from ...runtime.k2_weights import synthetic_state_dict_k2
    from .model import K2Model, read_tokens, synthetic_tokens
    repo_files(language, precision)                       # argument errors first, like the reference
    if device is None:
        device = "cuda"
    if not str(device).startswith("cuda"):
        raise RuntimeError(f"device {device!r}: reazonspeech_amd runs on MI355X (gfx950) only; no CPU / CoreML path exists "
                           "(use the reference package for those)")
    if not torch.cuda.is_available():
        raise RuntimeError("reazonspeech_amd needs a ROCm GPU: torch.cuda.is_available() is False")
    if compute not in ("bf16", "fp32", "fp32x3"):
        raise ValueError(f"compute must be 'bf16', 'fp32' or 'fp32x3', not {compute!r}")
    want_synthetic = config is not None or synthetic or os.environ.get(SYNTHETIC_ENV, "0") not in ("", "0")
    if want_synthetic and not checkpoint:
        basedir, files = None, None
    else:
        basedir, files = resolve_checkpoint(language, precision, checkpoint)
    if basedir:
        from ...runtime.k2_onnx import read_k2_onnx
        cfg, sd = read_k2_onnx(os.path.join(basedir, files["encoder"]), os.path.join(basedir, files["decoder"]), os.path.join(basedir, files["joiner"]))
        tokens = read_tokens(os.path.join(basedir, files["tokens"]))
        if len(tokens) != cfg.vocab_size:
            raise ValueError(f"tokens.txt has {len(tokens)} symbols, the joiner {cfg.vocab_size} outputs")
        cfg = cfg.with_(unk_id=tokens.index("<unk>") if "<unk>" in tokens else -1)
        return K2Model(cfg, sd, tokens, device=device, precision=compute)
    cfg = config or ZIPFORMER_159M
    print(f"[reazonspeech_amd] WARNING: SEEDED SYNTHETIC weights of the {cfg.n_params() / 1e6:.0f}M Zipformer architecture were requested "
          f"(`config=` / `synthetic=True` / ${SYNTHETIC_ENV}): timings are valid, transcripts are meaningless.", file=sys.stderr, flush=True)
    return K2Model(cfg, synthetic_state_dict_k2(cfg, seed), synthetic_tokens(cfg.vocab_size, seed), device=device, precision=compute)
