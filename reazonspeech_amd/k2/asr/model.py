"""The object `reazonspeech.k2.asr.load_model()` returns here: icefall's Zipformer2 transducer on one MI355X.

It stands where `sherpa_onnx.OfflineRecognizer` stands in the reference (pkg/k2-asr/src/huggingface.py:73-83) and answers the
calls the reference makes on it (transcribe.py:36-45):
    stream = model.create_stream(); stream.accept_waveform(samplerate, waveform); model.decode_stream(stream)
    stream.result.tokens / .timestamps / .text
plus the batched form `decode_streams` (sherpa-onnx has it too).  [UPSTREAM] conventions of sherpa-onnx's result conversion
(offline-recognizer-transducer-impl.h Convert, symbol-table.cc): a token's text is its tokens.txt symbol with a leading U+2581 replaced by
a space and `<0xNN>` byte tokens joined into UTF-8; text = the tokens concatenated; timestamp = frame index x 0.04 s."""
import re

import numpy as np

from ...runtime.model import AsrModel

_BYTE = re.compile(r"^<0x([0-9A-Fa-f]{2})>$")


class _Result:
    def __init__(self, tokens, timestamps, text):
        self.tokens, self.timestamps, self.text = tokens, timestamps, text


class _Stream:
    """sherpa_onnx.OfflineStream: holds one utterance's samples, then its result"""

    def __init__(self):
        self.samples = np.zeros((0,), np.float32)
        self.sample_rate = 16000
        self.result = _Result([], [], "")

    def accept_waveform(self, sample_rate, waveform):
        self.sample_rate = int(sample_rate)
        self.samples = np.concatenate([self.samples, np.asarray(waveform, dtype=np.float32).reshape(-1)])


def read_tokens(path):
    """tokens.txt: one `symbol id` pair per line ([UPSTREAM] sherpa-onnx SymbolTable) -> list indexed by id"""
    table = {}
    with open(path, encoding="utf-8") as fp:
        for line in fp:
            line = line.rstrip("\n")
            if not line.strip():
                continue
            sym, _, idx = line.rpartition(" ")
            table[int(idx)] = sym if sym else " "
    return [table.get(i, "<unk>") for i in range(max(table) + 1)]


def synthetic_tokens(vocab_size, seed=0):
    """an icefall-style tokens.txt for synthetic-weight runs: <blk> 0, <sos/eos> 1, <unk> 2, then punctuation and characters"""
    fixed = ["<blk>", "<sos/eos>", "<unk>", "。", "、", "?", "!", "▁"]
    pool = [chr(c) for c in range(0x3041, 0x3097)] + [chr(c) for c in range(0x30A1, 0x30FB)] + [chr(c) for c in range(0x4E00, 0x4E00 + 16384)]
    rng = np.random.default_rng(seed)
    rng.shuffle(pool)
    toks = (fixed + pool)[:vocab_size]
    assert len(toks) == vocab_size and len(set(toks)) == vocab_size
    return toks


class K2Model:
    def __init__(self, cfg, state_dict, tokens, device="cuda", pad_seconds=0.0, precision="bf16"):
        """precision: "bf16" = the throughput mode; "fp32" = float32 weights, activations and arithmetic end to end (what
        onnxruntime computes from the reference's default float32 graphs: pkg/k2-asr/src/huggingface.py:16,40-45)"""
        assert cfg.family == "k2" and len(tokens) == cfg.vocab_size
        self.cfg = cfg
        self.tokens = list(tokens)
        # the reference pads with np.pad before handing the samples over (transcribe.py:24); a stream's samples arrive padded
        self.am = AsrModel(cfg, state_dict, None, device=device, pad_seconds=pad_seconds, precision=precision)
        self.device = self.am.device

    # ---- sherpa-onnx's surface ------------------------------------------------------------------------------------------
    def create_stream(self):
        return _Stream()

    def decode_stream(self, stream):
        self.decode_streams([stream])

    def decode_streams(self, streams):
        for st in streams:
            if st.sample_rate != self.cfg.sample_rate:
                raise ValueError(f"sample rate {st.sample_rate}: the model expects {self.cfg.sample_rate} Hz (sherpa-onnx resamples; resample with norm_audio first)")
        res = self.am.transcribe_waveforms([st.samples for st in streams])
        for st, ids, frames in zip(streams, res.ids, res.frames):
            st.result = self.convert(ids, frames)

    # ---- result conversion ------------------------------------------------------------------------------------------------
    def symbol(self, i):
        """[UPSTREAM] sherpa-onnx SymbolTable: only a LEADING U+2581 (the SentencePiece word boundary) becomes a space"""
        s = self.tokens[i]
        return " " + s[1:] if s.startswith("▁") else s

    def convert(self, ids, frames):
        syms = [self.symbol(i) for i in ids]
        # byte-fallback pieces (<0xE3> ...) join into UTF-8 text; the token list keeps them as they are
        out, pending = [], bytearray()
        for s in syms:
            m = _BYTE.match(s)
            if m:
                pending.append(int(m.group(1), 16))
                continue
            if pending:
                out.append(pending.decode("utf-8", errors="replace"))
                pending = bytearray()
            out.append(s)
        if pending:
            out.append(pending.decode("utf-8", errors="replace"))
        step = self.cfg.seconds_per_frame()
        return _Result(syms, [float(np.float32(step * t)) for t in frames], "".join(out))
