"""A deterministic stand-in for sherpa_onnx.OfflineRecognizer, shared by tests/golden/make_reference_k2_golden.py (which drives
the REFERENCE's own pkg/k2-asr/src/transcribe.py with it) and tests/test_k2_host.py (which drives this repo's package with it):
both sides see the same tokens / timestamps for the same samples, so whatever differs is host logic."""
import numpy as np

TOKENS = ["<blk>", "<sos/eos>", "<unk>", "。", "、"] + [chr(c) for c in range(0x3041, 0x3041 + 40)] + [" the", " cat"]


class _Result:
    def __init__(self, tokens, timestamps):
        self.tokens, self.timestamps = tokens, timestamps
        self.text = "".join(tokens)


class _Stream:
    def __init__(self):
        self.sample_rate, self.samples, self.result = None, None, None

    def accept_waveform(self, sample_rate, waveform):
        self.sample_rate, self.samples = sample_rate, np.asarray(waveform)


class FakeRecognizer:
    def __init__(self):
        self.seen = []

    def create_stream(self):
        return _Stream()

    def decode_stream(self, stream):
        n = len(stream.samples)
        self.seen.append((int(stream.sample_rate), n, float(np.abs(stream.samples[:14400]).max()) if n else 0.0))
        rng = np.random.default_rng(n)
        frames = np.sort(rng.choice(max(n // 640, 1), size=min(12, max(n // 640, 1)), replace=False))
        toks = [TOKENS[int(rng.integers(3, len(TOKENS)))] for _ in frames]
        stream.result = _Result(toks, [float(np.float32(0.04 * f)) for f in frames])


def audio(seconds, seed=0):
    rng = np.random.default_rng(seed)
    return (0.1 * rng.standard_normal(int(seconds * 16000))).astype(np.float32)
