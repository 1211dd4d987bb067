"""A deterministic stand-in for the ESPnet model, shared by tests/golden/make_reference_espnet_golden.py (which drives the
REFERENCE's own transcribe.py / ctc.py with it) and tests/test_espnet_host.py (which drives this repo's restatement with it):
both sides see the same CTC posteriors and the same recognised text for the same samples, so whatever differs is host logic.

It answers both call forms: the reference's (`model.asr_model.encode` + `.ctc.softmax`, `model(padded)[0][0]`) and this
repo's direct one (`ctc_posteriors`)."""
import numpy as np
import torch

TOKENS = ["<blank>", "<unk>", "。", "、", "?", "!", ","] + [chr(c) for c in range(0x3041, 0x3041 + 60)] + ["<sos/eos>"]
FRAME = 512          # samples per encoder frame of the fake
PADDING = (16000, 8000)


def script(n_samples, salt=0):
    """-> (frames, [(frame, token id)]): what the fake 'hears' in n_samples samples.  Speech comes in bursts of ~2.5 s
    separated by ~1.3 s of silence; sentences end in punctuation now and then."""
    T = max(n_samples // FRAME, 1)
    rng = np.random.default_rng(n_samples * 7919 + salt)
    events = []
    t = int(rng.integers(2, 8))
    k = 0
    while t < T - 2:
        burst = int(rng.integers(50, 110))
        end = min(t + burst, T - 2)
        while t < end:
            k += 1
            tok = int(rng.integers(7, len(TOKENS) - 1))
            if k % 11 == 0:
                tok = 2 if rng.random() < 0.7 else 4
            elif k % 7 == 0:
                tok = 3
            events.append((t, tok))
            t += int(rng.integers(3, 9))
        t = end + int(rng.integers(25, 60))
    return T, events


def salt_of(samples):
    """content-dependent seed: two windows of equal length hear different things"""
    head = np.asarray(samples[:64], np.float64)
    return int(np.abs(head).sum() * 1e6) % 100003


def posteriors(samples):
    n = len(samples)
    T, events = script(n, salt_of(samples))
    V = len(TOKENS)
    lpz = np.full((T, V), 0.004 / (V - 1), np.float32)
    lpz[:, 0] = 0.996
    speech = np.zeros(T, bool)
    for t, tok in events:
        lpz[t, :] = 0.05 / (V - 1)
        lpz[t, tok] = 0.95
        speech[max(0, t - 2):t + 3] = True
    soft = speech & (lpz[:, 0] > 0.9)
    lpz[soft, 0] = 0.9                      # frames around a character: blank below find_blank's 0.98 threshold
    lpz[soft, 1] = 0.1 - 0.004
    return lpz


def text_of(samples):
    return "".join(TOKENS[tok] for _, tok in script(len(samples), salt_of(samples))[1])


class _Ctc:
    def softmax(self, enc):
        return enc._lpz


class _AsrModel:
    blank_id = 0
    token_list = TOKENS

    def __init__(self):
        self.ctc = _Ctc()

    def encode(self, speech, length):
        wav = speech.detach().cpu().numpy().reshape(-1)
        lpz = posteriors(wav)
        enc = torch.zeros((1, lpz.shape[0], 1))
        enc._lpz = torch.from_numpy(lpz)[None]
        return enc, torch.tensor([lpz.shape[0]])


class FakeEspnetModel:
    dtype = "float32"
    device = "cpu"

    def __init__(self):
        self.asr_model = _AsrModel()
        self.calls = []

    def __call__(self, speech):
        wav = np.asarray(speech)[PADDING[0]:len(speech) - PADDING[1]]
        self.calls.append(len(wav))
        t = text_of(wav)
        return [(t, list(t), [], None)]

    def ctc_posteriors(self, samples):
        return posteriors(np.asarray(samples))


def long_audio(seconds, seed=3):
    rng = np.random.default_rng(seed)
    return (0.1 * rng.standard_normal(int(seconds * 16000))).astype(np.float32)


class ColumnModel:
    """a model whose CTC posteriors are GIVEN (blank column `col`, the rest spread over one other token): drives
    `find_blank` on arbitrary blank patterns through both call forms"""
    dtype = "float32"
    device = "cpu"

    def __init__(self, col):
        col = np.asarray(col, np.float32)
        self._lpz = np.zeros((len(col), 4), np.float32)
        self._lpz[:, 0] = col
        self._lpz[:, 2] = 1.0 - col
        outer = self

        class _A:
            blank_id = 0
            token_list = ["<blank>", "<unk>", "a", "<sos/eos>"]

            def __init__(self):
                self.ctc = _Ctc()

            def encode(self, speech, length):
                enc = torch.zeros((1, outer._lpz.shape[0], 1))
                enc._lpz = torch.from_numpy(outer._lpz)[None]
                return enc, torch.tensor([outer._lpz.shape[0]])
        self.asr_model = _A()

    def ctc_posteriors(self, samples):
        return self._lpz


def blank_patterns(n_cases=120, seed=11):
    """(n_samples, blank column) cases for `find_blank`: runs that touch either end, ties in length, windows with fewer
    samples than frames, all-silent and all-speech windows"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_cases):
        T = int(rng.integers(1, 90))
        n = int(rng.integers(1, 40)) if k % 9 == 0 else int(rng.integers(T, 400 * T + 2))
        col = np.full(T, 0.5, np.float32)
        t = 0 if k % 4 == 0 else int(rng.integers(0, 6))
        while t < T:
            run = int(rng.integers(1, 12)) if k % 5 else 4
            col[t:t + run] = 0.99
            t += run + int(rng.integers(1, 9))
        if k % 7 == 0:
            col[:] = 0.99
        if k % 13 == 0:
            col[:] = 0.2
        if k % 3 == 0:
            col[-int(rng.integers(1, 5)):] = 0.995
        out.append((n, col))
    return out
