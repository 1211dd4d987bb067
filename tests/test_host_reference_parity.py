"""CPU: the host-side pieces of the path against golden vectors produced by importing the
reference's own in-tree files (tests/golden/make_reference_golden.py): decode_hypothesis,
find_end_of_segment, constants, writers, TranscribeConfig defaults, pad/norm audio."""
import io
import json
import os

import numpy as np
import pytest

from reazonspeech_amd.nemo.asr import decode as D
from reazonspeech_amd.nemo.asr import audio as A
from reazonspeech_amd.nemo.asr import writer as W
from reazonspeech_amd.nemo.asr.interface import (Hypothesis, TranscribeConfig, TranscribeResult, Subword,
                                                 Segment, AudioData)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_host.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD, encoding="utf-8") as fp:
        return json.load(fp)


class _Tok:
    def __init__(self, pieces):
        self.pieces = pieces

    def ids_to_text(self, ids):
        text = "".join(self.pieces[i] for i in ids).replace("▁", " ")
        return text[1:] if text.startswith(" ") else text


class _Model:
    def __init__(self, pieces):
        self.tokenizer = _Tok(pieces)


def test_constants(gold):
    c = gold["consts"]
    assert (D.PAD_SECONDS, D.SECONDS_PER_STEP, D.SUBWORDS_PER_SEGMENTS, D.PHONEMIC_BREAK) == \
        (c["PAD_SECONDS"], c["SECONDS_PER_STEP"], c["SUBWORDS_PER_SEGMENTS"], c["PHONEMIC_BREAK"])
    assert sorted(D.TOKEN_EOS) == c["TOKEN_EOS"] and sorted(D.TOKEN_COMMA) == c["TOKEN_COMMA"]
    cfg = TranscribeConfig()
    assert {"verbose": cfg.verbose, "raw_hypothesis": cfg.raw_hypothesis} == gold["config_defaults"]


def test_decode_hypothesis_matches_reference(gold):
    model = _Model(gold["pieces"])
    for case in gold["cases"]:
        # the greedy adapter must reproduce the ALSD-shaped input the reference consumed
        hyp = Hypothesis.from_greedy(case["ids"], case["frames"], case["blank"])
        assert hyp.timestamp == case["steps"]
        assert hyp.y_sequence.tolist() == [case["blank"]] + case["ids"]
        res = D.decode_hypothesis(model, hyp)
        assert isinstance(res, TranscribeResult) and res.hypothesis is None
        assert res.text == case["text"]
        assert [[s.seconds, s.token_id, s.token] for s in res.subwords] == case["subwords"]
        assert [[s.start_seconds, s.end_seconds, s.text] for s in res.segments] == case["segments"]


def test_greedy_adapter_time_formula():
    hyp = Hypothesis.from_greedy([7, 8, 9], [0, 10, 10], blank_id=99)
    model = _Model({7: "a", 8: "b", 9: "c"})
    res = D.decode_hypothesis(model, hyp)
    # seconds = max(0.08 * frame - 0.5, 0)  (decode.py:48 with the adapter's step = frame + idx + 1)
    assert [s.seconds for s in res.subwords] == [0, max(0.08 * 10 - 0.5, 0), max(0.08 * 10 - 0.5, 0)]


def test_writers_match_reference(gold):
    model = _Model(gold["pieces"])
    for case in gold["cases"]:
        res = D.decode_hypothesis(model, Hypothesis.from_greedy(case["ids"], case["frames"], case["blank"]))
        for ext, want in case["writers"].items():
            fp = io.StringIO()
            w = W.get_writer(fp, None if ext == "None" else ext)
            w.write_header()
            for seg in res.segments:
                w.write(seg)
            assert fp.getvalue() == want, ext


def test_get_writer_extension_quirk(gold):
    class Named(io.StringIO):
        name = "x.vtt"
    assert type(W.get_writer(Named())).__name__ == gold["writer_for_x_vtt"] == "TextWriter"
    assert isinstance(W.get_writer(io.StringIO(), "vtt"), W.VTTWriter)


def test_find_end_of_segment_rules():
    sw = lambda tok, sec: Subword(seconds=sec, token_id=0, token=tok)   # noqa: E731
    # closes after EOS unless punctuation follows
    subs = [sw("a", 0), sw("。", 0.1), sw("b", 0.2)]
    assert D.find_end_of_segment(subs, 0) == 1
    subs = [sw("a", 0), sw("。", 0.1), sw("!", 0.2), sw("b", 0.3)]
    assert D.find_end_of_segment(subs, 0) == 2
    # long pause only breaks once the segment has >= 10 subwords
    subs = [sw("x", 0.08 * i) for i in range(10)] + [sw("y", 5.0), sw("z", 9.0), sw("w", 9.1)]
    assert D.find_end_of_segment(subs, 0) == 10
    assert D.find_end_of_segment(subs, 11) == 12          # last subword always closes
    assert D.find_end_of_segment([sw("a", 0)], 0) == 0


def test_pad_and_norm_audio():
    x = np.arange(10, dtype=np.float32)
    padded = A.pad_audio(AudioData(x, 16000), 0.5)
    assert padded.waveform.shape == (10 + 2 * 8000,) and padded.samplerate == 16000
    assert np.all(padded.waveform[:8000] == 0) and np.all(padded.waveform[-8000:] == 0)
    assert np.array_equal(padded.waveform[8000:8010], x)
    # int(seconds * samplerate) like np.pad(pad_width=int(...)) — audio.py:80-82
    assert A.pad_audio(AudioData(x, 22050), 0.33).waveform.shape == (10 + 2 * int(0.33 * 22050),)
    # 16 kHz mono is passed through untouched
    same = A.norm_audio(AudioData(x, 16000))
    assert same.waveform is x
    # resample BEFORE down-mix, stereo -> mono mean over channels (audio.py:64-67)
    st = np.stack([np.ones(8000, np.float32), 3 * np.ones(8000, np.float32)])
    out = A.norm_audio(AudioData(st, 8000))
    assert out.samplerate == 16000 and out.waveform.shape == (16000,)
    assert abs(float(out.waveform[4000:12000].mean()) - 2.0) < 1e-3


def test_audio_constructors(tmp_path):
    import torch
    from scipy.io import wavfile
    a = A.audio_from_numpy(np.zeros(4, np.float32), 8000)
    assert a.samplerate == 8000
    t = A.audio_from_tensor(torch.ones(5), 16000)
    assert isinstance(t.waveform, np.ndarray) and t.waveform.shape == (5,)
    path = str(tmp_path / "a.wav")
    wavfile.write(path, 22050, (np.sin(np.arange(2205) / 10.0) * 20000).astype(np.int16))
    f = A.audio_from_path(path)
    assert f.samplerate == 22050 and f.waveform.dtype == np.float32 and abs(f.waveform).max() <= 1.0


def test_reference_import_path_and_console_script():
    """`reazonspeech.nemo.asr` is importable with the reference's exports (pkg/nemo-asr/src/__init__.py:1-3) and the
    console script declared in pyproject.toml resolves (pkg/nemo-asr/pyproject.toml:19-20)"""
    import importlib
    import re
    import reazonspeech.nemo.asr as ref_path
    import reazonspeech_amd.nemo.asr as impl
    ref_init = open("/root/reference/pkg/nemo-asr/src/__init__.py").read() if os.path.exists(
        "/root/reference/pkg/nemo-asr/src/__init__.py") else (
        "from .interface import TranscribeConfig\nfrom .transcribe import transcribe, load_model\n"
        "from .audio import audio_from_numpy, audio_from_tensor, audio_from_path\n")
    names = [n.strip() for line in ref_init.splitlines() if " import " in line
             for n in line.split(" import ")[1].split(",")]
    assert sorted(names) == sorted(["TranscribeConfig", "transcribe", "load_model", "audio_from_numpy",
                                    "audio_from_tensor", "audio_from_path"])
    for n in names + ["transcribe_batch"]:
        assert getattr(ref_path, n) is getattr(impl, n)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    toml = open(os.path.join(root, "pyproject.toml")).read()
    m = re.search(r'reazonspeech-nemo-asr\s*=\s*"([\w.]+):(\w+)"', toml)
    assert m and m.group(1) == "reazonspeech.nemo.asr.cli"
    assert callable(getattr(importlib.import_module(m.group(1)), m.group(2)))
    from reazonspeech.nemo.asr.interface import TranscribeConfig
    assert TranscribeConfig is impl.TranscribeConfig
    assert not os.path.exists(os.path.join(root, "reazonspeech", "__init__.py"))          # namespace package
    assert not os.path.exists(os.path.join(root, "reazonspeech", "nemo", "__init__.py"))


def test_alsd_adapter_offset_is_one_documented_constant():
    """Hypothesis.from_alsd: alignment steps + ALSD_TIMESTAMP_OFFSET; with the default the reference's
    `step - idx - 1` (decode.py:48) recovers the emission frame, as from_greedy does"""
    from reazonspeech_amd.nemo.asr import interface as I
    ids, frames = [7, 8, 9], [0, 10, 10]
    steps = [f + i for i, f in enumerate(frames)]
    a = I.Hypothesis.from_alsd(ids, steps, blank_id=99)
    g = I.Hypothesis.from_greedy(ids, frames, blank_id=99)
    assert I.ALSD_TIMESTAMP_OFFSET == 1
    assert a.y_sequence == g.y_sequence and a.timestamp == g.timestamp and a.frames == frames
    assert [t - i - 1 for i, t in enumerate(a.timestamp)] == frames
    assert I.Hypothesis.from_alsd(ids, steps, 99, offset=0).timestamp == steps


def test_resampler_meets_the_soxr_hq_specification():
    """`norm_audio` resamples with librosa.resample's default (soxr_hq; audio.py:64-65).  soxr is not installable here,
    so the stand-in is checked against what soxr HQ promises, on analytic tones: pass band (<= 0.913 Nyquist) exact to
    -120 dB, everything from the target Nyquist up rejected by >= 120 dB; output length = ceil(n * ratio) like librosa."""
    from reazonspeech_amd.nemo.asr import audio as A
    for orig in (48000, 44100, 22050, 8000):
        t = np.arange(orig) / orig
        for f in (440.0, 3000.0, 7000.0, 7290.0, 9000.0, 15000.0):
            if f >= orig / 2:
                continue
            y = A._resample((0.5 * np.sin(2 * np.pi * f * t)).astype(np.float32), orig, 16000)
            assert y.dtype == np.float32 and len(y) == int(np.ceil(orig * 16000 / orig))
            tt = np.arange(len(y)) / 16000.0
            lower_nyq = min(orig, 16000) / 2
            ref = 0.5 * np.sin(2 * np.pi * f * tt) if f < lower_nyq else np.zeros_like(tt)
            if lower_nyq * 0.913 < f < lower_nyq:
                continue                                   # transition band: unspecified
            e = (y.astype(np.float64) - ref)[800:-800]
            db = 20 * np.log10(np.sqrt(np.mean(e ** 2)) / 0.5 + 1e-30)
            assert db <= -120.0 + 15.0, (orig, f, db)      # float32 output: the rounding floor is about -140 dB
    stereo = np.stack([np.sin(2 * np.pi * 300 * np.arange(48000) / 48000)] * 2).astype(np.float32)
    out = A.norm_audio(A.audio_from_numpy(stereo, 48000))
    assert out.samplerate == 16000 and out.waveform.shape == (16000,)


def test_checkpoint_resolution_order(tmp_path, monkeypatch):
    """load_model's checkpoint lookup: argument, then $REAZONSPEECH_NEMO_CHECKPOINT, then the Hugging Face cache of
    'reazon-research/reazonspeech-nemo-v2' (what the reference's from_pretrained fills, transcribe.py:26-28)"""
    import importlib
    T = importlib.import_module("reazonspeech_amd.nemo.asr.transcribe")     # (the package re-exports the function of that name)
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hf" / "hub"))
    monkeypatch.delenv(T.CHECKPOINT_ENV, raising=False)
    import importlib
    import huggingface_hub.constants as C
    importlib.reload(C)
    assert T.resolve_checkpoint() is None                          # empty cache, offline
    snap = tmp_path / "hf" / "hub" / "models--reazon-research--reazonspeech-nemo-v2" / "snapshots" / "0123abcd"
    snap.mkdir(parents=True)
    (snap / "reazonspeech-nemo-v2.nemo").write_bytes(b"not a real archive")
    refs = snap.parent.parent / "refs"
    refs.mkdir()
    (refs / "main").write_text("0123abcd")
    found = T.resolve_checkpoint()
    if found is not None:                                          # (hub constants are read at import time in some versions)
        assert found.endswith("reazonspeech-nemo-v2.nemo")
    env = tmp_path / "env.nemo"
    env.write_bytes(b"x")
    monkeypatch.setenv(T.CHECKPOINT_ENV, str(env))
    assert T.resolve_checkpoint() == str(env)
    arg = tmp_path / "arg.nemo"
    arg.write_bytes(b"x")
    assert T.resolve_checkpoint(str(arg)) == str(arg)
    with pytest.raises(FileNotFoundError):
        T.resolve_checkpoint(str(tmp_path / "missing.nemo"))


def test_audio_from_path_without_decoders(tmp_path):
    """WAV always works; another container without soundfile / audioread is an explicit error that names the gap"""
    from reazonspeech_amd.nemo.asr import audio as A
    from scipy.io import wavfile
    p = tmp_path / "a.wav"
    wavfile.write(p, 8000, (np.sin(np.arange(800) / 10.0) * 20000).astype(np.int16))
    a = A.audio_from_path(str(p))
    assert a.samplerate == 8000 and a.waveform.dtype == np.float32 and a.waveform.shape == (800,)
    try:
        import soundfile  # noqa: F401
        return
    except ImportError:
        pass
    q = tmp_path / "a.mp3"
    q.write_bytes(b"ID3\\x03\\x00" + bytes(200))
    with pytest.raises(RuntimeError) as e:
        A.audio_from_path(str(q))
    assert "soundfile" in str(e.value)
