"""-m gpu: the ESPnet2 Conformer-Transducer path of `reazonspeech.espnet.asr` through the C ABI against
its CPU oracle (oracle/espnet.py — PARITY UNPINNED against ESPnet itself, which cannot run here; the conformer block it
shares with oracle/model.py is pinned to transformers' parakeet).

Stated tolerances (bf16 GEMM operands and stored activations, f32 accumulation; the oracle rounds at the same points):
  front-end features (DefaultFrontend + GlobalMVN)          max |err| <= 2e-3
  encoder output after after_norm, joint projection         toy: max <= 6e-2, mean <= 6e-3;  120M (17 blocks): max <= 0.08, mean <= 0.01
  CTC posteriors (probabilities)                            max |err| <= 3e-2
  transducer greedy search (tanh joint, one symbol / frame) ids and frames BIT-EXACT vs oracle/rnnt_greedy.c on the same projection
"""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.config import ESPNET_TINY, ESPNET_CONFORMER_120M
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet
from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list
from reazonspeech_amd.espnet.asr import ctc as ectc, interface
from oracle import espnet as oe, greedy as og
import importlib

etr = importlib.import_module("reazonspeech_amd.espnet.asr.transcribe")
pytestmark = pytest.mark.gpu


def build(cfg, seed):
    sd = synthetic_state_dict_espnet(cfg, seed)
    return EspnetModel(cfg, sd, synthetic_token_list(cfg.vocab_size, seed), device="cuda:0"), sd


@pytest.fixture(scope="module")
def tiny(gpu_device):
    return build(ESPNET_TINY, 3)


def run(model, audio, lens, want_ctc=True):
    am = model.am
    buf = am.stage([audio[b, :int(lens[b])] for b in range(audio.shape[0])])
    M = buf.B * buf.tp_max
    enc = torch.zeros((buf.B, buf.tp_max, am.cfg.d_model), dtype=torch.float32, device=am.device)
    probs = torch.zeros((M, am.cfg.n_logits), dtype=torch.float32, device=am.device)
    blank = torch.zeros((M,), dtype=torch.float32, device=am.device)
    am.ctx.set_ctc_out(probs if want_ctc else None, blank)
    try:
        am.run_device(buf, want_enc=enc)
        torch.cuda.synchronize()
    finally:
        am.ctx.set_ctc_out(None, None)
    return buf, enc.cpu(), probs.view(buf.B, buf.tp_max, -1).cpu(), blank.view(buf.B, buf.tp_max).cpu(), am.collect(buf)


def compare(cfg, sd, model, audio, lens, tol_max, tol_mean):
    buf, enc, probs, blank, got = run(model, audio, lens)
    ref = oe.forward(cfg, sd, torch.from_numpy(audio), torch.from_numpy(lens), "bf16-fused-glu", taps := {})
    assert buf.n_frames.cpu().tolist() == taps["n_frames"].tolist()
    assert got.enc_lens == ref["enc_lens"].tolist()
    feats = buf.feats.cpu()
    stats = {}
    for b in range(len(lens)):
        nf, n = int(taps["n_frames"][b]), int(ref["enc_lens"][b])
        assert (feats[b, :nf] - taps["feats"][b, :nf]).abs().max() <= 2e-3
        assert torch.all(feats[b, nf:] == 0)
        for name, a, r in (("enc", enc, ref["enc"]), ("joint", buf.joint_enc.cpu(), ref["joint_enc"])):
            d = (a[b, :n] - r[b, :n]).abs()
            stats[name] = max(stats.get(name, 0.0), d.max().item())
            assert d.max() <= tol_max and d.mean() <= tol_mean, (name, b, d.max().item(), d.mean().item())
        dp = (probs[b, :n] - ref["ctc"][b, :n]).abs().max().item()
        stats["ctc"] = max(stats.get("ctc", 0.0), dp)
        assert dp <= 3e-2, (b, dp)
        assert torch.equal(blank[b, :n], probs[b, :n, cfg.blank_id])
        assert abs(float(probs[b, :n].sum(-1).mean()) - 1.0) < 1e-4
    same = og.rnnt_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), np.asarray(got.enc_lens, np.int32))
    assert got.ids == [r[0] for r in same] and got.frames == [r[1] for r in same]
    assert all(len(set(f)) == len(f) for f in got.frames), "ESPnet greedy emits at most one symbol per frame"
    return stats, got


def test_tiny_pipeline_vs_oracle(tiny):
    model, sd = tiny
    audio, lens = synthetic_batch(4, 3.0, seed=5, ragged=True, min_seconds=0.7)
    stats, got = compare(ESPNET_TINY, sd, model, audio, lens, 6e-2, 6e-3)
    assert sum(len(x) for x in got.ids) > 20
    print("espnet tiny:", stats, [len(x) for x in got.ids])


def test_tiny_batch_invariance_bits(tiny):
    model, sd = tiny
    audio, lens = synthetic_batch(5, 3.0, seed=9, ragged=True, min_seconds=0.5)
    buf, enc, probs, _, together = run(model, audio, lens)
    for b in (0, 3):
        _, e1, p1, _, alone = run(model, audio[b:b + 1], lens[b:b + 1])
        n = alone.enc_lens[0]
        assert torch.equal(e1[0, :n], enc[b, :n]) and torch.equal(p1[0, :n], probs[b, :n])
        assert alone.ids[0] == together.ids[b] and alone.frames[0] == together.frames[b]


def test_screened_joint_is_bit_identical_with_the_tanh_joint(tiny):
    """the screened joint (bf16 screening product, exact float32 re-evaluation of every column that can still be the argmax) serves
    the tanh JointNetwork too: its error bound only needs a = act(f + g) computed exactly before it is rounded"""
    model, sd = tiny
    am = model.am
    audio, lens = synthetic_batch(6, 3.0, seed=21, ragged=True, min_seconds=0.5)
    buf = am.stage([audio[b, :int(lens[b])] for b in range(6)])
    am.run_device(buf)
    torch.cuda.synchronize()
    outs = []
    for screen in (0, 1):
        am.ctx.set_option("decode_screen", screen)
        am.decode(am.ctx, buf, buf.ws, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(am.collect(buf))
    want = og.rnnt_greedy(ESPNET_TINY, sd, buf.joint_enc.cpu().numpy(), np.asarray(outs[0].enc_lens, np.int32))
    assert outs[0].ids == outs[1].ids == [r[0] for r in want] and outs[0].frames == outs[1].frames == [r[1] for r in want]
    assert sum(len(x) for x in outs[0].ids) > 10


def test_120m_geometry_vs_oracle(gpu_device):
    """d = 512, 8 heads of 64, FFN 2048, kernel 31, 17 blocks, Conv2dSubsampling with 512 channels (the dense 3x3 conv as a
    GEMM over gathered patches, in chunks), vocabulary 2600: two ragged utterances with the reference's (16000, 8000) padding"""
    cfg = ESPNET_CONFORMER_120M
    model, sd = build(cfg, 0)
    audio, lens = synthetic_batch(2, 3.0, seed=123, ragged=True, min_seconds=1.5)
    padded = np.zeros((2, audio.shape[1] + 24000), np.float32)
    for b in range(2):
        padded[b, 16000:16000 + lens[b]] = audio[b, :lens[b]]
    stats, got = compare(cfg, sd, model, padded, lens + 24000, 0.08, 0.01)
    print("espnet 120M:", stats, [len(x) for x in got.ids])
    am = model.am                                   # the screened joint at V = 2600 (48 x 64 columns in registers), tanh
    buf = am.stage([padded[b, :int(lens[b]) + 24000] for b in range(2)])
    am.run_device(buf)
    torch.cuda.synchronize()
    outs = []
    for screen in (0, 1):
        am.ctx.set_option("decode_screen", screen)
        am.decode(am.ctx, buf, buf.ws, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(am.collect(buf))
    assert outs[0].ids == outs[1].ids == got.ids and outs[0].frames == outs[1].frames == got.frames


def test_model_object_answers_the_reference_call_forms(tiny):
    """`model(padded)[0][0]`, `model.asr_model.encode` + `.ctc.softmax`, `.blank_id`, `.token_list` (what pkg/espnet-asr's
    transcribe.py / ctc.py call) agree with the direct forms, and a 47 s recording goes through the windowing loop"""
    model, sd = tiny
    wav = synthetic_batch(1, 47.0, seed=77)[0][0]
    lpz = model.ctc_posteriors(wav[:32000])
    speech = torch.tensor(wav[:32000]).unsqueeze(0)
    enc = model.asr_model.encode(speech, speech.new_full([1], dtype=torch.long, fill_value=speech.size(1)))[0]
    lpz2 = model.asr_model.ctc.softmax(enc).detach().squeeze(0).cpu().numpy()
    assert np.array_equal(lpz, lpz2) and lpz.shape[1] == len(model.asr_model.token_list) == ESPNET_TINY.vocab_size
    assert model.asr_model.blank_id == 0
    text = model(np.pad(wav[:32000], etr.PADDING))[0][0]
    assert text == model.recognize(wav[:32000]) and isinstance(text, str)
    res = etr.transcribe(model, interface.AudioData(wav, 16000), interface.TranscribeConfig(verbose=False))
    assert isinstance(res.text, str) and len(res.segments) >= 1
    assert all(0.0 <= s.start_seconds <= s.end_seconds <= 47.0 + 1e-6 for s in res.segments)
    assert "".join(s.text for s in res.segments) == res.text
    batch = etr.transcribe_batch(model, [interface.AudioData(wav[:48000], 16000), interface.AudioData(wav[48000:80000], 16000)])
    assert batch[0].text == model.recognize(wav[:48000]) and batch[1].text == model.recognize(wav[48000:80000])
    blank = ectc.find_blank(model, wav[:20 * 16000])
    assert 0 <= blank.start <= blank.end <= 20 * 16000
