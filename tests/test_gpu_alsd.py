"""-m gpu: the device ALSD beam search (rs_rnnt_alsd, k_rnnt_alsd.hip) against oracle/rnnt_alsd.c — labels,
alignment steps AND scores of the best hypothesis BIT-EXACT (the C restatement documents the evaluation order of
every float; it is itself checked against the readable restatement oracle/alsd.py in tests/test_oracle_alsd.py).

Covers: beam 1 / 2 / 4 / 8, both recombination modes, score normalisation on and off, a float (multiple of T')
and an absolute label budget, ragged batches with zero-frame utterances, batches that are not a multiple of the
32-row joint tile, the 2-layer toy geometry and the 619M decoder geometry (V + 1 = 3001: 47 column tiles, ragged
last tile; K slices of five 16-blocks), and the strategy reached through load_model(decoding="alsd") /
transcribe_batch()."""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import TINY, WIDE2
from reazonspeech_amd.runtime.model import AsrModel
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import greedy as og

pytestmark = pytest.mark.gpu


def run_alsd(model, f, lens, beam, max_target_len, score_norm=True, merge=False, out_cap=None):
    B, Tp, _ = f.shape
    dev = model.device
    budget = int(max_target_len * Tp) if isinstance(max_target_len, float) else int(max_target_len)
    out_cap = out_cap or max(1, Tp + budget)
    ids = torch.zeros((B, out_cap), dtype=torch.int32, device=dev)
    steps = torch.zeros_like(ids)
    n_ids = torch.full((B,), -7, dtype=torch.int32, device=dev)
    scores = torch.full((B,), float("nan"), dtype=torch.float32, device=dev)
    ws = torch.empty((model.ctx.alsd_workspace_bytes(B, beam, Tp, max_target_len),), dtype=torch.uint8, device=dev)
    model.ctx.rnnt_alsd(f.to(dev), lens.to(dev), B, Tp, beam, max_target_len, score_norm, merge, ids, steps, n_ids, scores,
                        ws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = n_ids.cpu().numpy()
    sc = scores.cpu().numpy()
    return [(ids[b, :n[b]].cpu().tolist(), steps[b, :n[b]].cpu().tolist(), float(sc[b])) for b in range(B)]


def same_bits(a, b):
    return np.float32(a).tobytes() == np.float32(b).tobytes()


def check(got, ref):
    assert len(got) == len(ref)
    for b, (g, r) in enumerate(zip(got, ref)):
        assert g[0] == r[0], f"labels differ for utterance {b}: {g[0]} vs {r[0]}"
        assert g[1] == r[1], f"alignment steps differ for utterance {b}"
        assert same_bits(g[2], r[2]), f"score differs for utterance {b}: {g[2]!r} vs {r[2]!r}"


@pytest.fixture(scope="module")
def tiny(gpu_device):
    sd = synthetic_state_dict(TINY, 0)
    return AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0"), sd


@pytest.fixture(scope="module")
def wide(gpu_device):
    sd = synthetic_state_dict(WIDE2, 0, blank_bias=6.5)
    return AsrModel(WIDE2, sd, SyntheticTokenizer(WIDE2.vocab_size), device="cuda:0"), sd


@pytest.mark.parametrize("beam,max_target_len,norm,merge", [
    (1, 1.0, False, False), (2, 1.0, True, True), (4, 2.0, True, False), (4, 7, False, True), (8, 0.5, True, False),
    (3, 1.5, True, True),
])
def test_alsd_bit_exact_toy_geometry(tiny, beam, max_target_len, norm, merge):
    model, sd = tiny
    cfg = model.cfg
    g = torch.Generator().manual_seed(100 + beam)
    B, Tp = 37, 21
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * (0.7 + 0.6 * torch.rand((B, 1, 1), generator=g))
    lens = torch.randint(1, Tp + 1, (B,), generator=g, dtype=torch.int32)
    lens[3] = 0
    lens[5] = Tp
    lens[36] = 1
    ref = og.rnnt_alsd(cfg, sd, f.numpy(), lens.numpy(), beam=beam, max_target_len=max_target_len, score_norm=norm,
                       recombine="merge" if merge else "upstream", out_cap=Tp + 64)
    got = run_alsd(model, f, lens, beam, max_target_len, norm, merge, out_cap=Tp + 64)
    assert sum(len(r[0]) for r in ref) > 40, "fixture should emit labels"
    assert got[3] == ([], [], 0.0)
    check(got, ref)


@pytest.mark.parametrize("beam,merge", [(4, False), (2, True)])
def test_alsd_bit_exact_at_619m_geometry(wide, beam, merge):
    model, sd = wide
    cfg = model.cfg
    assert (cfg.pred_hidden, cfg.joint_hidden, cfg.n_logits) == (640, 640, 3001)
    g = torch.Generator().manual_seed(11 + beam)
    B, Tp = 7, 12
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * (0.8 + 0.4 * torch.rand((B, 1, 1), generator=g))
    lens = torch.randint(3, Tp + 1, (B,), generator=g, dtype=torch.int32)
    lens[2] = Tp
    ref = og.rnnt_alsd(cfg, sd, f.numpy(), lens.numpy(), beam=beam, max_target_len=1.0, recombine="merge" if merge else "upstream")
    got = run_alsd(model, f, lens, beam, 1.0, True, merge)
    toks = [t for r in ref for t in r[0]]
    assert len(toks) > 12 and len(set(toks)) > 5, "fixture should emit a varied label stream"
    check(got, ref)


def test_alsd_recombination_happens(tiny):
    """the fixture must actually exercise recombination (the same labels reached through different alignments),
    otherwise the merge / upstream modes are not being compared: on some utterance the two modes disagree in score"""
    model, sd = tiny
    g = torch.Generator().manual_seed(21)
    B, Tp = 24, 16
    f = torch.randn((B, Tp, model.cfg.joint_hidden), generator=g) * 0.6
    lens = torch.full((B,), Tp, dtype=torch.int32)
    a = run_alsd(model, f, lens, 4, 1.0, True, False)
    b = run_alsd(model, f, lens, 4, 1.0, True, True)
    check(a, og.rnnt_alsd(model.cfg, sd, f.numpy(), lens.numpy(), beam=4, max_target_len=1.0, recombine="upstream"))
    check(b, og.rnnt_alsd(model.cfg, sd, f.numpy(), lens.numpy(), beam=4, max_target_len=1.0, recombine="merge"))
    assert any(x != y for x, y in zip(a, b)), "recombination never changed a result in this fixture"


def test_alsd_argument_errors(tiny):
    model, _ = tiny
    f = torch.zeros((1, 4, model.cfg.joint_hidden))
    lens = torch.tensor([4], dtype=torch.int32)
    with pytest.raises(capi.RsError) as e:
        run_alsd(model, f, lens, 9, 1.0)
    assert e.value.code == capi.RS_EINVAL
    g = torch.Generator().manual_seed(4)
    f = torch.randn((8, 16, model.cfg.joint_hidden), generator=g)
    lens = torch.full((8,), 16, dtype=torch.int32)
    longest = max(len(r[0]) for r in run_alsd(model, f, lens, 2, 1.0))
    assert longest >= 2
    with pytest.raises(capi.RsError) as e:                      # a result that does not fit out_cap
        run_alsd(model, f, lens, 2, 1.0, out_cap=longest - 1)
    assert e.value.code == capi.RS_EOVERFLOW


def test_transcribe_with_alsd_strategy(gpu_device):
    """load_model(decoding="alsd") -> transcribe_batch(): the hypotheses are the oracle search over the device's own
    joint-encoder frames; the Hypothesis keeps the reference's ALSD conventions (leading blank, step - idx - 1)"""
    from reazonspeech_amd.nemo.asr import TranscribeConfig, audio_from_numpy, load_model, transcribe_batch
    model = load_model(device="cuda:0", config=TINY, decoding="alsd", beam_size=3)
    assert model.cfg.decoding == "alsd" and model.cfg.beam_size == 3
    audio, lens = synthetic_batch(5, 2.0, seed=13)
    audios = [audio_from_numpy(audio[b, :lens[b] - 1500 * b], 16000) for b in range(5)]
    res = transcribe_batch(model, audios, TranscribeConfig(raw_hypothesis=True, verbose=False))
    # the same batch again, keeping the device joint_enc to run the oracle search on
    waves = [np.ascontiguousarray(a.waveform, dtype=np.float32) for a in audios]
    buf = model.stage(waves)
    model.run_device(buf)
    dec = model.collect(buf)
    sd = synthetic_state_dict(TINY, 0)
    ref = og.rnnt_alsd(TINY, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy(), beam=3,
                       max_target_len=TINY.alsd_max_target_len, score_norm=True)
    n_tok = 0
    for b in range(5):
        hyp = res[b].hypothesis
        assert hyp.y_sequence.tolist()[0] == TINY.blank_id
        assert hyp.y_sequence.tolist()[1:] == ref[b][0] == dec.ids[b]
        assert [s - idx - 1 for idx, s in enumerate(hyp.timestamp)] == [i - u for u, i in enumerate(ref[b][1])] == dec.frames[b]
        assert same_bits(hyp.score, ref[b][2])
        n_tok += len(ref[b][0])
    assert n_tok > 5
    # and the single-utterance entry point agrees with the batch
    from reazonspeech_amd.nemo.asr import transcribe
    one = transcribe(model, audios[1])
    assert one.text == res[1].text and [s.seconds for s in one.subwords] == [s.seconds for s in res[1].subwords]


def test_alsd_long_list_through_the_host_pipeline(gpu_device):
    """more utterances than max_batch with the beam search: length-sorted groups through the persistent host pipeline
    (narrowed buffer views, two decode lanes, hypotheses and float32 scores copied back by the decode workers) give,
    per utterance, exactly what decoding it alone gives — ids, frames and score bits"""
    from reazonspeech_amd.nemo.asr import load_model
    model = load_model(device="cuda:0", config=TINY, decoding="alsd", beam_size=2)
    audio, lens = synthetic_batch(11, 2.0, seed=21, ragged=True, min_seconds=0.4)
    waves = [audio[b, :lens[b]] for b in range(11)]
    got = model.transcribe_waveforms(waves, max_batch=4)
    assert got.scores is not None and len(got.ids) == 11
    for b in (0, 4, 7, 10):
        alone = model.transcribe_waveforms([waves[b]])
        assert got.ids[b] == alone.ids[0] and got.frames[b] == alone.frames[0]
        assert same_bits(got.scores[b], alone.scores[0])
