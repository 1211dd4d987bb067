"""-m gpu: the default transducer beam search (reazonspeech_amd/csrc/k_rnnt_beam.hip — what Speech2Text runs for
reazonspeech.espnet.asr: beam_size 20, score_norm) through the C ABI against oracle/espnet_beam.c on the same joint-encoder
projection: labels, scores (float32, BIT-EXACT) and the number of prediction-network evaluations per utterance.
oracle/espnet_beam.c itself follows the torch restatement of ESPnet's algorithm (tests/test_oracle_espnet_beam.py).

The synthetic checkpoints here use dec_gain = 8 and their own blank offset (see tests/test_oracle_espnet_beam.py for why)."""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.config import ESPNET_TINY, ESPNET_CONFORMER_120M
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet
from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list
from oracle import greedy as og

pytestmark = pytest.mark.gpu


def build(cfg, seed, bias, beam_size=1, max_pops=0):
    sd = synthetic_state_dict_espnet(cfg, seed, blank_bias=bias, dec_gain=8.0)
    return EspnetModel(cfg, sd, synthetic_token_list(cfg.vocab_size, seed), device="cuda:0", beam_size=beam_size, max_pops=max_pops), sd


def encode(model, audio, lens):
    """front-end + encoder on the device -> the staged buffers (joint_enc, enc_lens in HBM)"""
    am = model.am
    buf = am.stage([audio[b, :int(lens[b])] for b in range(audio.shape[0])])
    with torch.cuda.device(am.device):
        stream = torch.cuda.current_stream().cuda_stream
        am.ctx.frontend(buf.audio, buf.lens, 0, 0, buf.t_max, buf.feats, buf.n_frames, buf.ws, stream)
        am.ctx.encoder(buf.feats, buf.n_frames, buf.B, buf.t_max, None, buf.joint_enc, buf.enc_lens, buf.ws, stream)
        torch.cuda.synchronize()
    return buf


def device_beam(model, buf, beam, score_norm=True, max_pops=0, enc_lens=None, out_cap=None, with_frames=False):
    am = model.am
    B = buf.B
    dev = am.device
    el = buf.enc_lens if enc_lens is None else torch.as_tensor(enc_lens, dtype=torch.int32, device=dev)
    cap = out_cap or (2 * buf.tp_max + 16)
    ids = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
    frames = torch.full((B, cap), -7, dtype=torch.int32, device=dev) if with_frames else None
    n_ids = torch.full((B,), -7, dtype=torch.int32, device=dev)
    scores = torch.full((B,), 123.0, dtype=torch.float32, device=dev)
    pops = torch.full((B,), -7, dtype=torch.int32, device=dev)
    ws = torch.empty((am.ctx.beam_workspace_bytes(B, beam, buf.tp_max, max_pops),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        am.ctx.rnnt_beam(buf.joint_enc, el, B, buf.tp_max, beam, score_norm, max_pops, ids, n_ids, scores, pops, ws,
                         torch.cuda.current_stream().cuda_stream, frames=frames)
        torch.cuda.synchronize()
    n = n_ids.cpu().numpy()
    if with_frames:
        return [(ids[b, :n[b]].cpu().tolist(), frames[b, :n[b]].cpu().tolist(), float(scores[b]), int(pops[b])) for b in range(B)]
    return [(ids[b, :n[b]].cpu().tolist(), float(scores[b]), int(pops[b])) for b in range(B)]


@pytest.fixture(scope="module")
def tiny(gpu_device):
    model, sd = build(ESPNET_TINY, 11, 12.0)
    audio, lens = synthetic_batch(6, 2.0, seed=21)
    lens = lens.copy()
    lens[1] = lens[1] // 2
    lens[4] = 3000
    buf = encode(model, audio, lens)
    buf.sample_lens = lens
    return model, sd, buf


@pytest.mark.parametrize("beam", [1, 2, 4, 20])
def test_tiny_bit_exact(tiny, beam):
    model, sd, buf = tiny
    f = buf.joint_enc.cpu().numpy()
    el = buf.enc_lens.cpu().numpy()
    want = og.espnet_beam(model.cfg, sd, f, el, beam=beam, max_pops=16 * beam, out_cap=2 * buf.tp_max + 16, with_frames=True)
    got = device_beam(model, buf, beam, with_frames=True)
    assert [g[0] for g in got] == [w[0] for w in want]
    assert [g[1] for g in got] == [w[1] for w in want]            # the frame each label was appended at
    assert [g[3] for g in got] == [w[3] for w in want]
    assert [np.float32(g[2]) for g in got] == [np.float32(w[2]) for w in want]
    assert sum(len(g[0]) for g in got) > 0


@pytest.mark.parametrize("ks", ["1", "2", "8"])
def test_guesses_do_not_change_results(tiny, monkeypatch, ks):
    """how many expansions are asked for ahead of need per iteration ($RS_BEAM_SPEC, default 3) is a scheduling matter"""
    model, sd, buf = tiny
    want = device_beam(model, buf, 10)
    monkeypatch.setenv("RS_BEAM_SPEC", ks)
    assert device_beam(model, buf, 10) == want


def test_record_kernel_lds_variant(tiny, monkeypatch):
    """vocabularies past 3072 entries take the record kernel that re-reads the row from LDS; forced here on the small one"""
    model, sd, buf = tiny
    want = device_beam(model, buf, 6)
    monkeypatch.setenv("RS_BEAM_RECORD_LDS", "1")
    assert device_beam(model, buf, 6) == want


def test_tiny_score_norm_off_and_empty_rows(tiny):
    model, sd, buf = tiny
    f = buf.joint_enc.cpu().numpy()
    el = buf.enc_lens.cpu().numpy().copy()
    el[2] = 0
    el[5] = 1
    want = og.espnet_beam(model.cfg, sd, f, el, beam=5, score_norm=False, max_pops=80, out_cap=2 * buf.tp_max + 16)
    got = device_beam(model, buf, 5, score_norm=False, enc_lens=el)
    assert got == [(w[0], float(np.float32(w[1])), w[2]) for w in want]
    assert got[2] == ([], 0.0, 0)


def test_results_do_not_depend_on_the_batch(tiny):
    """an utterance searched alone gives what it gives inside the batch (rows are independent state machines)"""
    model, sd, buf = tiny
    full = device_beam(model, buf, 8)
    am = model.am
    for b in (0, 3):
        T = int(buf.enc_lens[b])
        one = am.stage([np.zeros(int(buf.sample_lens[b]), np.float32)])   # a 1-row buffer set; its joint_enc is overwritten below
        assert one.tp_max >= T
        one.joint_enc.zero_()
        one.joint_enc[0, :T] = buf.joint_enc[b, :T]
        got = device_beam(model, one, 8, enc_lens=[T])
        assert got[0] == full[b]


def test_overflow_is_reported_not_truncated(tiny):
    model, sd, buf = tiny
    with pytest.raises(RuntimeError, match="max_pops"):
        device_beam(model, buf, 8, max_pops=9)
    with pytest.raises(RuntimeError, match="out_cap"):
        device_beam(model, buf, 4, out_cap=1)
    # and the context is usable afterwards
    assert device_beam(model, buf, 2)


def test_model_call_uses_the_beam_search(gpu_device):
    model, sd = build(ESPNET_TINY, 11, 12.0, beam_size=6)
    audio, lens = synthetic_batch(2, 1.5, seed=4)
    wav = audio[0, :int(lens[0])]
    text, tokens, ids, _ = model(wav)[0]
    buf = encode(model, wav[None, :], np.asarray([len(wav)], np.int32))
    want = og.espnet_beam(model.cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy(), beam=6, max_pops=96,
                          out_cap=2 * buf.tp_max + 16, with_frames=True)
    assert ids == want[0][0] and text == "".join(model.token_list[i] for i in ids)
    res = model.am.transcribe_waveforms([audio[b, :int(lens[b])] for b in range(2)])
    assert res.ids[0] == ids and res.scores is not None and len(res.scores) == 2
    assert res.frames[0] == want[0][1]


def test_120m_one_utterance_bit_exact(gpu_device):
    """the published shape (V = 2600, joint 640, prediction net 512), beam 20, 4 s of audio"""
    cfg = ESPNET_CONFORMER_120M
    model, sd = build(cfg, 5, 17.0)
    audio, lens = synthetic_batch(2, 4.0, seed=8)
    buf = encode(model, audio, lens)
    got = device_beam(model, buf, 20, max_pops=640)
    want = og.espnet_beam(cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy(), beam=20, max_pops=640,
                          out_cap=2 * buf.tp_max + 16)
    assert got == [(w[0], float(np.float32(w[1])), w[2]) for w in want]
    print("120M beam-20 pops per frame:", [g[2] / max(1, int(t)) for g, t in zip(got, buf.enc_lens.cpu())], "labels", [len(g[0]) for g in got])


def test_nemo_family_two_lstm_layers_blank_last(gpu_device):
    """the same search over the NeMo-shaped decoder (`decoding.strategy: beam` is NeMo's default_beam_search, the same Graves
    algorithm): ReLU joint with a prediction bias, TWO LSTM layers, blank as the LAST index — against oracle/espnet_beam.c"""
    from reazonspeech_amd.runtime.config import TINY
    from reazonspeech_amd.runtime.weights import synthetic_state_dict
    from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
    from reazonspeech_amd.runtime.model import AsrModel
    cfg = TINY.with_(decoding="beam", beam_size=6)
    sd = synthetic_state_dict(cfg, 11, blank_bias=6.0)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    g = torch.Generator().manual_seed(3)
    B, Tp = 5, 30
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * 1.2
    lens = np.asarray([30, 17, 1, 0, 25], np.int32)
    dev = model.device
    cap = 2 * Tp + 16
    ids = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    frames = torch.full((B, cap), -7, dtype=torch.int32, device=dev)
    n_ids = torch.zeros((B,), dtype=torch.int32, device=dev)
    scores = torch.zeros((B,), dtype=torch.float32, device=dev)
    pops = torch.zeros((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((model.ctx.beam_workspace_bytes(B, 6, Tp, 600),), dtype=torch.uint8, device=dev)
    try:
        model.ctx.rnnt_beam(f.to(dev), torch.from_numpy(lens).to(dev), B, Tp, 6, True, 600, ids, n_ids, scores, pops, ws,
                            torch.cuda.current_stream().cuda_stream, frames=frames)
        torch.cuda.synchronize()
        got = [(ids[b, :int(n_ids[b])].cpu().tolist(), frames[b, :int(n_ids[b])].cpu().tolist(), float(scores[b]), int(pops[b]))
               for b in range(B)]
        want = og.espnet_beam(cfg, sd, f.numpy(), lens, beam=6, max_pops=600, out_cap=cap, with_frames=True)
    except RuntimeError as e:            # an untrained joint that never settles: the overflow must be the oracle's too
        assert "max_pops" in str(e)
        with pytest.raises(RuntimeError):
            og.espnet_beam(cfg, sd, f.numpy(), lens, beam=6, max_pops=600, out_cap=cap)
        pytest.skip("this synthetic decoder does not settle within max_pops (both sides agree)")
    assert got == [(w[0], w[1], float(np.float32(w[2])), w[3]) for w in want]
    assert sum(len(g_[0]) for g_ in got) > 0


def test_search_retries_the_decode_only_and_flags_a_degraded_batch(gpu_device):
    """`EspnetModel._search`: an overflowing beam search is retried on the SAME joint projection with a larger bound
    passed as a call argument (the shared model configuration is never touched: `transcribe_batch` workers may share the model) and,
    if that fails too, decoded greedily — the result then says so (`DecodedBatch.degraded`)"""
    import warnings
    model, sd = build(ESPNET_TINY, 11, 12.0, beam_size=6)
    audio, lens = synthetic_batch(3, 1.5, seed=4)
    waves = [audio[b, :int(lens[b])] for b in range(3)]
    cfg0 = model.am.cfg
    fine = model._search(waves)
    assert model.am.cfg is cfg0 and fine.degraded == [False] * 3 and fine.scores is not None
    # the smallest bound the library accepts (max_pops == beam): it may overflow at 1x, 4x and 16x — then the greedy fallback answers,
    # flagged and with a warning — or a retry succeeds and the beam result stands
    model.am.cfg = cfg0.with_(beam_max_pops=6)
    tight = model.am.cfg
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        res = model._search(waves)
    assert model.am.cfg is tight, "the retry policy must not write into the shared configuration"
    greedy = model.am.transcribe_waveforms.__self__           # (same object: just to keep the model alive in this scope)
    assert greedy is model.am
    if res.degraded == [True] * 3:
        assert any("greedy search instead" in str(x.message) for x in w)
        model.am.cfg = cfg0.with_(decoding="greedy_batch")
        want = model.am.transcribe_waveforms(waves)
        assert res.ids == want.ids and res.frames == want.frames
    else:                                                     # 16 evaluations were enough for this toy model: then the beam result stands
        assert res.degraded == [False] * 3 and res.ids == fine.ids
    model.am.cfg = cfg0
