"""oracle/espnet_beam.c (the bit-exact float32 checker of the HIP search) against oracle/espnet.py::default_beam_search_torch
(the statement-for-statement restatement of ESPnet's BeamSearchTransducer.default_beam_search): same labels, scores within
float32 accumulation error, the same number of prediction-network evaluations.

The synthetic checkpoint is built with dec_gain = 8 (the prediction network weighs as much as the encoder in the joint) and its
own blank offset: with the encoder dominating, an untrained joint that prefers a label at some frame keeps preferring it after
emitting it, and the default search (which has no per-frame symbol limit upstream either) extends within that frame for ever;
max_pops bounds that and reports it."""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.config import ESPNET_TINY
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet
from reazonspeech_amd.runtime.synth import synthetic_batch
from oracle import espnet as oe, greedy as og


def _inputs(cfg, seed, B, seconds, blank_bias):
    sd = synthetic_state_dict_espnet(cfg, seed + 10, blank_bias=blank_bias, dec_gain=8.0)
    audio, lens = synthetic_batch(B, seconds, seed=seed)
    ref = oe.forward(cfg, sd, torch.from_numpy(audio), torch.from_numpy(lens), "fp32")
    return sd, ref["joint_enc"], ref["enc_lens"].to(torch.int32)


@pytest.mark.parametrize("beam,bias,seed", [(4, 12.0, 0), (20, 12.0, 1), (3, 12.0, 2), (20, 10.0, 0)])
def test_c_follows_torch_restatement(beam, bias, seed):
    cfg = ESPNET_TINY
    sd, f, lens = _inputs(cfg, seed, 2, 1.0, bias)
    ref = oe.default_beam_search_torch(cfg, sd, f, lens, beam_size=beam)
    got = og.espnet_beam(cfg, sd, f.numpy(), lens.numpy(), beam=beam, max_pops=64 * beam)
    for (rid, rs, rp), (gid, gs, gp) in zip(ref, got):
        assert gid == rid
        assert abs(gs - rs) <= 1e-4 * max(1.0, abs(rs))
        assert gp == rp
    assert any(len(r[0]) for r in ref)


def test_no_score_norm_and_empty_utterance():
    cfg = ESPNET_TINY
    sd, f, lens = _inputs(cfg, 7, 3, 1.0, 12.0)
    lens[1] = 0
    ref = oe.default_beam_search_torch(cfg, sd, f, lens, beam_size=5, score_norm=False)
    got = og.espnet_beam(cfg, sd, f.numpy(), lens.numpy(), beam=5, score_norm=False)
    assert [g[0] for g in got] == [r[0] for r in ref]
    assert got[1] == ([], 0.0, 0)


def test_overflow_is_reported():
    cfg = ESPNET_TINY
    sd, f, lens = _inputs(cfg, 7, 1, 0.5, 12.0)
    with pytest.raises(RuntimeError):
        og.espnet_beam(cfg, sd, f.numpy(), lens.numpy(), beam=8, max_pops=3)


def test_beam_one_is_not_greedy_but_close():
    """beam 1 of the default search may emit several labels per frame (greedy_search emits at most one): they agree whenever no
    frame wants two labels, which is the case for the blank-heavy synthetic checkpoint"""
    cfg = ESPNET_TINY
    sd, f, lens = _inputs(cfg, 11, 2, 1.0, 30.0)
    b1 = og.espnet_beam(cfg, sd, f.numpy(), lens.numpy(), beam=1)
    gr = og.rnnt_greedy(cfg, sd, f.numpy(), lens.numpy())
    assert [x[0] for x in b1] == [x[0] for x in gr]


def test_c_follows_torch_restatement_nemo_decoder():
    """the NeMo-shaped decoder (`decoding.strategy: beam`): two LSTM layers, blank as the last index, ReLU joint with a
    prediction bias — the C checker against the torch restatement on random joint-encoder projections"""
    from reazonspeech_amd.runtime.config import TINY
    from reazonspeech_amd.runtime.weights import synthetic_state_dict
    cfg = TINY
    sd = synthetic_state_dict(cfg, 11, blank_bias=6.0)
    g = torch.Generator().manual_seed(3)
    f = torch.randn((3, 14, cfg.joint_hidden), generator=g) * 1.2
    lens = torch.tensor([14, 9, 0], dtype=torch.int32)
    ref = oe.default_beam_search_torch(cfg, sd, f, lens, beam_size=5, with_frames=True)
    got = og.espnet_beam(cfg, sd, f.numpy(), lens.numpy(), beam=5, max_pops=400, with_frames=True)
    for (rid, rfr, rs, rp), (gid, gfr, gs, gp) in zip(ref, got):
        assert gid == rid and gp == rp
        assert gfr == rfr and all(a <= b_ for a, b_ in zip(gfr, gfr[1:]))      # the frame each label was appended at
        assert abs(gs - rs) <= 1e-4 * max(1.0, abs(rs))
    assert all(cfg.blank_id not in r[0] for r in ref) and any(r[1] for r in ref)
