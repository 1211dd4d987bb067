"""CPU: the ALSD restatement (oracle/alsd.py — parity unpinned upstream, see its header) against what CAN be
pinned here: with beam = 1, alignment-length synchronous search makes exactly the decisions of greedy decoding
without a per-frame symbol cap, for as many alignment steps as it is given."""
import numpy as np
import pytest
import torch

from oracle import alsd, model as om
from reazonspeech_amd.runtime.config import TINY
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.weights import synthetic_state_dict


@pytest.fixture(scope="module")
def fixture():
    sd = synthetic_state_dict(TINY, seed=0)
    audio, lens = synthetic_batch(1, 2.0, seed=11)
    wav = np.pad(audio[0, :lens[0]], 8000)
    f, el = om.forward_to_joint(TINY, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32")
    return sd, f[0], int(el[0])


def _greedy_trace(cfg, sd, f, t_len, n_steps):
    """independent greedy (own LSTM instance, no per-frame cap): the first n_steps blank/token decisions"""
    H = cfg.pred_hidden
    lstm = torch.nn.LSTM(H, H, cfg.pred_layers, batch_first=True)
    P = "decoder.prediction.dec_rnn.lstm."
    with torch.no_grad():
        for l in range(cfg.pred_layers):
            for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                getattr(lstm, f"{n}_l{l}").copy_(sd[P + f"{n}_l{l}"])
        emb = sd["decoder.prediction.embed.weight"]
        Wp, bp = sd["joint.pred.weight"], sd["joint.pred.bias"]
        Wo, bo = sd["joint.joint_net.2.weight"], sd["joint.joint_net.2.bias"]
        z = torch.zeros(cfg.pred_layers, 1, H)
        y, st = lstm(emb[cfg.blank_id].view(1, 1, H), (z, z.clone()))
        g = y[0, 0] @ Wp.t() + bp
        ids, frames, t = [], [], 0
        for _ in range(n_steps):
            if t >= t_len:
                break
            k = int(torch.argmax(torch.relu(f[t] + g) @ Wo.t() + bo))
            if k == cfg.blank_id:
                t += 1
                continue
            ids.append(k)
            frames.append(t)
            y, st = lstm(emb[k].view(1, 1, H), st)
            g = y[0, 0] @ Wp.t() + bp
    return ids, frames


def test_beam1_follows_greedy_decisions(fixture):
    sd, f, t_len = fixture
    u_max = t_len                         # T' + u_max alignment steps
    ids, frames = _greedy_trace(TINY, sd, f, t_len, t_len + u_max)
    assert len(ids) > 3, "degenerate fixture: greedy emitted almost nothing"
    best = alsd.alsd_decode(TINY, sd, f, t_len, beam=1, max_target_len=u_max, score_norm=False)[0]
    assert best.y_sequence[0] == TINY.blank_id                      # the artifact decode.py:40 trims
    assert best.y_sequence[1:] == ids
    assert best.frames() == frames
    assert len(best.timestamp) == len(ids) and best.timestamp == sorted(best.timestamp)


def test_nbest_shape_and_order(fixture):
    sd, f, t_len = fixture
    for mode in ("upstream", "merge"):
        nbest = alsd.alsd_decode(TINY, sd, f, t_len, beam=3, max_target_len=0.5, recombine=mode)
        assert len(nbest) >= 1
        norm = [h.score / len(h.y_sequence) for h in nbest]
        assert norm == sorted(norm, reverse=True)
        for h in nbest:
            assert h.y_sequence[0] == TINY.blank_id and len(h.timestamp) == len(h.y_sequence) - 1
            fr = h.frames()
            assert all(0 <= t < t_len for t in fr) and fr == sorted(fr)
            assert h.score <= 1e-6                                   # a sum of log-probabilities
