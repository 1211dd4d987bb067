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


# ---- the fixed-order C restatement (oracle/rnnt_alsd.c), the bit-exact checker of the HIP search ----------------
def _batch(n, secs, seed, blank_bias=0.0):
    sd = synthetic_state_dict(TINY, seed=0)
    sd["joint.joint_net.2.bias"][TINY.blank_id] += blank_bias
    audio, lens = synthetic_batch(n, secs, seed=seed)
    f, el = om.forward_to_joint(TINY, sd, torch.from_numpy(audio), torch.from_numpy(lens.astype(np.int64)), "fp32")
    return sd, f, el


def test_c_math_helpers():
    import math
    from oracle import greedy as og
    L = og.lib()
    rng = np.random.default_rng(3)
    for x in np.concatenate([rng.uniform(1e-5, 5000.0, 2000), [1.0, 2.0, 0.5, 3001.0]]).astype(np.float32):
        assert abs(L.rs_oracle_logf(float(x)) - math.log(float(x))) <= 2e-7 * max(1.0, abs(math.log(float(x))))
    z = rng.normal(0, 3, 3001).astype(np.float32)
    lse = L.rs_oracle_lse(z.ctypes.data_as(og.ctypes.POINTER(og.ctypes.c_float)), len(z))
    assert abs(lse - float(torch.logsumexp(torch.from_numpy(z).double(), 0))) < 2e-6
    assert abs(L.rs_oracle_logaddexpf(-3.5, -1.25) - np.logaddexp(-3.5, -1.25)) < 1e-6


@pytest.mark.parametrize("beam,mode,norm", [(1, "upstream", False), (2, "merge", True), (4, "upstream", True),
                                            (4, "merge", False)])
def test_c_alsd_matches_python_restatement(beam, mode, norm):
    """same decisions as oracle/alsd.py (float64 score bookkeeping there, float32 here): labels, alignment steps
    and score of the best hypothesis, ragged batch incl. an utterance of zero frames"""
    from oracle import greedy as og
    sd, f, el = _batch(3, 1.5, seed=5, blank_bias=0.0 if norm else -2.5)   # un-normalised scores favour short outputs
    el = el.clone()
    el[1] = max(1, int(el[1]) // 2)
    f = torch.cat([f, torch.zeros_like(f[:1])])
    el = torch.cat([el, torch.zeros(1, dtype=el.dtype)])
    got = og.rnnt_alsd(TINY, sd, f.numpy(), el.numpy(), beam=beam, max_target_len=1.0, score_norm=norm, recombine=mode)
    n_tok = 0
    for b in range(f.shape[0]):
        if int(el[b]) == 0:
            assert got[b] == ([], [], 0.0)
            continue
        want = alsd.alsd_decode(TINY, sd, f[b], int(el[b]), beam=beam, max_target_len=1.0, score_norm=norm,
                                recombine=mode)[0]
        assert got[b][0] == want.y_sequence[1:]
        assert got[b][1] == want.timestamp
        assert abs(got[b][2] - want.score) < 1e-3 * max(1.0, abs(want.score))
        n_tok += len(got[b][0])
    assert n_tok > 6, "degenerate fixture"


def test_c_alsd_beam1_is_c_greedy_without_symbol_cap():
    import dataclasses
    from oracle import greedy as og
    sd, f, el = _batch(2, 1.5, seed=9)
    cfg = dataclasses.replace(TINY, max_symbols=10 ** 6)
    want = og.rnnt_greedy(cfg, sd, f.numpy(), el.numpy(), u_max=4 * f.shape[1])
    got = og.rnnt_alsd(TINY, sd, f.numpy(), el.numpy(), beam=1, max_target_len=3 * int(f.shape[1]), score_norm=False)
    for (ids, frames), (aids, asteps, _) in zip(want, got):
        assert aids == ids and [i - u for u, i in enumerate(asteps)] == frames
