"""ALSD beam search — CPU restatement.  TEST INFRASTRUCTURE ONLY (groundwork for SURVEY.md §8f "next" row 2;
nothing in the product imports this).

PARITY UNPINNED.  The reference delegates decoding to NeMo (`model.transcribe`, pkg/nemo-asr/src/transcribe.py:48-53)
and only documents that the shipped checkpoint decodes with ALSD (pkg/nemo-asr/src/decode.py:29,38-41: "NeMo prepends
a blank token to y_sequence with ALSD").  NeMo is neither installed here nor vendored under /root/reference, and the
checkpoint's decoding config (beam size, alsd_max_target_len, score_norm) is not reachable, so this file restates
  * the published algorithm: Saon, Tüske, Audhkhasi, "Alignment-Length Synchronous Decoding for RNN Transducer",
    ICASSP 2020 (hypotheses are expanded in lockstep over the alignment length i = t + u; a hypothesis either takes
    blank (t+1) or one of its top-`beam` non-blank tokens (u+1); after each step the best `beam` are kept and
    hypotheses with equal label sequences are recombined with log-add), and
  * the structure of [UPSTREAM] nemo.collections.asr.parts.submodules.rnnt_beam_decoding.BeamRNNTInfer
    .align_length_sync_decoding as far as it is visible from the reference (leading blank in y_sequence,
    `timestamp` entries are the alignment index i of each emitted token).
Every behaviour that cannot be checked against upstream is a keyword argument with the value we believe upstream
uses as its default.  What IS pinned (tests/test_oracle_alsd.py): with beam = 1 the search is exactly greedy
decoding without a per-frame symbol cap, token for token and frame for frame.
"""
import dataclasses
import math
from typing import Dict, List, Optional, Tuple

import torch


@dataclasses.dataclass
class Hyp:
    y_sequence: List[int]                  # starts with the blank id (the artifact decode.py:40 trims)
    score: float
    timestamp: List[int]                   # alignment index i = t + u at which each non-blank token was emitted
    dec_state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None   # LSTM state BEFORE consuming y_sequence[-1]

    def frames(self) -> List[int]:
        """encoder frame of each emitted token: t = i - u"""
        return [i - u for u, i in enumerate(self.timestamp)]


class _PredNet:
    """embedding + LSTM + joint.pred projection, one step at a time, with the per-sequence cache upstream keeps"""

    def __init__(self, cfg, sd):
        H = cfg.pred_hidden
        self.cfg, self.H = cfg, H
        self.lstm = torch.nn.LSTM(H, H, cfg.pred_layers, batch_first=True)
        P = "decoder.prediction.dec_rnn.lstm."
        with torch.no_grad():
            for l in range(cfg.pred_layers):
                for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    getattr(self.lstm, f"{n}_l{l}").copy_(sd[P + f"{n}_l{l}"])
        self.emb = sd["decoder.prediction.embed.weight"]          # blank row is zero (padding_idx)
        self.Wp, self.bp = sd["joint.pred.weight"], sd["joint.pred.bias"]
        self.cache: Dict[Tuple[int, ...], Tuple[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]] = {}

    def zero_state(self):
        z = torch.zeros(self.cfg.pred_layers, 1, self.H)
        return (z, z.clone())

    def score(self, hyp: Hyp):
        """-> (g = joint.pred(pred_net output) for hyp, LSTM state after consuming hyp.y_sequence[-1])"""
        key = tuple(hyp.y_sequence)
        if key not in self.cache:
            with torch.no_grad():
                y, st = self.lstm(self.emb[hyp.y_sequence[-1]].view(1, 1, self.H), hyp.dec_state)
                self.cache[key] = (y[0, 0] @ self.Wp.t() + self.bp, st)
        return self.cache[key]


def alsd_decode(cfg, sd, f: torch.Tensor, t_len: int, beam: int = 4, max_target_len=2.0, score_norm: bool = True,
                recombine: str = "upstream", softmax_temperature: float = 1.0) -> List[Hyp]:
    """f: [T, joint_hidden] encoder frames already through joint.enc (what rs_encoder_forward emits) of ONE utterance.
    Returns the n-best list, best first.

    max_target_len: float = multiple of t_len, int = absolute (upstream `alsd_max_target_len`).
    recombine: "upstream" = scores of equal label sequences are log-added into the first occurrence and the list is
               returned with its duplicates (what we recall upstream's `recombine_hypotheses` returning);
               "merge" = duplicates are dropped (the paper)."""
    blank = cfg.blank_id
    V1 = sd["joint.joint_net.2.weight"].shape[0]
    Wo, bo = sd["joint.joint_net.2.weight"], sd["joint.joint_net.2.bias"]
    non_blank = torch.tensor([k for k in range(V1) if k != blank])
    beam = min(beam, V1 - 1)
    pn = _PredNet(cfg, sd)
    u_max = int(max_target_len * t_len) if isinstance(max_target_len, float) else int(max_target_len)
    B = [Hyp(y_sequence=[blank], score=0.0, timestamp=[], dec_state=pn.zero_state())]
    final: List[Hyp] = []
    for i in range(t_len + u_max):
        A: List[Hyp] = []
        live = [(h, i - (len(h.y_sequence) - 1)) for h in B]
        live = [(h, t) for h, t in live if t <= t_len - 1]
        if not live:
            break
        for hyp, t in live:
            g, new_state = pn.score(hyp)
            with torch.no_grad():
                logits = torch.relu(f[t] + g) @ Wo.t() + bo
                logp = torch.log_softmax(logits / softmax_temperature, dim=-1)
                top_v, top_i = logp[non_blank].topk(beam)
            blank_hyp = Hyp(hyp.y_sequence[:], hyp.score + float(logp[blank]), hyp.timestamp[:], hyp.dec_state)
            A.append(blank_hyp)
            if t == t_len - 1:
                final.append(blank_hyp)
            for lp, k in zip(top_v.tolist(), non_blank[top_i].tolist()):
                A.append(Hyp(hyp.y_sequence + [int(k)], hyp.score + float(lp), hyp.timestamp + [i], new_state))
        B = sorted(A, key=lambda h: h.score, reverse=True)[:beam]      # stable: ties keep expansion order
        B = _recombine(B, recombine)
    out = final if final else B
    if score_norm:
        return sorted(out, key=lambda h: h.score / len(h.y_sequence), reverse=True)
    return sorted(out, key=lambda h: h.score, reverse=True)


def _recombine(hyps: List[Hyp], mode: str) -> List[Hyp]:
    merged: List[Hyp] = []
    for h in hyps:
        for m in merged:
            if m.y_sequence == h.y_sequence:
                m.score = _logaddexp(m.score, h.score)
                break
        else:
            merged.append(h)
    return hyps if mode == "upstream" else merged


def _logaddexp(a: float, b: float) -> float:
    hi, lo = (a, b) if a >= b else (b, a)
    return hi + math.log1p(math.exp(lo - hi))
