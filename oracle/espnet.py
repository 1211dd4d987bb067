"""CPU restatement of the ESPnet2 Conformer-Transducer forward pass of `reazonspeech.espnet.asr` (front-end, encoder, CTC
posteriors, joint projection, greedy search).  TEST INFRASTRUCTURE — see oracle/__init__.py.

The reference obtains the model with `Speech2Text.from_pretrained(...)` (pkg/espnet-asr/src/transcribe.py:26-32) and calls
`model(np.pad(samples, (16000, 8000)))` (:69), `model.asr_model.encode` and `model.asr_model.ctc.softmax` (ctc.py:12-27).
ESPnet is neither vendored under /root/reference nor installed here and its checkpoint cannot be downloaded: PARITY
UNPINNED against ESPnet itself.  Each function cites the [UPSTREAM] ESPnet2 module it restates; the conformer block is the
same arithmetic as NeMo's (NeMo's was derived from ESPnet's) and reuses oracle/model.py, which IS pinned to an independent
implementation (transformers.models.parakeet) — with ESPnet's LayerNorm eps (1e-12), kernel size and head count.
"""
import math

import torch
import torch.nn.functional as F

from . import model as om


def frontend(cfg, sd, audio: torch.Tensor, lengths: torch.Tensor):
    """audio f32[B, Lmax], lengths i64[B] -> (features f32[B, Tmax, n_mels], n_frames i64[B]).

    [UPSTREAM] espnet2 DefaultFrontend: Stft (torch.stft, n_fft, hop, win_length, periodic Hann, center=True, reflect
    padding, onesided) -> power -> LogMel (librosa Slaney filters `melmat` [n_freq, n_mels], log(clamp(x, 1e-10))) with frames
    past 1 + L // hop zeroed, then GlobalMVN ((x - mean) / std) with the padding zeroed again.  Each utterance is transformed
    on its own samples (reflect padding looks at the utterance's last samples, not at the batch padding)."""
    B = audio.shape[0]
    window = torch.hann_window(cfg.win_length, periodic=True, dtype=torch.float32)
    melmat = sd["frontend.logmel.melmat"].to(torch.float32)
    mean, std = sd["normalize.mean"].to(torch.float32), sd["normalize.std"].to(torch.float32).clamp_min(cfg.norm_eps)
    n = 1 + lengths // cfg.hop_length
    Tmax = int(n.max())
    out = torch.zeros((B, Tmax, cfg.n_mels), dtype=torch.float32)
    for b in range(B):
        x = audio[b, :int(lengths[b])].to(torch.float32)
        st = torch.stft(x, cfg.n_fft, hop_length=cfg.hop_length, win_length=cfg.win_length, window=window, center=True,
                        pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        power = st.real ** 2 + st.imag ** 2                          # [n_freq, frames]
        logmel = torch.log(torch.clamp(power.t() @ melmat, min=cfg.log_guard))
        out[b, :int(n[b])] = (logmel - mean) / std
    return out, n


def subsampling(cfg, sd, feats, n_frames, recipe="fp32", taps=None):
    """[UPSTREAM] Conv2dSubsampling (espnet/nets/pytorch_backend/transformer/subsampling.py): Conv2d(1, C, 3, 2) ReLU
    Conv2d(C, C, 3, 2) ReLU, Linear(C * F2, d) on the (c, f)-flattened map, then RelPositionalEncoding's x * sqrt(d).
    Lengths follow the mask slicing `[:, :, :-2:2]` twice: (T - 1) // 2 each time.  bf16 recipe rounding points: the first
    conv's output, the second conv's weights and output, the Linear's weights."""
    E = "encoder.embed."
    rb = lambda t: om._rb(t, recipe)        # noqa: E731
    h = feats.unsqueeze(1)
    h = rb(F.relu(F.conv2d(h, sd[E + "conv.0.weight"], sd[E + "conv.0.bias"], stride=2)))
    h = rb(F.relu(F.conv2d(h, rb(sd[E + "conv.2.weight"]), sd[E + "conv.2.bias"], stride=2)))
    lens = n_frames.clone()
    for _ in range(2):
        lens = torch.where(lens >= 3, (lens - 3) // 2 + 1, torch.zeros_like(lens))
    B, C, Tp, Fq = h.shape
    if taps is not None:
        taps["sub_conv_out"] = h.clone()
    flat = h.transpose(1, 2).reshape(B, Tp, C * Fq)
    x = flat @ rb(sd[E + "out.0.weight"]).t() + sd[E + "out.0.bias"]
    if cfg.xscaling:
        x = x * math.sqrt(cfg.d_model)
    return x, lens


def nemo_keys(cfg, sd):
    """the conformer blocks under the key names oracle/model.py reads (feed_forward_macaron -> feed_forward1, ...)"""
    out = {}
    for i in range(cfg.n_layers):
        S, D = f"encoder.encoders.{i}.", f"encoder.layers.{i}."
        for a, b in (("feed_forward_macaron.w_1", "feed_forward1.linear1"), ("feed_forward_macaron.w_2", "feed_forward1.linear2"),
                     ("feed_forward.w_1", "feed_forward2.linear1"), ("feed_forward.w_2", "feed_forward2.linear2"),
                     ("norm_ff_macaron", "norm_feed_forward1"), ("norm_mha", "norm_self_att"), ("norm_conv", "norm_conv"),
                     ("norm_ff", "norm_feed_forward2"), ("norm_final", "norm_out"),
                     ("self_attn.linear_q", "self_attn.linear_q"), ("self_attn.linear_k", "self_attn.linear_k"),
                     ("self_attn.linear_v", "self_attn.linear_v"), ("self_attn.linear_out", "self_attn.linear_out"),
                     ("conv_module.pointwise_conv1", "conv.pointwise_conv1"), ("conv_module.depthwise_conv", "conv.depthwise_conv"),
                     ("conv_module.pointwise_conv2", "conv.pointwise_conv2")):
            for p in ("weight", "bias"):
                out[D + b + "." + p] = sd[S + a + "." + p]
        out[D + "self_attn.linear_pos.weight"] = sd[S + "self_attn.linear_pos.weight"]
        out[D + "self_attn.pos_bias_u"] = sd[S + "self_attn.pos_bias_u"]
        out[D + "self_attn.pos_bias_v"] = sd[S + "self_attn.pos_bias_v"]
        for p in ("weight", "bias", "running_mean", "running_var"):
            out[D + "conv.batch_norm." + p] = sd[S + "conv_module.norm." + p]
    return out


def encoder(cfg, sd, feats, n_frames, recipe="fp32", taps=None):
    """[UPSTREAM] espnet2 ConformerEncoder.forward: embed -> blocks (macaron FFN, rel-pos MHSA, conv module, FFN, norm_final)
    -> after_norm.  -> (enc f32[B, T', d], T'_b)"""
    x, lens = subsampling(cfg, sd, feats, n_frames, recipe, taps)
    if taps is not None:
        taps["sub_out"] = x.clone()
    sdn = nemo_keys(cfg, sd)
    pos_tab = om.rel_pos_table(cfg, x.shape[1])
    for i in range(cfg.n_layers):
        x = om.conformer_layer(cfg, sdn, i, x, pos_tab, lens, recipe)
        if taps is not None:
            taps[f"layer{i}"] = x.clone()
    x = om._ln(x, sd["encoder.after_norm.weight"], sd["encoder.after_norm.bias"], cfg.ln_eps)
    return x, lens


def forward(cfg, sd, audio, lengths, recipe="fp32", taps=None):
    """audio -> dict(enc, enc_lens, joint_enc = lin_enc(enc), ctc = softmax(ctc_lo(enc)))
    ([UPSTREAM] ESPnetASRModel.encode; CTC.softmax = F.softmax(ctc_lo(hs_pad), dim=2) — probabilities, not logs, which is
    what the reference's ctc.py feeds to ctc_segmentation: pkg/espnet-asr/src/ctc.py:25-27)."""
    with torch.no_grad():
        feats, n = frontend(cfg, sd, audio, lengths)
        if taps is not None:
            taps["feats"], taps["n_frames"] = feats.clone(), n.clone()
        enc, lens = encoder(cfg, sd, feats, n, recipe, taps)
        e = om._rb(enc, recipe)
        f = e @ om._rb(sd["joint_network.lin_enc.weight"], recipe).t() + sd["joint_network.lin_enc.bias"]
        ctc = torch.softmax(e @ om._rb(sd["ctc.ctc_lo.weight"], recipe).t() + sd["ctc.ctc_lo.bias"], dim=-1)
    return {"enc": enc, "enc_lens": lens, "joint_enc": f, "ctc": ctc}


def greedy_torch(cfg, sd, f: torch.Tensor, lens: torch.Tensor):
    """[UPSTREAM] espnet2 BeamSearchTransducer.greedy_search: one pass over the frames, at most ONE symbol per frame
    (`for enc_out_t in enc_out: logp = log_softmax(joint(enc_out_t, dec_out)); if argmax != blank: append, decoder.score`),
    joint = lin_out(tanh(lin_enc(h_enc) + lin_dec(h_dec))).  `f` is lin_enc(enc) (hoisted out of the loop).
    -> [(ids, frames)] per utterance; the bit-exact checker of the HIP decode is oracle/rnnt_greedy.c with act = tanh."""
    H = cfg.pred_hidden
    lstms = []
    for l in range(cfg.pred_layers):
        m = torch.nn.LSTM(H, H, 1, batch_first=True)
        with torch.no_grad():
            for nm in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                getattr(m, nm).copy_(sd[f"decoder.decoder.{l}." + nm])
        lstms.append(m)
    emb = sd["decoder.embed.weight"]
    wd = sd["joint_network.lin_dec.weight"]
    wo, bo = sd["joint_network.lin_out.weight"], sd["joint_network.lin_out.bias"]
    out = []
    with torch.no_grad():
        for b in range(f.shape[0]):
            states = [(torch.zeros(1, 1, H), torch.zeros(1, 1, H)) for _ in lstms]

            def step(tok):
                x = emb[tok].view(1, 1, H)
                for l, m in enumerate(lstms):
                    x, states[l] = m(x, states[l])
                return x[0, 0] @ wd.t()
            g = step(cfg.blank_id)
            ids, frames = [], []
            for t in range(int(lens[b])):
                k = int(torch.argmax(torch.tanh(f[b, t] + g) @ wo.t() + bo))
                if k != cfg.blank_id:
                    ids.append(k)
                    frames.append(t)
                    g = step(k)
            out.append((ids, frames))
    return out


def default_beam_search_torch(cfg, sd, f: torch.Tensor, lens: torch.Tensor, beam_size=20, score_norm=True, with_frames=False):
    """[UPSTREAM] espnet2 BeamSearchTransducer.default_beam_search + sort_nbest (espnet 202308, no LM) — the decode
    reazonspeech.espnet.asr runs, since the reference builds Speech2Text with its defaults (beam_size 20, search_type
    "default", score_norm True, nbest 1: pkg/espnet-asr/src/transcribe.py:27-31).  Restated statement for statement:

        kept_hyps = [Hypothesis(0.0, [blank], init_state)]
        for enc_out_t in enc_out:
            hyps, kept_hyps = kept_hyps, []
            while True:
                max_hyp = max(hyps, key=score); hyps.remove(max_hyp)
                dec_out, state = decoder.score(max_hyp)          # LSTM on yseq[-1] from max_hyp.dec_state
                logp = log_softmax(joint(enc_out_t, dec_out))
                top_k = logp[1:].topk(beam_k)
                kept_hyps.append(Hypothesis(max_hyp.score + float(logp[0]), max_hyp.yseq, max_hyp.dec_state))
                for logp_k, k in zip(*top_k):
                    hyps.append(Hypothesis(max_hyp.score + float(logp_k), max_hyp.yseq + [k + 1], state))
                hyps_max = max(hyps, key=score).score
                kept_most_prob = sorted([h for h in kept_hyps if h.score > hyps_max], key=score)
                if len(kept_most_prob) >= beam: kept_hyps = kept_most_prob; break
        return sorted(kept_hyps, key=score / len(yseq), reverse=True)[0]

    Scores are Python floats (float64 sums of float32 log-probabilities) as upstream; decoder.score's cache only saves work
    (the state is a function of the label sequence).  -> [(ids, score, pops)] per utterance.  The bit-exact checker of the HIP
    search is oracle/espnet_beam.c (float32 sums, fixed order); tests/test_oracle_espnet_beam.py compares the two."""
    H, V, blank = cfg.pred_hidden, cfg.n_logits, cfg.blank_id
    nemo = not getattr(cfg, "espnet", False)
    # the same search over a NeMo-shaped decoder ([UPSTREAM] BeamRNNTInfer.default_beam_search, `decoding.strategy: beam`): blank
    # is the LAST index there (top-k over logp[ids], ids = every index but blank), the joint is ReLU with a prediction bias
    lstms = []
    for l in range(cfg.pred_layers):
        m = torch.nn.LSTM(H, H, 1, batch_first=True)
        with torch.no_grad():
            for nm in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                src = sd[f"decoder.prediction.dec_rnn.lstm.{nm}_l{l}"] if nemo else sd[f"decoder.decoder.{l}.{nm}_l0"]
                getattr(m, nm + "_l0").copy_(src)
        lstms.append(m)
    if nemo:
        emb = sd["decoder.prediction.embed.weight"]
        wd, bd = sd["joint.pred.weight"], sd["joint.pred.bias"]
        wo, bo = sd["joint.joint_net.2.weight"], sd["joint.joint_net.2.bias"]
        act = torch.relu
    else:
        emb = sd["decoder.embed.weight"]
        wd, bd = sd["joint_network.lin_dec.weight"], 0.0
        wo, bo = sd["joint_network.lin_out.weight"], sd["joint_network.lin_out.bias"]
        act = torch.tanh
    labels = torch.tensor([v for v in range(V) if v != blank])
    beam = min(beam_size, V)
    beam_k = min(beam, V - 1)

    def score_fn(hyp):
        x = emb[hyp["yseq"][-1]].view(1, 1, H)
        new = []
        for l, m in enumerate(lstms):
            x, st = m(x, hyp["state"][l])
            new.append(st)
        return x[0, 0] @ wd.t() + bd, new

    out = []
    with torch.no_grad():
        for b in range(f.shape[0]):
            init = [(torch.zeros(1, 1, H), torch.zeros(1, 1, H)) for _ in lstms]
            kept = [dict(score=0.0, yseq=[blank], state=init, frames=[])]   # frames: [UPSTREAM] NeMo Hypothesis.timestep (ESPnet keeps none)
            pops = 0
            for t in range(int(lens[b])):
                hyps, kept = kept, []
                while True:
                    max_hyp = max(hyps, key=lambda h: h["score"])
                    hyps.remove(max_hyp)
                    pops += 1
                    dec_out, state = score_fn(max_hyp)
                    logp = torch.log_softmax(act(f[b, t] + dec_out) @ wo.t() + bo, dim=-1)
                    top = logp[labels].topk(beam_k)                  # (upstream: logp[1:] with blank = 0)
                    kept.append(dict(score=max_hyp["score"] + float(logp[blank]), yseq=max_hyp["yseq"][:], state=max_hyp["state"], frames=max_hyp["frames"]))
                    for lp, k in zip(*top):
                        hyps.append(dict(score=max_hyp["score"] + float(lp), yseq=max_hyp["yseq"][:] + [int(labels[k])], state=state, frames=max_hyp["frames"] + [t]))
                    hyps_max = float(max(hyps, key=lambda h: h["score"])["score"])
                    most = sorted([h for h in kept if h["score"] > hyps_max], key=lambda h: h["score"])
                    if len(most) >= beam:
                        kept = most
                        break
            if score_norm:
                kept = sorted(kept, key=lambda h: h["score"] / len(h["yseq"]), reverse=True)
            else:
                kept = sorted(kept, key=lambda h: h["score"], reverse=True)
            out.append((kept[0]["yseq"][1:], kept[0]["frames"], kept[0]["score"], pops) if with_frames else (kept[0]["yseq"][1:], kept[0]["score"], pops))
    return out
