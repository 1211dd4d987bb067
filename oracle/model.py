"""CPU restatement of the FastConformer-RNNT forward pass (front-end + encoder + joint
encoder projection).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Each function cites the upstream module it restates ([UPSTREAM] = NeMo, not in
/root/reference) and the line of the independent in-image implementation it was checked
against (HF = transformers/models/parakeet/...).

`recipe`:
  "fp32"  plain float32 everywhere (what NeMo does on CPU)
  "bf16"  same math, but tensors are rounded to bfloat16 exactly where the HIP path
          stores / feeds bf16 (GEMM operands, stored activations); accumulation,
          LayerNorm statistics, softmax, residual stream stay float32.
  "bf16-fused-glu"  the bf16 recipe of the batches the big-tile GEMM serves (M >= 1024 rows and at least two
          chip-fulls of tiles): the conv module's GLU is applied to the float32 accumulators in the pw1 GEMM
          epilogue and ITS output is what gets stored as bf16 (the plain "bf16" recipe rounds the pw1 output and
          applies the GLU in float32 inside the depthwise kernel).
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _rb(x: torch.Tensor, recipe: str) -> torch.Tensor:
    """round to bf16 and back (identity in the fp32 recipe)"""
    if recipe in ("bf16", "bf16-fused-glu"):
        return x.to(torch.bfloat16).to(torch.float32)
    return x


# --------------------------------------------------------------------------------------
# F1-F4: log-mel front-end
# --------------------------------------------------------------------------------------

def frontend(cfg, sd, audio: torch.Tensor, lengths: torch.Tensor):
    """audio f32[B, Lmax] (already zero padded like pad_audio), lengths i64[B]
    -> (features f32[B, Tmax, n_mels], n_frames i64[B]).

    [UPSTREAM] AudioToMelSpectrogramPreprocessor / FilterbankFeatures.forward:
    pre-emphasis (HF feature_extraction_parakeet.py:252-260), torch.stft(n_fft, hop,
    win, hann(periodic=False), center=True, pad_mode='constant') (HF:101-111), power
    (HF:114-117), mel filterbank matmul + log(x + 2^-24) (HF:120-122), per-feature
    normalisation over valid frames with unbiased variance, std + 1e-5, zero the padding
    (HF:263-276).  Valid frames = floor(L / hop) (HF:263-265).
    """
    B, Lmax = audio.shape
    t = torch.arange(Lmax)[None, :]
    valid = t < lengths[:, None]
    x = audio.to(torch.float32)
    if cfg.preemph:
        x = torch.cat([x[:, :1], x[:, 1:] - cfg.preemph * x[:, :-1]], dim=1)
    x = x.masked_fill(~valid, 0.0)
    window = sd["preprocessor.featurizer.window"].to(torch.float32)
    stft = torch.stft(x, cfg.n_fft, hop_length=cfg.hop_length, win_length=cfg.win_length,
                      window=window, center=True, pad_mode="constant", return_complex=True)
    power = stft.real ** 2 + stft.imag ** 2                      # [B, 257, frames]
    fb = sd["preprocessor.featurizer.fb"].to(torch.float32).reshape(cfg.n_mels, -1)
    mel = torch.log(fb @ power + cfg.log_guard).permute(0, 2, 1)  # [B, frames, n_mels]
    n = (lengths + (cfg.n_fft // 2) * 2 - cfg.n_fft) // cfg.hop_length
    Tmax = int(n.max())
    mel = mel[:, :Tmax]
    mask = (torch.arange(Tmax)[None, :] < n[:, None]).unsqueeze(-1)
    mm = mel * mask
    mean = mm.sum(1, keepdim=True) / n.view(-1, 1, 1)
    var = (((mm - mean) ** 2) * mask).sum(1, keepdim=True) / (n - 1).view(-1, 1, 1)
    out = (mel - mean) / (var.sqrt() + cfg.norm_eps)
    return out * mask, n


# --------------------------------------------------------------------------------------
# S1-S5: subsampling
# --------------------------------------------------------------------------------------

def _len_mask(lengths, T):
    return (torch.arange(T)[None, :] < lengths[:, None])


def subsampling(cfg, sd, feats: torch.Tensor, n_frames: torch.Tensor, recipe="fp32",
                taps: Optional[dict] = None):
    """feats f32[B, T, n_mels] -> (x f32[B, T', d_model], T'_b).

    [UPSTREAM] ConvSubsampling(dw_striding) with length masking after every conv
    (HF modeling_parakeet.py:377-431), flatten (C, F) and Linear, then xscaling
    (HF:562,607).  bf16 recipe rounding points: after the first depthwise conv (input of
    the pointwise GEMM), after each pointwise ReLU, after the 2nd depthwise conv.
    """
    pre = "encoder.pre_encode."
    h = feats.unsqueeze(1)                                   # [B,1,T,F]
    lens = n_frames.clone()
    lens = (lens + 2 - 3) // 2 + 1
    h = F.relu(F.conv2d(h, sd[pre + "conv.0.weight"], sd[pre + "conv.0.bias"], stride=2, padding=1))
    h = h * _len_mask(lens, h.shape[2])[:, None, :, None]
    idx = 2
    for _ in range(cfg.n_sub_stages - 1):
        c = h.shape[1]
        h = F.conv2d(h, sd[pre + f"conv.{idx}.weight"], sd[pre + f"conv.{idx}.bias"],
                     stride=2, padding=1, groups=c)
        lens = (lens + 2 - 3) // 2 + 1
        m = _len_mask(lens, h.shape[2])[:, None, :, None]
        h = _rb(h * m, recipe)
        w = _rb(sd[pre + f"conv.{idx + 1}.weight"], recipe)
        h = F.conv2d(h, w, sd[pre + f"conv.{idx + 1}.bias"])
        h = _rb(F.relu(h * m) * m, recipe)
        idx += 3
    if taps is not None:
        taps["sub_conv_out"] = h.clone()
    B, C, Tp, Fq = h.shape
    flat = h.transpose(1, 2).reshape(B, Tp, C * Fq)          # index c*F + f
    x = flat @ _rb(sd[pre + "out.weight"], recipe).t() + sd[pre + "out.bias"]
    if cfg.xscaling:
        x = x * math.sqrt(cfg.d_model)
    return x, lens


def rel_pos_table(cfg, T: int) -> torch.Tensor:
    """[2T-1, d] ([UPSTREAM] RelPositionalEncoding; HF:77-106)."""
    d = cfg.d_model
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)
    ang = pos[:, None] * inv_freq[None, :]
    return torch.stack([ang.sin(), ang.cos()], dim=-1).reshape(2 * T - 1, d)


# --------------------------------------------------------------------------------------
# L1-L7: conformer layer
# --------------------------------------------------------------------------------------

def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _linear(x, sd, name, recipe, bias=True):
    y = x @ _rb(sd[name + ".weight"], recipe).t()
    return y + sd[name + ".bias"] if bias else y


def feed_forward(cfg, sd, prefix, h, recipe):
    """[UPSTREAM] ConformerFeedForward: Linear -> SiLU -> Linear (HF:109-121)."""
    u = _rb(F.silu(_linear(h, sd, prefix + ".linear1", recipe)), recipe)
    return _linear(u, sd, prefix + ".linear2", recipe)


def attention_allowed(cfg, T, lens):
    """bool[B, T, T]: key j visible from query i.  Padding mask (both i and j must be
    valid, NeMo pad_mask_for_att) and the optional local window / global tokens
    (SURVEY.md §8a row L5)."""
    valid = _len_mask(lens, T)
    allowed = valid[:, :, None] & valid[:, None, :]
    if cfg.att_left >= 0 or cfg.att_right >= 0:
        i = torch.arange(T)[:, None]
        j = torch.arange(T)[None, :]
        left = cfg.att_left if cfg.att_left >= 0 else T
        right = cfg.att_right if cfg.att_right >= 0 else T
        win = ((i - j) <= left) & ((j - i) <= right)
        if cfg.n_global > 0:
            win = win | (i < cfg.n_global) | (j < cfg.n_global)
        allowed = allowed & win[None]
    return allowed


def attention_core(cfg, q, k, v, p, bias_u, bias_v, lens, recipe):
    """q,k,v f32[B,T,H,dh] (already rounded per recipe), p f32[2T-1,H,dh] -> ctx f32[B,T,H*dh].

    ac = (q+u) k^T, bd[i,j] = (q_i+v) . p[j-i+T-1]  (rel-shift closed form, SURVEY.md §10.3;
    HF:329-331,356-362), softmax((ac+bd)/sqrt(dh)) over visible keys; masked keys weigh 0 and
    padded / fully masked query rows output 0 (NeMo: -10000 fill, softmax, masked_fill(0))."""
    B, T, H, dh = q.shape
    qu = _rb(q + bias_u, recipe).transpose(1, 2)   # [B,H,T,dh]
    qv = _rb(q + bias_v, recipe).transpose(1, 2)
    kt = k.transpose(1, 2)
    vt = v.transpose(1, 2)
    ac = qu @ kt.transpose(-1, -2)                                      # [B,H,T,T]
    bd_full = qv @ p.permute(1, 2, 0)                                   # [B,H,T,2T-1]
    ii = torch.arange(T)[:, None]
    jj = torch.arange(T)[None, :]
    bd = bd_full.gather(-1, (jj - ii + T - 1).expand(B, H, T, T))
    s = (ac + bd) * (dh ** -0.5)
    allowed = attention_allowed(cfg, T, lens)[:, None]
    s = s.masked_fill(~allowed, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    e = torch.exp(s - m)
    e = e.masked_fill(~allowed, 0.0)
    den = e.sum(-1, keepdim=True)
    ctx = (_rb(e, recipe) @ vt) / torch.where(den > 0, den, torch.ones_like(den))
    ctx = ctx.masked_fill(den == 0, 0.0)
    return _rb(ctx.transpose(1, 2).reshape(B, T, H * dh), recipe)


def rel_pos_attention(cfg, sd, prefix, h, pos_tab, lens, recipe):
    """[UPSTREAM] RelPositionMultiHeadAttention (HF:299-362): q,k,v,pos projections,
    attention_core, linear_out.

    bf16 recipe: q,k,v,p stored bf16; (q+u),(q+v) rounded to bf16 (MFMA operands);
    probabilities rounded to bf16 for the PV product while the normaliser sums the
    unrounded values; context stored bf16."""
    B, T, d = h.shape
    H, dh = cfg.n_heads, cfg.head_dim
    q = _rb(_linear(h, sd, prefix + ".linear_q", recipe), recipe).view(B, T, H, dh)
    k = _rb(_linear(h, sd, prefix + ".linear_k", recipe), recipe).view(B, T, H, dh)
    v = _rb(_linear(h, sd, prefix + ".linear_v", recipe), recipe).view(B, T, H, dh)
    p = _rb(_linear(_rb(pos_tab, recipe), sd, prefix + ".linear_pos", recipe, bias=False), recipe)
    p = p.view(2 * T - 1, H, dh)
    ctx = attention_core(cfg, q, k, v, p, sd[prefix + ".pos_bias_u"], sd[prefix + ".pos_bias_v"], lens, recipe)
    return _linear(ctx, sd, prefix + ".linear_out", recipe)


def conv_module(cfg, sd, prefix, h, lens, recipe):
    """[UPSTREAM] ConformerConvolution (HF:159-193): pw1 -> GLU -> zero padded frames ->
    depthwise k -> BatchNorm (eval) -> SiLU -> pw2.  BatchNorm is folded into the
    depthwise weights exactly like the device weight prep does."""
    B, T, d = h.shape
    w1 = _rb(sd[prefix + ".pointwise_conv1.weight"].squeeze(-1), recipe)
    y = h @ w1.t() + sd[prefix + ".pointwise_conv1.bias"]  # [B,T,2d]
    if recipe != "bf16-fused-glu":
        y = _rb(y, recipe)                   # pw1 output stored bf16, GLU inside the conv kernel (f32, not rounded)
    a, g = y[..., :d], y[..., d:]
    u = a * torch.sigmoid(g)
    if recipe == "bf16-fused-glu":
        u = _rb(u, recipe)                   # GLU in the pw1 GEMM epilogue (f32 accumulators), its output stored bf16
    u = u * _len_mask(lens, T)[:, :, None]
    wdw, bdw = fold_batchnorm(cfg, sd, prefix)
    z = F.conv1d(u.transpose(1, 2), wdw[:, None, :], bdw, padding=(cfg.conv_kernel - 1) // 2,
                 groups=d).transpose(1, 2)
    z = _rb(F.silu(z), recipe)
    w2 = _rb(sd[prefix + ".pointwise_conv2.weight"].squeeze(-1), recipe)
    return z @ w2.t() + sd[prefix + ".pointwise_conv2.bias"]


def fold_batchnorm(cfg, sd, prefix):
    """dw'[c,k] = dw[c,k]*gamma/sqrt(var+eps); b' = (b - mean)*gamma/sqrt(var+eps) + beta
    (float64 on the host, stored float32)."""
    g = sd[prefix + ".batch_norm.weight"].double()
    b = sd[prefix + ".batch_norm.bias"].double()
    mu = sd[prefix + ".batch_norm.running_mean"].double()
    var = sd[prefix + ".batch_norm.running_var"].double()
    s = g / torch.sqrt(var + cfg.bn_eps)
    w = sd[prefix + ".depthwise_conv.weight"].double().squeeze(1) * s[:, None]
    bb = (sd[prefix + ".depthwise_conv.bias"].double() - mu) * s + b
    return w.float(), bb.float()


def conformer_layer(cfg, sd, i, x, pos_tab, lens, recipe):
    """[UPSTREAM] ConformerLayer.forward (HF:450-478)."""
    L = f"encoder.layers.{i}."
    eps = cfg.ln_eps

    def ln(name, t):
        return _rb(_ln(t, sd[L + name + ".weight"], sd[L + name + ".bias"], eps), recipe)

    x = x + 0.5 * feed_forward(cfg, sd, L + "feed_forward1", ln("norm_feed_forward1", x), recipe)
    x = x + rel_pos_attention(cfg, sd, L + "self_attn", ln("norm_self_att", x), pos_tab, lens, recipe)
    x = x + conv_module(cfg, sd, L + "conv", ln("norm_conv", x), lens, recipe)
    x = x + 0.5 * feed_forward(cfg, sd, L + "feed_forward2", ln("norm_feed_forward2", x), recipe)
    return _ln(x, sd[L + "norm_out.weight"], sd[L + "norm_out.bias"], eps)


def encoder(cfg, sd, feats, n_frames, recipe="fp32", taps: Optional[dict] = None):
    """feats f32[B,T,n_mels] -> (enc f32[B,T',d], T'_b)."""
    x, lens = subsampling(cfg, sd, feats, n_frames, recipe, taps)
    if taps is not None:
        taps["sub_out"] = x.clone()
    pos_tab = rel_pos_table(cfg, x.shape[1])
    for i in range(cfg.n_layers):
        x = conformer_layer(cfg, sd, i, x, pos_tab, lens, recipe)
        if taps is not None:
            taps[f"layer{i}"] = x.clone()
    return x, lens


def joint_enc_projection(cfg, sd, enc, recipe="fp32"):
    """[UPSTREAM] RNNTJoint.enc: Linear d_model -> joint_hidden on every frame (HF:930,949)."""
    return _linear(_rb(enc, recipe), sd, "joint.enc", recipe)


def forward_to_joint(cfg, sd, audio, lengths, recipe="fp32", taps=None):
    """audio -> (f f32[B,T',J], T'_b): everything the greedy loop consumes."""
    with torch.no_grad():
        feats, n = frontend(cfg, sd, audio, lengths)
        if taps is not None:
            taps["feats"] = feats.clone()
            taps["n_frames"] = n.clone()
        enc, lens = encoder(cfg, sd, feats, n, recipe, taps)
        f = joint_enc_projection(cfg, sd, enc, recipe)
        if taps is not None:
            taps["enc"] = enc.clone()
    return f, lens


# --------------------------------------------------------------------------------------
# D2-D4 in torch (used to pin rnnt_greedy.c; the C version is the bit-exact checker)
# --------------------------------------------------------------------------------------

def greedy_torch(cfg, sd, f: torch.Tensor, lens: torch.Tensor):
    """Per-utterance greedy RNN-T ([UPSTREAM] GreedyBatchedRNNTInfer, max_symbols;
    HF generation_parakeet.py:141-163).  Returns lists of (ids, frames) per utterance."""
    H = cfg.pred_hidden
    lstm = torch.nn.LSTM(H, H, cfg.pred_layers, batch_first=True)
    with torch.no_grad():
        for l in range(cfg.pred_layers):
            P = "decoder.prediction.dec_rnn.lstm."
            getattr(lstm, f"weight_ih_l{l}").copy_(sd[P + f"weight_ih_l{l}"])
            getattr(lstm, f"weight_hh_l{l}").copy_(sd[P + f"weight_hh_l{l}"])
            getattr(lstm, f"bias_ih_l{l}").copy_(sd[P + f"bias_ih_l{l}"])
            getattr(lstm, f"bias_hh_l{l}").copy_(sd[P + f"bias_hh_l{l}"])
    emb = sd["decoder.prediction.embed.weight"]
    Wp, bp = sd["joint.pred.weight"], sd["joint.pred.bias"]
    Wo, bo = sd["joint.joint_net.2.weight"], sd["joint.joint_net.2.bias"]
    out = []
    with torch.no_grad():
        for b in range(f.shape[0]):
            ids, frames = [], []
            state = (torch.zeros(cfg.pred_layers, 1, H), torch.zeros(cfg.pred_layers, 1, H))
            y, state = lstm(emb[cfg.blank_id].view(1, 1, H), state)
            g = y[0, 0] @ Wp.t() + bp
            t, sym = 0, 0
            while t < int(lens[b]):
                logits = torch.relu(f[b, t] + g) @ Wo.t() + bo
                k = int(torch.argmax(logits))
                if k == cfg.blank_id:
                    t += 1
                    sym = 0
                    continue
                ids.append(k)
                frames.append(t)
                y, state = lstm(emb[k].view(1, 1, H), state)
                g = y[0, 0] @ Wp.t() + bp
                sym += 1
                if sym >= cfg.max_symbols:
                    t += 1
                    sym = 0
            out.append((ids, frames))
    return out
