"""Weights of the icefall Zipformer2 transducer behind `reazonspeech.k2.asr`: a seeded synthetic generator under icefall's
state-dict key names, the host-side re-layout into what librs_asr.so's rs_k2_* entry points consume, and a reader for the
three ONNX files + tokens.txt the reference hands to sherpa-onnx (pkg/k2-asr/src/huggingface.py:41-83) that needs neither
`onnx` nor `onnxruntime` (runtime/onnx_lite.py parses the protobuf wire format).

No checkpoint is reachable here (HF_HUB_OFFLINE, no cache): every key name is [UPSTREAM] icefall
(egs/librispeech/ASR/zipformer/{zipformer,subsampling,scaling,decoder,joiner}.py as exported by export-onnx.py).
"""
import math
from typing import Dict

import numpy as np
import torch

from .config import UnsupportedCheckpoint
from .k2_config import ZipformerConfig
from .weights import _randn, _seed_for, banded_filterbank, fft_twiddles, glu_interleave_index, screen_tensors, to_fragment_major

K2_POS_CAP = 1024      # relative positions kept resident per stack: frames of the 50 Hz stack up to 1024 (~20 s); grown on demand
BRANCH = 0.25          # gain of the residual branches' output projections (icefall's ScaledLinear initial_scale plays this role)


def _rand_scale(name, seed, d, meta):
    if meta:
        return torch.empty((d,), dtype=torch.float32, device="meta")
    g = torch.Generator().manual_seed(_seed_for(name, seed))
    return 0.35 + 0.3 * torch.rand((d,), generator=g)


def layer_prefix(cfg: ZipformerConfig, s: int, j: int) -> str:
    """icefall: a full-rate stack is a Zipformer2Encoder (`layers`), a down-sampled one wraps it (`encoder.layers`)"""
    return f"encoder.encoders.{s}." + ("" if cfg.downsampling[s] == 1 else "encoder.") + f"layers.{j}."


def expected_shapes_k2(cfg: ZipformerConfig) -> Dict[str, tuple]:
    """every key of an icefall Zipformer2 transducer state dict at this architecture -> its shape (the checkpoint readers check
    what they recovered against this list: runtime/k2_onnx.py)"""
    return {k: tuple(v.shape) for k, v in synthetic_state_dict_k2(cfg, meta=True).items()}


def synthetic_state_dict_k2(cfg: ZipformerConfig, seed: int = 0, blank_bias: float = None, meta: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded random weights with icefall's keys and shapes.  Linears are N(0, 1 / fan_in) with small gains on the residual
    branches; BiasNorm log-scales around 0; bypass scales around 0.5 (icefall's initial value); a blank-logit offset makes
    greedy search emit a realistic number of tokens.  meta = True: shapes only (tensors on the "meta" device)."""
    cfg.validate()
    sd: Dict[str, torch.Tensor] = {}
    if meta:
        def _randn(name, seed, shape, std):                     # noqa: F811 (shadows the generator: no storage, no random numbers)
            return torch.empty(shape, dtype=torch.float32, device="meta")
    else:
        from .weights import _randn

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = _randn(name + ".weight", seed, (out_f, in_f), gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = _randn(name + ".bias", seed, (out_f,), 0.05)

    def conv(name, cout, cin_g, *k, gain=1.0):
        fan = cin_g * int(np.prod(k))
        sd[name + ".weight"] = _randn(name + ".weight", seed, (cout, cin_g) + tuple(k), gain / math.sqrt(fan))
        sd[name + ".bias"] = _randn(name + ".bias", seed, (cout,), 0.05)

    def biasnorm(name, n):
        sd[name + ".log_scale"] = _randn(name + ".log_scale", seed, (), 0.1)
        sd[name + ".bias"] = _randn(name + ".bias", seed, (n,), 0.05)

    c1, c2, c3 = cfg.embed_channels
    E = "encoder_embed."
    conv(E + "conv.0", c1, 1, 3, 3, gain=0.5)            # features are log-mel energies around -10: keep the first conv small
    conv(E + "conv.4", c2, c1, 3, 3, gain=1.5)
    conv(E + "conv.7", c3, c2, 3, 3, gain=1.5)
    conv(E + "convnext.depthwise_conv", c3, 1, 7, 7)
    conv(E + "convnext.pointwise_conv1", 3 * c3, c3, 1, 1)
    conv(E + "convnext.pointwise_conv2", c3, 3 * c3, 1, 1, gain=BRANCH)
    lin(E + "out", cfg.encoder_dim[0], cfg.embed_freq * c3)
    biasnorm(E + "out_norm", cfg.encoder_dim[0])
    qd, pd, vd = cfg.query_head_dim, cfg.pos_head_dim, cfg.value_head_dim
    for s in range(cfg.n_stacks):
        d, h, k = cfg.encoder_dim[s], cfg.num_heads[s], cfg.cnn_kernel[s]
        hid = cfg.nonlin_hidden(s)
        for j in range(cfg.num_layers[s]):
            L = layer_prefix(cfg, s, j)
            lin(L + "self_attn_weights.in_proj", (2 * qd + pd) * h, d, gain=1.3)
            lin(L + "self_attn_weights.linear_pos", h * pd, cfg.pos_dim, bias=False, gain=1.0)
            for a in ("self_attn1", "self_attn2"):
                lin(L + a + ".in_proj", h * vd, d)
                lin(L + a + ".out_proj", d, h * vd, gain=BRANCH)
            for name, f in zip(("feed_forward1", "feed_forward2", "feed_forward3"), cfg.layer_ff(s)):
                lin(L + name + ".in_proj", f, d, gain=2.0)
                lin(L + name + ".out_proj", d, f, gain=BRANCH)
            lin(L + "nonlin_attention.in_proj", 3 * hid, d)
            lin(L + "nonlin_attention.out_proj", d, hid, gain=BRANCH)
            for cm in ("conv_module1", "conv_module2"):
                lin(L + cm + ".in_proj", 2 * d, d)
                conv(L + cm + ".depthwise_conv", d, 1, k)
                lin(L + cm + ".out_proj", d, d, gain=BRANCH)
            biasnorm(L + "norm", d)
            for b in ("bypass", "bypass_mid"):
                sd[L + b + ".bypass_scale"] = _rand_scale(L + b, seed, d, meta)
        if cfg.downsampling[s] > 1:
            P = f"encoder.encoders.{s}."
            sd[P + "downsample.bias"] = _randn(P + "downsample.bias", seed, (cfg.downsampling[s],), 0.3)
            sd[P + "out_combiner.bypass_scale"] = _rand_scale(P + "out_combiner", seed, d, meta)
    sd["encoder.downsample_output.bias"] = _randn("encoder.downsample_output.bias", seed, (cfg.output_downsampling,), 0.3)
    D, J, V = cfg.decoder_dim, cfg.joiner_dim, cfg.vocab_size
    sd["decoder.embedding.weight"] = _randn("decoder.embedding.weight", seed, (V, D), 1.0)
    sd["decoder.conv.weight"] = _randn("decoder.conv.weight", seed, (D, 4, cfg.context_size), 1.0 / math.sqrt(8.0))
    lin("joiner.encoder_proj", J, cfg.out_dim)
    lin("joiner.decoder_proj", J, D, gain=3.0)
    lin("joiner.output_linear", V, J, gain=6.0)
    if meta:
        return sd
    if blank_bias is None:
        # scanned with the CPU oracle (oracle/zipformer.py + oracle/k2_greedy.c): about 35 tokens per 10 s utterance (293 frames) at the 159M shape
        blank_bias = {(768, 10720): 11.0, (128, 97): 6.0}.get((cfg.out_dim, cfg.vocab_size), 10.0)
    sd["joiner.output_linear.bias"][cfg.blank_id] += float(blank_bias)
    return sd


# ------------------------------------------------------------------------------------------------------------------
# feature extraction constants (kaldi-native-fbank)
# ------------------------------------------------------------------------------------------------------------------

def povey_window(n: int) -> np.ndarray:
    """[UPSTREAM] knf FeatureWindowFunction("povey"): pow(0.5 - 0.5 cos(2 pi i / (N - 1)), 0.85), computed in double"""
    i = np.arange(n, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(2.0 * np.pi * i / (n - 1)), 0.85).astype(np.float32)


def kaldi_mel_banks(cfg: ZipformerConfig) -> np.ndarray:
    """[UPSTREAM] knf MelBanks: triangular filters in the mel domain (mel = 1127 ln(1 + f / 700)) between low_freq and
    high_freq (<= 0: Nyquist + high_freq) over the n_fft / 2 bins below the Nyquist bin.  -> float32 [n_mels][n_fft / 2 + 1]
    (the Nyquist column is zero)."""
    nyq = 0.5 * cfg.sample_rate
    hi = cfg.high_freq if cfg.high_freq > 0 else nyq + cfg.high_freq
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)        # noqa: E731
    n_bins = cfg.n_fft // 2
    width = cfg.sample_rate / cfg.n_fft
    m_lo, m_hi = mel(cfg.low_freq), mel(hi)
    delta = (m_hi - m_lo) / (cfg.n_mels + 1)
    fb = np.zeros((cfg.n_mels, n_bins + 1), np.float32)
    mels = mel(width * np.arange(n_bins, dtype=np.float64))
    for b in range(cfg.n_mels):
        left, center, right = m_lo + b * delta, m_lo + (b + 1) * delta, m_lo + (b + 2) * delta
        for i in range(n_bins):
            m = mels[i]
            if left < m < right:
                fb[b, i] = np.float32((m - left) / (center - left) if m <= center else (right - m) / (right - center))
    return fb


def compact_rel_pos_table(cfg: ZipformerConfig, cap: int) -> np.ndarray:
    """[UPSTREAM] CompactRelPositionalEncoding(embed_dim = pos_dim, length_factor = 1): row n <-> relative position n - (cap - 1)
    (key index minus query index).  float32 [2 * cap - 1][pos_dim], computed in float32 like torch does."""
    D = cfg.pos_dim
    x = torch.arange(-(cap - 1), cap, dtype=torch.float32).unsqueeze(1)
    freqs = 1 + torch.arange(D // 2, dtype=torch.float32)
    comp = D ** 0.5
    xc = comp * x.sign() * ((x.abs() + comp).log() - math.log(comp))
    length_scale = 1.0 * D / (2.0 * math.pi)
    xa = (xc / length_scale).atan()
    pe = torch.zeros((x.shape[0], D), dtype=torch.float32)
    pe[:, 0::2] = (xa * freqs).cos()
    pe[:, 1::2] = (xa * freqs).sin()
    pe[:, -1] = 1.0
    return pe.numpy()


def pad_cols(w: torch.Tensor, mult: int = 64) -> torch.Tensor:
    """zero-pad the K extent (columns) of a [N][K] weight to a multiple of `mult` (the GEMM's K tiles)"""
    n, k = w.shape
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros((n, kp), dtype=w.dtype)
    out[:, :k] = w
    return out


def prepare_weights_k2(cfg: ZipformerConfig, sd: Dict[str, torch.Tensor], pos_cap: int = K2_POS_CAP, f32: bool = False):
    """-> dict name -> CPU tensor as registered with rs_k2_set_tensor (include/rs_asr.h).  One-off host transforms:
      * GEMM weights bf16 [N][K] with K zero-padded to a multiple of 64 where the model's extent is not one (attention values
        H * 12, the 3/4-width non-linear attention of a 192-wide stack, the 3x3x32 patches of encoder_embed's third conv);
      * encoder_embed convs tap-major / channels-last; `out` columns permuted from (c, f) to (f, c);
      * conv modules' in_proj rows interleaved in blocks of 32 (values, gates) for the GLU epilogue;
      * BiasNorm: exp(log_scale) taken on the host; SimpleDownsample: softmax(bias) taken on the host;
      * relative positions: CompactRelPositionalEncoding rows for |rel| < pos_cap projected by every layer's linear_pos
        (float32 [2 * cap - 1][H * 4]) — the encoding depends on the relative position only;
      * decoder / joiner: float32, the joiner's two matrices fragment-major for the exact-f32 decode kernels;
      * f32 = True adds the float32 parity mode's dense weights as "<name>.f32" (unrounded; K zero-padded to a multiple of 32
        where the bf16 copy pads to 64; the conv modules' in_proj and its bias in icefall's own row order: values, then gates)."""
    cfg.validate()
    out, used = {}, set()
    bf = lambda t: t.detach().to(torch.float32).to(torch.bfloat16).contiguous()   # noqa: E731
    f32_ = lambda t: t.detach().to(torch.float32).contiguous()                     # noqa: E731

    def get(key):
        if key not in sd:
            raise UnsupportedCheckpoint(f"checkpoint has no tensor {key!r} (architecture differs from the configuration?)")
        used.add(key)
        return sd[key]

    def scale4(t):
        return torch.tensor([float(torch.exp(t.double())), 0.0, 0.0, 0.0], dtype=torch.float32)

    def soft8(t):
        w = torch.zeros((8,), dtype=torch.float32)
        w[:t.numel()] = torch.softmax(t.detach().to(torch.float32), dim=0)
        return w

    idx, w = banded_filterbank(kaldi_mel_banks(cfg))
    out["fe.window"] = torch.from_numpy(povey_window(cfg.frame_length))
    out["fe.twiddle"] = torch.from_numpy(fft_twiddles(cfg.n_fft))
    out["fe.fb_idx"] = torch.from_numpy(idx)
    out["fe.fb_w"] = torch.from_numpy(w)
    c1, c2, c3 = cfg.embed_channels
    E = "encoder_embed."
    out["emb.conv0.w"] = f32_(get(E + "conv.0.weight").reshape(c1, 9).t())                       # [9][c1]
    out["emb.conv0.b"] = f32_(get(E + "conv.0.bias"))
    out["emb.conv1.w"] = f32_(get(E + "conv.4.weight").permute(2, 3, 1, 0))                      # [3][3][c1][c2]
    out["emb.conv1.b"] = f32_(get(E + "conv.4.bias"))
    out["emb.conv2.w"] = bf(pad_cols(get(E + "conv.7.weight").permute(0, 2, 3, 1).reshape(c3, 9 * c2).float()))   # K = (kh, kw, cin)
    out["emb.conv2.b"] = f32_(get(E + "conv.7.bias"))
    out["emb.cnx.dw.w"] = f32_(get(E + "convnext.depthwise_conv.weight").reshape(c3, 49).t())    # [49][c3]
    out["emb.cnx.dw.b"] = f32_(get(E + "convnext.depthwise_conv.bias"))
    out["emb.cnx.pw1.w"] = bf(get(E + "convnext.pointwise_conv1.weight").reshape(3 * c3, c3))
    out["emb.cnx.pw1.b"] = f32_(get(E + "convnext.pointwise_conv1.bias"))
    out["emb.cnx.pw2.w"] = bf(get(E + "convnext.pointwise_conv2.weight").reshape(c3, 3 * c3))
    out["emb.cnx.pw2.b"] = f32_(get(E + "convnext.pointwise_conv2.bias"))
    F, d0 = cfg.embed_freq, cfg.encoder_dim[0]
    out["emb.out.w"] = bf(get(E + "out.weight").reshape(d0, c3, F).permute(0, 2, 1).reshape(d0, F * c3))
    out["emb.out.b"] = f32_(get(E + "out.bias"))
    out["emb.norm.bias"] = f32_(get(E + "out_norm.bias"))
    out["emb.norm.scale"] = scale4(get(E + "out_norm.log_scale"))
    pe = torch.from_numpy(compact_rel_pos_table(cfg, pos_cap))
    qd, pd = cfg.query_head_dim, cfg.pos_head_dim
    for s in range(cfg.n_stacks):
        d, h = cfg.encoder_dim[s], cfg.num_heads[s]
        for j in range(cfg.num_layers[s]):
            L, p = layer_prefix(cfg, s, j), f"S{s}.L{j}."
            out[p + "attw.in.w"] = bf(get(L + "self_attn_weights.in_proj.weight"))
            out[p + "attw.in.b"] = f32_(get(L + "self_attn_weights.in_proj.bias"))
            wp = get(L + "self_attn_weights.linear_pos.weight").to(torch.float32)              # [h * pd][pos_dim]
            out[p + "attw.pos_proj"] = (pe @ wp.t()).contiguous()                                 # [2 cap - 1][h * pd]
            for a, q in (("self_attn1", "sa1"), ("self_attn2", "sa2")):
                out[p + q + ".in.w"] = bf(get(L + a + ".in_proj.weight"))
                out[p + q + ".in.b"] = f32_(get(L + a + ".in_proj.bias"))
                out[p + q + ".out.w"] = bf(pad_cols(get(L + a + ".out_proj.weight").float()))
                out[p + q + ".out.b"] = f32_(get(L + a + ".out_proj.bias"))
            for n, q in (("feed_forward1", "ff1"), ("feed_forward2", "ff2"), ("feed_forward3", "ff3")):
                out[p + q + ".in.w"] = bf(get(L + n + ".in_proj.weight"))
                out[p + q + ".in.b"] = f32_(get(L + n + ".in_proj.bias"))
                out[p + q + ".out.w"] = bf(get(L + n + ".out_proj.weight"))
                out[p + q + ".out.b"] = f32_(get(L + n + ".out_proj.bias"))
            out[p + "na.in.w"] = bf(get(L + "nonlin_attention.in_proj.weight"))
            out[p + "na.in.b"] = f32_(get(L + "nonlin_attention.in_proj.bias"))
            out[p + "na.out.w"] = bf(pad_cols(get(L + "nonlin_attention.out_proj.weight").float()))
            out[p + "na.out.b"] = f32_(get(L + "nonlin_attention.out_proj.bias"))
            rows = glu_interleave_index(d)
            for cm, q in (("conv_module1", "cm1"), ("conv_module2", "cm2")):
                out[p + q + ".in.w"] = bf(get(L + cm + ".in_proj.weight")[rows])
                out[p + q + ".in.b"] = f32_(get(L + cm + ".in_proj.bias")[rows])
                out[p + q + ".dw.w"] = f32_(get(L + cm + ".depthwise_conv.weight").squeeze(1).t())   # [k][d]
                out[p + q + ".dw.b"] = f32_(get(L + cm + ".depthwise_conv.bias"))
                out[p + q + ".out.w"] = bf(get(L + cm + ".out_proj.weight"))
                out[p + q + ".out.b"] = f32_(get(L + cm + ".out_proj.bias"))
            out[p + "norm.bias"] = f32_(get(L + "norm.bias"))
            out[p + "norm.scale"] = scale4(get(L + "norm.log_scale"))
            out[p + "bypass.scale"] = f32_(get(L + "bypass.bypass_scale"))
            out[p + "bypass_mid.scale"] = f32_(get(L + "bypass_mid.bypass_scale"))
        if cfg.downsampling[s] > 1:
            P = f"encoder.encoders.{s}."
            out[f"S{s}.ds.w"] = soft8(get(P + "downsample.bias"))
            out[f"S{s}.comb.scale"] = f32_(get(P + "out_combiner.bypass_scale"))
    out["out.ds.w"] = soft8(get("encoder.downsample_output.bias"))
    out["joint.enc.w"] = bf(get("joiner.encoder_proj.weight"))
    out["joint.enc.b"] = f32_(get("joiner.encoder_proj.bias"))
    out["dec.embed"] = f32_(get("decoder.embedding.weight"))
    out["dec.conv.w"] = f32_(get("decoder.conv.weight"))                                   # [D][4][context]
    out["joint.pred.w"] = to_fragment_major(get("joiner.decoder_proj.weight"))
    out["joint.pred.b"] = f32_(get("joiner.decoder_proj.bias"))
    out["joint.out.w"] = to_fragment_major(get("joiner.output_linear.weight"))
    out["joint.out.b"] = f32_(get("joiner.output_linear.bias"))
    out.update(screen_tensors(get("joiner.output_linear.weight"), get("joiner.output_linear.bias")))      # screened joint (greedy search)
    if f32:
        p32 = lambda t: pad_cols(t.detach().to(torch.float32), 32)        # noqa: E731
        out["emb.conv2.w.f32"] = p32(sd[E + "conv.7.weight"].permute(0, 2, 3, 1).reshape(c3, 9 * c2))
        out["emb.cnx.pw1.w.f32"] = f32_(sd[E + "convnext.pointwise_conv1.weight"].reshape(3 * c3, c3))
        out["emb.cnx.pw2.w.f32"] = f32_(sd[E + "convnext.pointwise_conv2.weight"].reshape(c3, 3 * c3))
        out["emb.out.w.f32"] = f32_(sd[E + "out.weight"].reshape(d0, c3, F).permute(0, 2, 1).reshape(d0, F * c3))
        out["joint.enc.w.f32"] = f32_(sd["joiner.encoder_proj.weight"])
        for s in range(cfg.n_stacks):
            for j in range(cfg.num_layers[s]):
                L, p = layer_prefix(cfg, s, j), f"S{s}.L{j}."
                out[p + "attw.in.w.f32"] = f32_(sd[L + "self_attn_weights.in_proj.weight"])
                for a, q in (("self_attn1", "sa1"), ("self_attn2", "sa2")):
                    out[p + q + ".in.w.f32"] = f32_(sd[L + a + ".in_proj.weight"])
                    out[p + q + ".out.w.f32"] = p32(sd[L + a + ".out_proj.weight"])
                for n, q in (("feed_forward1", "ff1"), ("feed_forward2", "ff2"), ("feed_forward3", "ff3")):
                    out[p + q + ".in.w.f32"] = f32_(sd[L + n + ".in_proj.weight"])
                    out[p + q + ".out.w.f32"] = f32_(sd[L + n + ".out_proj.weight"])
                out[p + "na.in.w.f32"] = f32_(sd[L + "nonlin_attention.in_proj.weight"])
                out[p + "na.out.w.f32"] = p32(sd[L + "nonlin_attention.out_proj.weight"])
                for cm, q in (("conv_module1", "cm1"), ("conv_module2", "cm2")):
                    out[p + q + ".in.w.f32"] = f32_(sd[L + cm + ".in_proj.weight"])
                    out[p + q + ".in.b.f32"] = f32_(sd[L + cm + ".in_proj.bias"])
                    out[p + q + ".out.w.f32"] = f32_(sd[L + cm + ".out_proj.weight"])
    left = [k for k in sd if k not in used]
    if left:
        raise UnsupportedCheckpoint(f"{len(left)} checkpoint tensor(s) have no counterpart in this implementation: "
                                    + ", ".join(left[:8]) + (" ..." if len(left) > 8 else ""))
    return out
