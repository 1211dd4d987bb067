"""Weights of the ESPnet2 Conformer-Transducer (`reazonspeech.espnet.asr`): seeded synthetic generator with ESPnet's
state-dict key names, and the host-side re-layout into what librs_asr.so consumes.

The reference loads `Speech2Text.from_pretrained("https://huggingface.co/reazon-research/reazonspeech-espnet-v2", lm_weight=0)`
(pkg/espnet-asr/src/transcribe.py:26-32): an ESPnet2 `ESPnetASRModel` (config.yaml + a .pth state dict).  Neither ESPnet nor
the checkpoint is reachable here, so every key name below is [UPSTREAM] ESPnet2 (espnet2/asr/encoder/conformer_encoder.py,
espnet/nets/pytorch_backend/conformer/{encoder_layer,convolution}.py, .../transformer/{subsampling,attention,embedding}.py,
espnet2/asr/decoder/transducer_decoder.py, espnet2/asr/transducer/joint_network.py, espnet2/asr/ctc.py,
espnet2/layers/{stft,log_mel,global_mvn}.py) as of ESPnet 202x; `prepare_weights_espnet` refuses a state dict with leftovers.
"""
import math
from typing import Dict

import numpy as np
import torch

from .config import ModelConfig, UnsupportedCheckpoint
from .weights import (BRANCH_GAIN, _randn, _seed_for, banded_filterbank, fft_twiddles, glu_interleave_index, rel_pos_table,
                      screen_tensors, slaney_mel_filterbank, to_fragment_major, DEFAULT_POS_CAP)


def synthetic_state_dict_espnet(cfg: ModelConfig, seed: int = 0, blank_bias: float = None, dec_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights under ESPnet2's keys and shapes (the recipe of `synthetic_state_dict`: 1/sqrt(fan_in) linears,
    small residual-branch gains, a blank-logit offset so greedy emits a realistic number of tokens)."""
    assert cfg.espnet
    cfg.validate()
    sd: Dict[str, torch.Tensor] = {}
    d, f, H, J, V, k = cfg.d_model, cfg.ff_dim, cfg.pred_hidden, cfg.joint_hidden, cfg.vocab_size, cfg.conv_kernel

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = _randn(name + ".weight", seed, (out_f, in_f), gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = _randn(name + ".bias", seed, (out_f,), 0.05)

    def norm(name, n):
        sd[name + ".weight"] = 1.0 + _randn(name + ".weight", seed, (n,), 0.05)
        sd[name + ".bias"] = _randn(name + ".bias", seed, (n,), 0.05)

    # front-end buffers: LogMel.melmat [n_freq, n_mels] (librosa Slaney filters, transposed), GlobalMVN mean / std
    sd["frontend.logmel.melmat"] = torch.from_numpy(slaney_mel_filterbank(cfg)).t().contiguous()
    sd["normalize.mean"] = -9.0 + _randn("normalize.mean", seed, (cfg.n_mels,), 1.0)       # log-mel of speech sits around -9
    sd["normalize.std"] = 3.0 + _randn("normalize.std", seed, (cfg.n_mels,), 0.2).abs()

    E = "encoder.embed."
    sd[E + "conv.0.weight"] = _randn(E + "conv.0.weight", seed, (d, 1, 3, 3), 1.0 / 3.0)
    sd[E + "conv.0.bias"] = _randn(E + "conv.0.bias", seed, (d,), 0.05)
    sd[E + "conv.2.weight"] = _randn(E + "conv.2.weight", seed, (d, d, 3, 3), 1.4 / math.sqrt(9 * d))
    sd[E + "conv.2.bias"] = _randn(E + "conv.2.bias", seed, (d,), 0.05)
    lin(E + "out.0", d, d * cfg.sub_freq)

    for i in range(cfg.n_layers):
        L = f"encoder.encoders.{i}."
        for ff in ("feed_forward_macaron", "feed_forward"):
            lin(L + ff + ".w_1", f, d)
            lin(L + ff + ".w_2", d, f, gain=BRANCH_GAIN)
        for nm in ("norm_ff_macaron", "norm_mha", "norm_conv", "norm_ff", "norm_final"):
            norm(L + nm, d)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            lin(L + "self_attn." + nm, d, d, gain=(2.0 if nm in ("linear_q", "linear_k") else BRANCH_GAIN if nm == "linear_out" else 1.0))
        lin(L + "self_attn.linear_pos", d, d, bias=False)
        sd[L + "self_attn.pos_bias_u"] = _randn(L + "pos_bias_u", seed, (cfg.n_heads, cfg.head_dim), 0.1)
        sd[L + "self_attn.pos_bias_v"] = _randn(L + "pos_bias_v", seed, (cfg.n_heads, cfg.head_dim), 0.1)
        C = L + "conv_module."
        sd[C + "pointwise_conv1.weight"] = _randn(C + "pw1.w", seed, (2 * d, d, 1), 1.0 / math.sqrt(d))
        sd[C + "pointwise_conv1.bias"] = _randn(C + "pw1.b", seed, (2 * d,), 0.05)
        sd[C + "depthwise_conv.weight"] = _randn(C + "dw.w", seed, (d, 1, k), 1.0 / math.sqrt(k))
        sd[C + "depthwise_conv.bias"] = _randn(C + "dw.b", seed, (d,), 0.05)
        sd[C + "norm.weight"] = 1.0 + _randn(C + "bn.w", seed, (d,), 0.05)
        sd[C + "norm.bias"] = _randn(C + "bn.b", seed, (d,), 0.05)
        sd[C + "norm.running_mean"] = _randn(C + "bn.m", seed, (d,), 0.05)
        sd[C + "norm.running_var"] = 1.0 + 0.1 * torch.rand((d,), generator=torch.Generator().manual_seed(_seed_for(C + "bn.v", seed)))
        sd[C + "norm.num_batches_tracked"] = torch.tensor(1, dtype=torch.int64)
        sd[C + "pointwise_conv2.weight"] = _randn(C + "pw2.w", seed, (d, d, 1), BRANCH_GAIN / math.sqrt(d))
        sd[C + "pointwise_conv2.bias"] = _randn(C + "pw2.b", seed, (d,), 0.05)
    norm("encoder.after_norm", d)
    lin("ctc.ctc_lo", V, d, gain=2.0)
    emb = _randn("decoder.embed.weight", seed, (V, H), 1.0)
    emb[cfg.blank_id].zero_()                                     # Embedding(padding_idx=blank)
    sd["decoder.embed.weight"] = emb
    for l in range(cfg.pred_layers):
        P = f"decoder.decoder.{l}."
        sd[P + "weight_ih_l0"] = _randn(P + "weight_ih_l0", seed, (4 * H, H), 1.0 / math.sqrt(H))
        sd[P + "weight_hh_l0"] = _randn(P + "weight_hh_l0", seed, (4 * H, H), 1.0 / math.sqrt(H))
        sd[P + "bias_ih_l0"] = _randn(P + "bias_ih_l0", seed, (4 * H,), 0.05)
        sd[P + "bias_hh_l0"] = _randn(P + "bias_hh_l0", seed, (4 * H,), 0.05)
    lin("joint_network.lin_enc", J, d)
    lin("joint_network.lin_dec", J, H, bias=False, gain=dec_gain)
    lin("joint_network.lin_out", V, J, gain=6.0)                  # tanh keeps |a| <= 1: a larger output gain spreads the logits
    if blank_bias is None:
        # scanned with the CPU oracle (oracle/espnet.py greedy): ~70 tokens per 10 s utterance (358 frames) at the 120M shape
        blank_bias = {(512, 17, 2600): 15.5, (256, 2, 96): 9.0}.get((cfg.d_model, cfg.n_layers, cfg.vocab_size), 10.0)
    sd["joint_network.lin_out.bias"][cfg.blank_id] += float(blank_bias)
    # a CTC head whose blank posterior exceeds find_blank's 0.98 threshold (pkg/espnet-asr/src/ctc.py:29) on roughly half
    # of the frames, so that the 20 s windowing has gaps to cut at
    sd["ctc.ctc_lo.bias"][cfg.blank_id] += {(512, 17, 2600): 13.0, (256, 2, 96): 8.5}.get((cfg.d_model, cfg.n_layers, cfg.vocab_size), 10.0)
    return sd


def hann_periodic(n: int) -> np.ndarray:
    """torch.hann_window(n) (periodic=True), what ESPnet's Stft builds for window='hann'"""
    return torch.hann_window(n, periodic=True, dtype=torch.float32).numpy()


def prepare_weights_espnet(cfg: ModelConfig, sd: Dict[str, torch.Tensor], pos_cap: int = DEFAULT_POS_CAP, f32: bool = False):
    """-> dict name -> CPU tensor as registered with rs_set_tensor for an `family="espnet"` context.  The conformer blocks
    take the names and layouts of the NeMo path (`prepare_weights`): ESPnet's block is the same arithmetic with
    feed_forward_macaron / feed_forward for feed_forward1 / feed_forward2.  What is new:
      * "sub.conv0.*": Conv2d(1, C, 3, 2) tap-major; "sub.conv1.w": Conv2d(C, C, 3, 2) as the bf16 [C][9*C] matrix of the
        implicit GEMM, K ordered (kernel row, kernel column, input channel) like the channels-last patches the gather kernel
        lays out; "sub.out.w" columns permuted from (c, f) to (f, c);
      * "fe.mvn_mean" / "fe.mvn_istd": GlobalMVN; "fe.window": periodic Hann over win_length;
      * "final_norm.*" (encoder.after_norm), "ctc.w" / "ctc.b" (ctc.ctc_lo);
      * joint: lin_enc -> "joint.enc.*", lin_dec (no bias) -> "joint.pred.*" with a zero bias, lin_out -> "joint.out.*".
    """
    assert cfg.espnet
    want_f32 = bool(f32)
    out = {}
    used = set()
    f32t = lambda t: t.detach().to(torch.float32).contiguous()                     # noqa: E731

    class _Dense:
        """a GEMM weight: bf16 for the throughput mode and, for the float32 parity mode (`f32=True`, include/rs_asr.h
        "precision_f32"), once more unrounded as "<name>.f32" in the same layout"""
        def __init__(self, t):
            self.t = t.detach().to(torch.float32).contiguous()

    def bf(t):
        return _Dense(t)

    def get(key):
        if key not in sd:
            raise UnsupportedCheckpoint(f"checkpoint has no tensor {key!r} (architecture differs from the configuration?)")
        used.add(key)
        return sd[key]

    d, C = cfg.d_model, cfg.sub_channels
    fb = get("frontend.logmel.melmat").to(torch.float32).t().contiguous().numpy()          # [n_mels][n_freq]
    if fb.shape != (cfg.n_mels, cfg.n_fft // 2 + 1):
        raise UnsupportedCheckpoint(f"frontend.logmel.melmat has shape {tuple(fb.shape[::-1])}")
    idx, w = banded_filterbank(fb)
    out["fe.window"] = torch.from_numpy(hann_periodic(cfg.win_length))
    out["fe.twiddle"] = torch.from_numpy(fft_twiddles(cfg.n_fft))
    out["fe.fb_idx"] = torch.from_numpy(idx)
    out["fe.fb_w"] = torch.from_numpy(w)
    std = get("normalize.std").to(torch.float64).clamp_min(cfg.norm_eps)
    out["fe.mvn_mean"] = f32t(get("normalize.mean"))
    out["fe.mvn_istd"] = (1.0 / std).to(torch.float32).contiguous()

    E = "encoder.embed."
    out["sub.conv0.w"] = f32t(get(E + "conv.0.weight").reshape(C, 9).t())
    out["sub.conv0.b"] = f32t(get(E + "conv.0.bias"))
    out["sub.conv1.w"] = bf(get(E + "conv.2.weight").permute(0, 2, 3, 1).reshape(C, 9 * C))
    out["sub.conv1.b"] = f32t(get(E + "conv.2.bias"))
    F = cfg.sub_freq
    out["sub.out.w"] = bf(get(E + "out.0.weight").reshape(d, C, F).permute(0, 2, 1).reshape(d, F * C))
    out["sub.out.b"] = f32t(get(E + "out.0.bias"))

    for i in range(cfg.n_layers):
        L = f"encoder.encoders.{i}."
        p = f"L{i}."
        for short, long in (("ln_ff1", "norm_ff_macaron"), ("ln_att", "norm_mha"), ("ln_conv", "norm_conv"),
                            ("ln_ff2", "norm_ff"), ("ln_out", "norm_final")):
            out[p + short + ".g"] = f32t(get(L + long + ".weight"))
            out[p + short + ".b"] = f32t(get(L + long + ".bias"))
        for short, long in (("ff1", "feed_forward_macaron"), ("ff2", "feed_forward")):
            out[p + short + ".w1"] = bf(get(L + long + ".w_1.weight"))
            out[p + short + ".b1"] = f32t(get(L + long + ".w_1.bias"))
            out[p + short + ".w2"] = bf(get(L + long + ".w_2.weight"))
            out[p + short + ".b2"] = f32t(get(L + long + ".w_2.bias"))
        A = L + "self_attn."
        out[p + "att.qkv.w"] = bf(torch.cat([get(A + "linear_q.weight"), get(A + "linear_k.weight"), get(A + "linear_v.weight")], dim=0))
        out[p + "att.qkv.b"] = f32t(torch.cat([get(A + "linear_q.bias"), get(A + "linear_k.bias"), get(A + "linear_v.bias")], dim=0))
        out[p + "att.out.w"] = bf(get(A + "linear_out.weight"))
        out[p + "att.out.b"] = f32t(get(A + "linear_out.bias"))
        out[p + "att.pos.w"] = bf(get(A + "linear_pos.weight"))
        out[p + "att.bias_u"] = f32t(get(A + "pos_bias_u").reshape(-1))
        out[p + "att.bias_v"] = f32t(get(A + "pos_bias_v").reshape(-1))
        Cm = L + "conv_module."
        rows = glu_interleave_index(d)
        out[p + "conv.pw1.w"] = get(Cm + "pointwise_conv1.weight").squeeze(-1)[rows].detach().to(torch.float32).to(torch.bfloat16).contiguous()
        out[p + "conv.pw1.b"] = f32t(get(Cm + "pointwise_conv1.bias")[rows])
        if want_f32:      # the float32 conv kernel applies the GLU itself: ESPnet's own row order (values | gates)
            out[p + "conv.pw1.w.f32"] = f32t(get(Cm + "pointwise_conv1.weight").squeeze(-1))
            out[p + "conv.pw1.b.f32"] = f32t(get(Cm + "pointwise_conv1.bias"))
        g, b = get(Cm + "norm.weight").double(), get(Cm + "norm.bias").double()
        mu, var = get(Cm + "norm.running_mean").double(), get(Cm + "norm.running_var").double()
        sc = g / torch.sqrt(var + cfg.bn_eps)
        out[p + "conv.dw.w"] = f32t((get(Cm + "depthwise_conv.weight").double().squeeze(1) * sc[:, None]).float().t())
        out[p + "conv.dw.b"] = f32t(((get(Cm + "depthwise_conv.bias").double() - mu) * sc + b).float())
        out[p + "conv.pw2.w"] = bf(get(Cm + "pointwise_conv2.weight").squeeze(-1))
        out[p + "conv.pw2.b"] = f32t(get(Cm + "pointwise_conv2.bias"))
    out["final_norm.g"] = f32t(get("encoder.after_norm.weight"))
    out["final_norm.b"] = f32t(get("encoder.after_norm.bias"))
    # the CTC head's rows padded to a multiple of 4 (the GEMMs' N % 4 rule; a real token list has any length): zero rows
    # with a bias of -1e30, never read back (the softmax kernel runs over vocab_size columns and zeroes the rest)
    V = cfg.n_logits
    Vp = (V + 3) // 4 * 4
    cw = torch.zeros((Vp, d), dtype=torch.float32)
    cw[:V] = get("ctc.ctc_lo.weight").detach().to(torch.float32)
    cb = torch.full((Vp,), -1.0e30, dtype=torch.float32)
    cb[:V] = get("ctc.ctc_lo.bias").detach().to(torch.float32)
    out["ctc.w"] = bf(cw)
    out["ctc.b"] = cb
    out["joint.enc.w"] = bf(get("joint_network.lin_enc.weight"))
    out["joint.enc.b"] = f32t(get("joint_network.lin_enc.bias"))
    out["pred.embed"] = f32t(get("decoder.embed.weight"))
    H = cfg.pred_hidden
    for l in range(cfg.pred_layers):
        P = f"decoder.decoder.{l}."
        wl = torch.cat([get(P + "weight_ih_l0"), get(P + "weight_hh_l0")], dim=1)
        out[f"pred.lstm{l}.w"] = to_fragment_major(wl)
        perm = torch.arange(4 * H).view(4, H // 4, 4).permute(1, 0, 2).reshape(-1)
        out[f"pred.lstm{l}.w4"] = to_fragment_major(wl.to(torch.float32)[perm])
        out[f"pred.lstm{l}.b"] = f32t(get(P + "bias_ih_l0").float() + get(P + "bias_hh_l0").float())
    out["joint.pred.w"] = to_fragment_major(get("joint_network.lin_dec.weight"))
    out["joint.pred.b"] = torch.zeros((cfg.joint_hidden,), dtype=torch.float32)           # lin_dec has no bias
    out["joint.out.w"] = to_fragment_major(get("joint_network.lin_out.weight"))
    out["joint.out.b"] = f32t(get("joint_network.lin_out.bias"))
    out.update(screen_tensors(get("joint_network.lin_out.weight"), get("joint_network.lin_out.bias")))     # screened joint (greedy search)
    for name in [k for k, v in out.items() if isinstance(v, _Dense)]:
        t = out[name].t
        out[name] = t.to(torch.bfloat16).contiguous()
        if want_f32:
            out[name + ".f32"] = t
    out["pos.table"] = torch.from_numpy(rel_pos_table(cfg, pos_cap)).to(torch.bfloat16).contiguous()
    if want_f32:
        out["pos.table.f32"] = torch.from_numpy(rel_pos_table(cfg, pos_cap)).contiguous()
    benign = ("num_batches_tracked",)
    left = [k for k in sd if k not in used and not k.endswith(benign)]
    if left:
        raise UnsupportedCheckpoint(f"{len(left)} checkpoint tensor(s) have no counterpart in this implementation: "
                                    + ", ".join(left[:8]) + (" ..." if len(left) > 8 else ""))
    return out


# ------------------------------------------------------------------------------------
# ESPnet2 model directories / model-zoo archives
# ------------------------------------------------------------------------------------

def config_from_espnet_yaml(doc: dict) -> ModelConfig:
    """Map an ESPnet2 ASR training `config.yaml` (parsed) onto ModelConfig(family="espnet").  [UPSTREAM] key names follow
    espnet2/tasks/asr.py (`frontend_conf`, `normalize`, `encoder` / `encoder_conf`, `decoder` / `decoder_conf`,
    `joint_net_conf`, `token_list`) and the constructor arguments of the classes cited in this module's header.  Strict: a
    variant the kernels do not compute raises UnsupportedCheckpoint instead of loading silently."""
    def need(cond, what):
        if not cond:
            raise UnsupportedCheckpoint(f"config.yaml: {what} is not implemented by the gfx950 kernels")

    fe = doc.get("frontend_conf") or {}
    enc = doc.get("encoder_conf") or {}
    dec = doc.get("decoder_conf") or {}
    jn = doc.get("joint_net_conf") or {}
    tokens = doc.get("token_list")
    need(str(doc.get("frontend", "default")) == "default", f"frontend={doc.get('frontend')!r}")
    need(str(doc.get("normalize", "global_mvn")) == "global_mvn", f"normalize={doc.get('normalize')!r} (GlobalMVN only)")
    need(str(doc.get("encoder", "conformer")) == "conformer", f"encoder={doc.get('encoder')!r}")
    need(str(doc.get("decoder", "transducer")) == "transducer", f"decoder={doc.get('decoder')!r} (a transducer model is expected)")
    need(doc.get("preencoder") in (None, "null") and doc.get("postencoder") in (None, "null"), "pre / post encoders")
    need(isinstance(tokens, (list, tuple)) and len(tokens) > 2, "a token_list")
    need(str(enc.get("input_layer", "conv2d")) == "conv2d", f"encoder_conf.input_layer={enc.get('input_layer')!r}")
    # Absent keys take ESPnet's OWN constructor defaults ([UPSTREAM] espnet2 ConformerEncoder: rel_pos_type "legacy",
    # macaron_style False, zero_triu False; TransducerDecoder hidden_size 320; JointNetwork joint_space_size 256): config.yaml
    # stores encoder_conf as the recipe wrote it, so a recipe that omits rel_pos_type TRAINED the legacy attention — which has
    # the same parameter shapes as the latest one and would load silently.
    need(str(enc.get("pos_enc_layer_type", "rel_pos")) == "rel_pos" and str(enc.get("selfattention_layer_type", "rel_selfattn")) == "rel_selfattn",
         f"pos_enc_layer_type={enc.get('pos_enc_layer_type')!r} / selfattention_layer_type={enc.get('selfattention_layer_type')!r}")
    need(str(enc.get("rel_pos_type", "legacy")) == "latest",
         f"rel_pos_type={enc.get('rel_pos_type', 'legacy (ESPnet default when the key is absent)')!r}: the legacy relative-position attention")
    need(not enc.get("zero_triu", False), "zero_triu=True")
    need(bool(enc.get("macaron_style", False)) and bool(enc.get("use_cnn_module", True)), "a conformer without macaron FFN / conv module")
    need(bool(enc.get("normalize_before", True)) and not enc.get("concat_after", False), "post-norm / concat_after blocks")
    need(str(enc.get("activation_type", "swish")) == "swish", f"encoder_conf.activation_type={enc.get('activation_type')!r}")
    need(str(enc.get("positionwise_layer_type", "linear")) == "linear", "non-linear position-wise layers")
    need(not enc.get("interctc_layer_idx") and not enc.get("stochastic_depth_rate"), "intermediate CTC / stochastic depth at inference")
    need(str(dec.get("rnn_type", "lstm")) == "lstm", f"decoder_conf.rnn_type={dec.get('rnn_type')!r}")
    need(str(jn.get("joint_activation_type", "tanh")) == "tanh", f"joint_net_conf.joint_activation_type={jn.get('joint_activation_type')!r}")
    fs_raw = str(fe.get("fs", 16000))                    # ESPnet accepts "16k" (humanfriendly) as well as 16000
    fs = int(float(fs_raw[:-1]) * 1000) if fs_raw.lower().endswith("k") else int(fs_raw)
    n_fft = int(fe.get("n_fft", 512))
    hidden = int(dec.get("hidden_size", 320))
    need(int(dec.get("embed_size", hidden)) == hidden or "embed_size" not in dec, "decoder embed_size != hidden_size")
    need(not fe.get("htk", False) and int(fe.get("fmin") or 0) == 0 and fe.get("fmax") in (None, "null", fs // 2),
         "a mel filterbank other than Slaney 0 .. fs/2")
    d = int(enc.get("output_size", 256))
    return ModelConfig(
        family="espnet", sample_rate=fs, n_fft=n_fft, win_length=int(fe.get("win_length") or n_fft), hop_length=int(fe.get("hop_length", 128)), n_mels=int(fe.get("n_mels", 80)),
        preemph=0.0, log_guard=1e-10, norm_eps=1e-20,
        d_model=d, n_heads=int(enc.get("attention_heads", 4)), ff_dim=int(enc.get("linear_units", 2048)), n_layers=int(enc.get("num_blocks", 6)),
        conv_kernel=int(enc.get("cnn_module_kernel", 31)), sub_channels=d, sub_factor=4, xscaling=True, ln_eps=1e-12,
        vocab_size=len(tokens), pred_hidden=hidden, pred_layers=int(dec.get("num_layers", 1)),
        joint_hidden=int(jn.get("joint_space_size", 256)), max_symbols=1).validate()


def read_espnet(path: str):
    """Read an ESPnet2 ASR model — a directory or a model-zoo `.zip` holding the training `config.yaml`, the `*.pth` state
    dict and the GlobalMVN statistics (`feats_stats.npz`, named by `normalize_conf.stats_file`) — without ESPnet.
    -> (ModelConfig, state dict with "normalize.mean" / "normalize.std" added, token list).
    [UPSTREAM] layout of espnet_model_zoo archives (what `Speech2Text.from_pretrained` unpacks: pkg/espnet-asr/src/transcribe.py:26-32)."""
    import io
    import os
    import zipfile

    import yaml
    files = {}
    if os.path.isdir(path):
        for root, _, names in os.walk(path):
            for n in names:
                files[os.path.relpath(os.path.join(root, n), path)] = os.path.join(root, n)
        read = lambda k: open(files[k], "rb").read()          # noqa: E731
    else:
        zf = zipfile.ZipFile(path)
        files = {n: n for n in zf.namelist()}
        read = lambda k: zf.read(k)                           # noqa: E731
    cfg_key = next((k for k in sorted(files) if os.path.basename(k) == "config.yaml"), None)
    pth_key = next((k for k in sorted(files) if k.endswith(".pth")), None)
    if cfg_key is None or pth_key is None:
        raise ValueError(f"{path}: not an ESPnet2 model (config.yaml / *.pth missing)")
    doc = yaml.safe_load(read(cfg_key))
    cfg = config_from_espnet_yaml(doc)
    sd = torch.load(io.BytesIO(read(pth_key)), map_location="cpu", weights_only=True)
    token_type = str(doc.get("token_type", "char"))
    if token_type != "char":
        raise UnsupportedCheckpoint(f"{path}: token_type={token_type!r} (only character token lists are decoded to text here)")
    sd = dict(sd)
    if "normalize.mean" in sd and "normalize.std" in sd:
        # [UPSTREAM] GlobalMVN registers mean / std as buffers: a saved model normally carries them already
        return cfg, sd, list(doc["token_list"])
    stats_name = os.path.basename(str((doc.get("normalize_conf") or {}).get("stats_file", "feats_stats.npz")))
    st_key = next((k for k in sorted(files) if os.path.basename(k) == stats_name), None)
    if st_key is None:
        raise UnsupportedCheckpoint(f"{path}: GlobalMVN statistics {stats_name!r} not found (and no normalize.mean / .std in the state dict)")
    st = np.load(io.BytesIO(read(st_key)))
    count = float(st["count"])                        # [UPSTREAM] espnet2/layers/global_mvn.py: mean = sum / count, var = sum_sq / count - mean^2
    mean = st["sum"].astype(np.float64) / count
    var = st["sum_square"].astype(np.float64) / count - mean * mean
    sd["normalize.mean"] = torch.from_numpy(mean.astype(np.float32))
    sd["normalize.std"] = torch.from_numpy(np.sqrt(np.maximum(var, 1e-20)).astype(np.float32))
    return cfg, sd, list(doc["token_list"])


def write_espnet(path: str, cfg: ModelConfig, sd, token_list):
    """Write a model directory in the layout `read_espnet` reads (tests round-trip a synthetic model through it)."""
    import os

    import yaml
    os.makedirs(os.path.join(path, "exp", "asr_stats", "train"), exist_ok=True)
    os.makedirs(os.path.join(path, "exp", "asr_train"), exist_ok=True)
    mean, std = sd["normalize.mean"].double().numpy(), sd["normalize.std"].double().numpy()
    count = 1000.0
    np.savez(os.path.join(path, "exp", "asr_stats", "train", "feats_stats.npz"), count=np.array(count), sum=mean * count,
             sum_square=(std * std + mean * mean) * count)
    doc = {
        "frontend": "default", "frontend_conf": {"fs": "16k", "n_fft": cfg.n_fft, "win_length": cfg.win_length, "hop_length": cfg.hop_length,
                                                   "n_mels": cfg.n_mels},
        "normalize": "global_mvn", "normalize_conf": {"stats_file": "exp/asr_stats/train/feats_stats.npz"},
        "specaug": None, "preencoder": None, "postencoder": None,
        "encoder": "conformer",
        "encoder_conf": {"output_size": cfg.d_model, "attention_heads": cfg.n_heads, "linear_units": cfg.ff_dim, "num_blocks": cfg.n_layers,
                         "dropout_rate": 0.1, "positional_dropout_rate": 0.1, "attention_dropout_rate": 0.1, "input_layer": "conv2d",
                         "normalize_before": True, "macaron_style": True, "rel_pos_type": "latest", "pos_enc_layer_type": "rel_pos",
                         "selfattention_layer_type": "rel_selfattn", "activation_type": "swish", "use_cnn_module": True,
                         "cnn_module_kernel": cfg.conv_kernel},
        "decoder": "transducer",
        "decoder_conf": {"rnn_type": "lstm", "num_layers": cfg.pred_layers, "hidden_size": cfg.pred_hidden, "dropout": 0.1, "dropout_embed": 0.2},
        "joint_net_conf": {"joint_space_size": cfg.joint_hidden},
        "model_conf": {"ctc_weight": 0.3, "report_cer": False, "report_wer": False},
        "token_list": list(token_list), "token_type": "char", "init": None,
    }
    with open(os.path.join(path, "exp", "asr_train", "config.yaml"), "w") as fp:
        yaml.safe_dump(doc, fp, allow_unicode=True)
    keep = {k: v for k, v in sd.items() if not k.startswith("normalize.")}
    torch.save(keep, os.path.join(path, "exp", "asr_train", "valid.loss.ave.pth"))
