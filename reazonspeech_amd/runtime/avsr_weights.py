"""Weights of the AV-HuBERT encoder-decoder behind `reazonspeech.avsr`: a seeded synthetic generator under the REFERENCE's own
state-dict key names (pkg/avsr/src/avhubert/modeling_avhubert.py:119-152, :216-254, modeling_resnet.py:140-178, decoder.py:297-330,
:467-486 + transformers' HubertEncoder) — `AVHubertForConditionalGeneration.load_state_dict(sd, strict=True)` accepts it, which is
how tests/golden/make_avsr_golden.py makes the reference run on exactly these numbers — and the host-side re-layout into what
librs_asr.so's rs_avsr_* entry points consume (include/rs_asr.h).

No checkpoint is reachable here; a real one is a directory with `model.safetensors` / `pytorch_model.bin` + `config.json` under the
same key names (`read_avsr`)."""
import json
import math
import os
from typing import Dict

import numpy as np
import torch

from .avsr_config import AvsrConfig
from .config import UnsupportedCheckpoint
from .weights import _seed_for

BN_EPS = 1e-5
TRUNK = ((1, 64, 64, 1), (2, 64, 128, 2), (3, 128, 256, 2), (4, 256, 512, 2))      # (layer, inplanes, planes, stride of its first block)


def expected_shapes_avsr(cfg: AvsrConfig) -> Dict[str, tuple]:
    return {k: tuple(v.shape) for k, v in synthetic_state_dict_avsr(cfg, meta=True).items()}


def synthetic_state_dict_avsr(cfg: AvsrConfig, seed: int = 0, meta: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded random weights under the reference's keys.  Linear / conv weights N(0, gain / fan_in), norms around (1, 0), BatchNorm
    running statistics away from (0, 1) so that a wrong fold shows, PReLU slopes around 0.25, residual branches damped so that
    a deep random stack keeps O(1) activations.  meta = True: shapes only."""
    cfg.validate()
    sd: Dict[str, torch.Tensor] = {}

    def rnd(name, shape, std=1.0, mean=0.0, uniform=None):
        if meta:
            return torch.empty(shape, dtype=torch.float32, device="meta")
        g = torch.Generator().manual_seed(_seed_for(name, seed))
        if uniform is not None:
            return uniform[0] + (uniform[1] - uniform[0]) * torch.rand(shape, generator=g, dtype=torch.float32)
        return mean + std * torch.randn(shape, generator=g, dtype=torch.float32)

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = rnd(name + ".weight", (out_f, in_f), gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = rnd(name + ".bias", (out_f,), 0.05)

    def norm(name, n):
        sd[name + ".weight"] = rnd(name + ".weight", (n,), 0.1, 1.0)
        sd[name + ".bias"] = rnd(name + ".bias", (n,), 0.05)

    def bn(name, n):
        norm(name, n)
        sd[name + ".running_mean"] = rnd(name + ".running_mean", (n,), 0.2)
        sd[name + ".running_var"] = rnd(name + ".running_var", (n,), uniform=(0.5, 1.5))
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64, device="meta" if meta else None)

    def conv(name, cout, cin, *k, gain=1.4):
        sd[name + ".weight"] = rnd(name + ".weight", (cout, cin) + tuple(k), gain / math.sqrt(cin * int(np.prod(k))))

    d, ffn, dd, dffn, V = cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.decoder_embed_dim, cfg.decoder_ffn_embed_dim, cfg.vocab_size
    prelu = cfg.resnet_relu_type == "prelu"
    A = "avhubert."
    lin(A + "feature_extractor_audio.proj", d, cfg.audio_feat_dim)
    R = A + "feature_extractor_video.resnet."
    conv(R + "frontend3D.0", cfg.frontend_nout, 1, 5, 7, 7)
    bn(R + "frontend3D.1", cfg.frontend_nout)
    if prelu:
        sd[R + "frontend3D.2.weight"] = rnd(R + "frontend3D.2.weight", (cfg.frontend_nout,), 0.05, 0.25)
    for layer, inpl, planes, stride in TRUNK:
        for b in range(2):
            P = R + f"trunk.layer{layer}.{b}."
            conv(P + "conv1", planes, inpl if b == 0 else planes, 3, 3)
            bn(P + "bn1", planes)
            conv(P + "conv2", planes, planes, 3, 3, gain=0.7)
            bn(P + "bn2", planes)
            if prelu:
                sd[P + "relu1.weight"] = rnd(P + "relu1.weight", (planes,), 0.05, 0.25)
                sd[P + "relu2.weight"] = rnd(P + "relu2.weight", (planes,), 0.05, 0.25)
            if b == 0 and (stride != 1 or inpl != planes):
                conv(P + "downsample.0", planes, inpl, 1, 1, gain=1.0)
                bn(P + "downsample.1", planes)
    lin(A + "feature_extractor_video.proj", d, cfg.backend_out)
    if cfg.fused_dim != d:
        lin(A + "post_extract_proj", d, cfg.fused_dim)
    E = A + "encoder."
    sd[E + "pos_conv_embed.conv.bias"] = rnd(E + "pos_conv_embed.conv.bias", (d,), 0.05)
    sd[E + "pos_conv_embed.conv.parametrizations.weight.original0"] = rnd(E + "pos.g", (1, 1, cfg.conv_pos), 0.1, 1.0)
    sd[E + "pos_conv_embed.conv.parametrizations.weight.original1"] = rnd(E + "pos.v", (d, d // cfg.conv_pos_groups, cfg.conv_pos), 1.0)
    norm(E + "layer_norm", d)

    def attention(P, width, out_gain):
        for nm in ("k_proj", "v_proj", "q_proj"):
            lin(P + nm, width, width, gain=1.2)
        lin(P + "out_proj", width, width, gain=out_gain)

    for i in range(cfg.encoder_layers):
        P = E + f"layers.{i}."
        attention(P + "attention.", d, 0.3)                  # small branch gains: a deep random post-LayerNorm stack otherwise maps every frame to one vector
        norm(P + "layer_norm", d)
        lin(P + "feed_forward.intermediate_dense", ffn, d, gain=1.4)
        lin(P + "feed_forward.output_dense", d, ffn, gain=0.3)
        norm(P + "final_layer_norm", d)
    norm(A + "layer_norm", cfg.fused_dim)
    # token embeddings that dominate the decoder's residual stream (with small attention / FFN branches below) make the next token depend
    # on the previous one and on the clip: random weights otherwise fall into one repeated token after a step or two (tuned with
    # oracle/avsr.py: 12 different tokens in 12 steps at the 161M shape)
    sd["embed_tokens.weight"] = rnd("embed_tokens.weight", (V, dd), 4.0)
    if not meta:
        sd["embed_tokens.weight"][cfg.pad_token_id] = 0.0              # nn.Embedding(padding_idx=...)
    sd["decoder.pos_embed.position_embeddings"] = (torch.empty((cfg.max_target_positions, dd), device="meta") if meta
                                                   else sinusoidal_positions(cfg.max_target_positions, dd))
    norm("decoder.layer_norm", dd)
    for i in range(cfg.decoder_layers):
        P = f"decoder.layers.{i}."
        attention(P + "attention.", dd, 0.2)
        norm(P + "layer_norm", dd)
        attention(P + "encoder_attn.", dd, 0.2)
        norm(P + "encoder_layer_norm", dd)
        lin(P + "feed_forward.intermediate_dense", dffn, dd, gain=1.4)
        lin(P + "feed_forward.output_dense", dd, dffn, gain=0.3)
        norm(P + "final_layer_norm", dd)
    sd["lm_head.weight"] = rnd("lm_head.weight", (V, dd), 4.0 / math.sqrt(dd))
    return sd


def sinusoidal_positions(T: int, D: int) -> torch.Tensor:
    """decoder.py:49-64 SinusoidalPositionalEmbedding: angle[pos][j] = pos / 10000^(2 (j // 2) / D) in float64; even columns sin,
    odd columns cos, rounded to float32"""
    pos = np.arange(T, dtype=np.float64)[:, None]
    j = np.arange(D, dtype=np.float64)[None, :]
    ang = pos / np.power(10000.0, 2.0 * np.floor(j / 2.0) / D)
    out = np.empty((T, D), np.float32)
    out[:, 0::2] = np.sin(ang[:, 0::2]).astype(np.float32)
    out[:, 1::2] = np.cos(ang[:, 1::2]).astype(np.float32)
    return torch.from_numpy(out)


def pos_conv_weight(sd, prefix="avhubert.encoder.pos_conv_embed.conv.") -> torch.Tensor:
    """the effective kernel of the weight-normalised positional convolution ([UPSTREAM] torch.nn.utils.parametrizations.weight_norm
    with dim = 2: w[:, :, k] = g[k] * v[:, :, k] / ||v[:, :, k]||_F); older checkpoints store weight_g / weight_v"""
    if prefix + "parametrizations.weight.original0" in sd:
        g, v = sd[prefix + "parametrizations.weight.original0"], sd[prefix + "parametrizations.weight.original1"]
    elif prefix + "weight_g" in sd:
        g, v = sd[prefix + "weight_g"], sd[prefix + "weight_v"]
    else:
        return sd[prefix + "weight"].to(torch.float32)
    return torch._weight_norm(v.to(torch.float32), g.to(torch.float32), 2)


def pad_cols(w: torch.Tensor, mult: int = 32) -> torch.Tensor:
    n, k = w.shape
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros((n, kp), dtype=w.dtype)
    out[:, :k] = w
    return out


def pad_rows(w: torch.Tensor, mult: int = 4) -> torch.Tensor:
    n = w.shape[0]
    npad = (n + mult - 1) // mult * mult
    if npad == n:
        return w.contiguous()
    out = torch.zeros((npad,) + tuple(w.shape[1:]), dtype=w.dtype)
    out[:n] = w
    return out


def prepare_weights_avsr(cfg: AvsrConfig, sd: Dict[str, torch.Tensor]):
    """-> dict name -> CPU float32 tensor as registered with rs_set_tensor on a context made by rs_avsr_create.  Host transforms:
      * Conv3d / Conv2d kernels tap-major for channels-last activations: [kt*kh*kw][Cout] for the front-end's direct kernel,
        [Cout][(kh, kw, cin)] for the 3 x 3 convolutions (patch GEMMs), [Cout][Cin] for the 1 x 1 down-sampling ones;
      * BatchNorm in inference form, the two constants torch's own CPU kernel uses: alpha = w / sqrt(var + eps),
        beta = b - mean * alpha (float32);
      * the audio projection's K extent (104) zero-padded to 128; q / k / v projections concatenated ([3d][d]), the cross
        attention's k / v likewise; the positional convolution's effective (weight-normalised) kernel as [group][tap][cin][cout];
      * the output projection's rows padded to a multiple of 4."""
    cfg.validate()
    out, used = {}, set()
    f32 = lambda t: t.detach().to(torch.float32).contiguous()        # noqa: E731

    def get(key):
        if key not in sd:
            raise UnsupportedCheckpoint(f"checkpoint has no tensor {key!r} (architecture differs from the configuration?)")
        used.add(key)
        return sd[key]

    def bn(dst, src):
        w, b, m, v = (get(src + s).to(torch.float32) for s in (".weight", ".bias", ".running_mean", ".running_var"))
        used.add(src + ".num_batches_tracked")
        alpha = w / torch.sqrt(v + BN_EPS)
        out[dst + ".alpha"] = alpha.contiguous()
        out[dst + ".beta"] = (b - m * alpha).contiguous()

    def slope(dst, src, n):
        out[dst] = f32(get(src)) if cfg.resnet_relu_type == "prelu" else torch.zeros((n,), dtype=torch.float32)

    A = "avhubert."
    d = cfg.encoder_embed_dim
    out["fe.audio.w"] = pad_cols(f32(get(A + "feature_extractor_audio.proj.weight")))
    out["fe.audio.b"] = f32(get(A + "feature_extractor_audio.proj.bias"))
    R = A + "feature_extractor_video.resnet."
    c0 = cfg.frontend_nout
    out["v.conv3d.w"] = f32(get(R + "frontend3D.0.weight").reshape(c0, 5 * 7 * 7).t())           # [245][64]
    bn("v.bn0", R + "frontend3D.1")
    slope("v.prelu0", R + "frontend3D.2.weight", c0)
    for layer, inpl, planes, stride in ((1, 64, 64, 1), (2, 64, 128, 2), (3, 128, 256, 2), (4, 256, 512, 2)):
        for b in range(2):
            P, q = R + f"trunk.layer{layer}.{b}.", f"v.l{layer}.{b}."
            out[q + "conv1.w"] = f32(get(P + "conv1.weight").permute(0, 2, 3, 1).reshape(planes, -1))
            out[q + "conv2.w"] = f32(get(P + "conv2.weight").permute(0, 2, 3, 1).reshape(planes, -1))
            bn(q + "bn1", P + "bn1")
            bn(q + "bn2", P + "bn2")
            slope(q + "relu1", P + "relu1.weight", planes)
            slope(q + "relu2", P + "relu2.weight", planes)
            if b == 0 and (stride != 1 or inpl != planes):
                out[q + "ds.w"] = f32(get(P + "downsample.0.weight").reshape(planes, inpl))
                bn(q + "ds.bn", P + "downsample.1")
    out["v.proj.w"] = f32(get(A + "feature_extractor_video.proj.weight"))
    out["v.proj.b"] = f32(get(A + "feature_extractor_video.proj.bias"))
    out["fuse.ln.g"] = f32(get(A + "layer_norm.weight"))
    out["fuse.ln.b"] = f32(get(A + "layer_norm.bias"))
    if cfg.fused_dim != d:
        out["fuse.proj.w"] = f32(get(A + "post_extract_proj.weight"))
        out["fuse.proj.b"] = f32(get(A + "post_extract_proj.bias"))
    E = A + "encoder."
    for k in list(sd):
        if k.startswith(E + "pos_conv_embed.conv.") and not k.endswith("bias"):
            used.add(k)
    G, cg = cfg.conv_pos_groups, d // cfg.conv_pos_groups
    w = pos_conv_weight(sd, E + "pos_conv_embed.conv.")                                          # [d][cg][k]
    out["enc.pos.w"] = f32(w.reshape(G, cg, cg, cfg.conv_pos).permute(0, 3, 2, 1))                # [g][tap][cin][cout]
    out["enc.pos.b"] = f32(get(E + "pos_conv_embed.conv.bias"))
    out["enc.ln.g"] = f32(get(E + "layer_norm.weight"))
    out["enc.ln.b"] = f32(get(E + "layer_norm.bias"))

    def attn(dst, P, cross=False):
        qw, kw, vw = (get(P + f"{n}_proj.weight").to(torch.float32) for n in "qkv")
        qb, kb, vb = (get(P + f"{n}_proj.bias").to(torch.float32) for n in "qkv")
        if cross:
            out[dst + "q.w"], out[dst + "q.b"] = qw.contiguous(), qb.contiguous()
            out[dst + "kv.w"], out[dst + "kv.b"] = torch.cat([kw, vw]).contiguous(), torch.cat([kb, vb]).contiguous()
        else:
            out[dst + "qkv.w"], out[dst + "qkv.b"] = torch.cat([qw, kw, vw]).contiguous(), torch.cat([qb, kb, vb]).contiguous()
        out[dst + "o.w"] = f32(get(P + "out_proj.weight"))
        out[dst + "o.b"] = f32(get(P + "out_proj.bias"))

    def ln(dst, src):
        out[dst + ".g"], out[dst + ".b"] = f32(get(src + ".weight")), f32(get(src + ".bias"))

    def ffn(dst, P):
        out[dst + "ff1.w"], out[dst + "ff1.b"] = f32(get(P + "intermediate_dense.weight")), f32(get(P + "intermediate_dense.bias"))
        out[dst + "ff2.w"], out[dst + "ff2.b"] = f32(get(P + "output_dense.weight")), f32(get(P + "output_dense.bias"))

    for i in range(cfg.encoder_layers):
        P, q = E + f"layers.{i}.", f"E{i}."
        attn(q, P + "attention.")
        ln(q + "ln1", P + "layer_norm")
        ffn(q, P + "feed_forward.")
        ln(q + "ln2", P + "final_layer_norm")
    out["dec.embed"] = f32(get("embed_tokens.weight"))
    out["dec.pos"] = f32(get("decoder.pos_embed.position_embeddings"))
    ln("dec.ln", "decoder.layer_norm")
    for i in range(cfg.decoder_layers):
        P, q = f"decoder.layers.{i}.", f"D{i}."
        attn(q + "sa.", P + "attention.")
        ln(q + "ln1", P + "layer_norm")
        attn(q + "ca.", P + "encoder_attn.", cross=True)
        ln(q + "ln2", P + "encoder_layer_norm")
        ffn(q, P + "feed_forward.")
        ln(q + "ln3", P + "final_layer_norm")
    lm = get("embed_tokens.weight") if cfg.share_decoder_input_output_embed else get("lm_head.weight")
    used.add("lm_head.weight")
    out["dec.lm.w"] = pad_rows(f32(lm))
    left = [k for k in sd if k not in used]
    if left:
        raise UnsupportedCheckpoint(f"{len(left)} checkpoint tensor(s) have no counterpart in this implementation: " + ", ".join(left[:8]) + (" ..." if len(left) > 8 else ""))
    return out


def read_avsr(path: str):
    """a `from_pretrained`-style directory (config.json + model.safetensors or pytorch_model.bin) -> (AvsrConfig, state dict)"""
    with open(os.path.join(path, "config.json"), encoding="utf-8") as fp:
        raw = json.load(fp)
    fields = {f for f in AvsrConfig.__dataclass_fields__}
    cfg = AvsrConfig(**{k: v for k, v in raw.items() if k in fields and k != "family"}).validate()
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        sd = load_file(st)
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    return cfg, sd
