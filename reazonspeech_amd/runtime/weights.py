"""Weights of the FastConformer-RNNT path: seeded synthetic generator, `.nemo` reader,
and the host-side re-layout ("weight prep") into what the HIP kernels consume.

The reference obtains its weights with
`EncDecRNNTBPEModel.from_pretrained('reazon-research/reazonspeech-nemo-v2')`
(pkg/nemo-asr/src/transcribe.py:26-28): a `.nemo` tar holding `model_config.yaml`,
`model_weights.ckpt` and a SentencePiece `tokenizer.model`.  No checkpoint can be
fetched here, so `synthetic_state_dict` builds a state dict with NeMo's key names and
shapes from a seed (SURVEY.md §8d "Synthetic weights"); `read_nemo` reads a real one
when it is mounted.  Both feed `prepare_weights`.
"""
import io
import math
import tarfile
from typing import Dict

import numpy as np
import torch

from .config import ModelConfig, UnsupportedCheckpoint, from_nemo_yaml

# ------------------------------------------------------------------------------------
# front-end constants
# ------------------------------------------------------------------------------------

def slaney_mel_filterbank(cfg: ModelConfig) -> np.ndarray:
    """[n_mels, n_fft/2+1] float32 triangular filters, Slaney mel scale + Slaney area
    norm, 0 .. sr/2 — what `librosa.filters.mel(norm='slaney')` returns and NeMo's
    `FilterbankFeatures` stores as `fb` ([UPSTREAM]; closed form in SURVEY.md §10.5)."""
    sr, n_fft, n_mels = cfg.sample_rate, cfg.n_fft, cfg.n_mels
    fmin, fmax = 0.0, sr / 2.0
    f_sp = 200.0 / 3.0
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = math.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        lin = f / f_sp
        log = min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep
        return np.where(f >= min_log_hz, log, lin)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        lin = m * f_sp
        log = min_log_hz * np.exp(logstep * (m - min_log_mel))
        return np.where(m >= min_log_mel, log, lin)

    mel_pts = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    hz_pts = mel_to_hz(mel_pts)
    fft_freqs = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fft_freqs[None, :]
    fb = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        fb[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    fb *= enorm[:, None]
    return fb.astype(np.float32)


def hann_window(cfg: ModelConfig) -> np.ndarray:
    """symmetric (periodic=False) Hann of win_length samples, float32
    ([UPSTREAM] `torch.hann_window(win_length, periodic=False)`)."""
    return torch.hann_window(cfg.win_length, periodic=False, dtype=torch.float32).numpy()


def rel_pos_table(cfg: ModelConfig, T: int) -> np.ndarray:
    """[2T-1, d_model] float32; row n holds relative position r = T-1-n with
    P[n,2k] = sin(r*w_k), P[n,2k+1] = cos(r*w_k), w_k = 10000^(-2k/d)
    ([UPSTREAM] RelPositionalEncoding).  Computed like the reference
    stack does it: float32 frequencies, float32 product, float32 sin/cos."""
    d = cfg.d_model
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    pos = torch.arange(T - 1, -T, -1, dtype=torch.float32)
    ang = pos[:, None] * inv_freq[None, :]
    tab = torch.stack([ang.sin(), ang.cos()], dim=-1).reshape(2 * T - 1, d)
    return tab.numpy()


# ------------------------------------------------------------------------------------
# synthetic state dict (NeMo key names)
# ------------------------------------------------------------------------------------

# gain of the last linear of every residual branch.  Small on purpose: with O(1) branch gains a
# deep random-weight conformer collapses every frame onto the same vector (rank collapse), the
# joint logits stop depending on the frame and greedy decode emits either nothing or
# max_symbols tokens on every frame.  0.25 keeps ~60 % of the temporal variance through 24 layers.
BRANCH_GAIN = 0.25


def _seed_for(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def _randn(name, seed, shape, std):
    g = torch.Generator(device="cpu")
    g.manual_seed(_seed_for(name, seed))
    return (torch.randn(shape, generator=g, dtype=torch.float32) * std)


def synthetic_state_dict(cfg: ModelConfig, seed: int = 0, blank_bias: float = None, dec_gain: float = 1.0,
                         out_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with NeMo's state-dict keys and shapes.

    Every tensor draws from its own generator (seed x hash(name)), so the values do not
    depend on creation order or on which other tensors exist.  Linear/conv weights are
    N(0, 1/fan_in) so activations stay O(1); LayerNorm gamma=1 beta=0 (+ small noise so
    the affine path is exercised); BatchNorm running stats near (0, 1); the embedding
    row of the blank token is zero (NeMo `blank_as_pad`); the joint output bias of the
    blank gets `blank_bias` so greedy emits a realistic number of tokens
    (SURVEY.md §7.3 "Random weights make greedy pathological").
    """
    cfg.validate()
    sd: Dict[str, torch.Tensor] = {}
    d, f, c, H, J = cfg.d_model, cfg.ff_dim, cfg.sub_channels, cfg.pred_hidden, cfg.joint_hidden

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = _randn(name + ".weight", seed, (out_f, in_f), gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = _randn(name + ".bias", seed, (out_f,), 0.05)

    def norm(name, n):
        sd[name + ".weight"] = 1.0 + _randn(name + ".weight", seed, (n,), 0.05)
        sd[name + ".bias"] = _randn(name + ".bias", seed, (n,), 0.05)

    sd["preprocessor.featurizer.fb"] = torch.from_numpy(slaney_mel_filterbank(cfg))[None]
    sd["preprocessor.featurizer.window"] = torch.from_numpy(hann_window(cfg))

    # --- dw_striding subsampling: conv.0 | ReLU | (dw conv.2, pw conv.3, ReLU) | (conv.5, conv.6, ReLU)
    pre = "encoder.pre_encode."
    sd[pre + "conv.0.weight"] = _randn(pre + "conv.0.weight", seed, (c, 1, 3, 3), 1.0 / 3.0)
    sd[pre + "conv.0.bias"] = _randn(pre + "conv.0.bias", seed, (c,), 0.05)
    idx = 2
    for _ in range(cfg.n_sub_stages - 1):
        sd[pre + f"conv.{idx}.weight"] = _randn(pre + f"conv.{idx}.weight", seed, (c, 1, 3, 3), 1.0 / 3.0)
        sd[pre + f"conv.{idx}.bias"] = _randn(pre + f"conv.{idx}.bias", seed, (c,), 0.05)
        sd[pre + f"conv.{idx + 1}.weight"] = _randn(pre + f"conv.{idx + 1}.weight", seed, (c, c, 1, 1),
                                                    1.4 / math.sqrt(c))
        sd[pre + f"conv.{idx + 1}.bias"] = _randn(pre + f"conv.{idx + 1}.bias", seed, (c,), 0.05)
        idx += 3
    lin(pre + "out", d, c * cfg.sub_freq)

    for i in range(cfg.n_layers):
        L = f"encoder.layers.{i}."
        for ff in ("feed_forward1", "feed_forward2"):
            norm(L + "norm_" + ff, d)
            lin(L + ff + ".linear1", f, d)
            lin(L + ff + ".linear2", d, f, gain=BRANCH_GAIN)
        norm(L + "norm_self_att", d)
        for nm in ("linear_q", "linear_k", "linear_v", "linear_out"):
            lin(L + "self_attn." + nm, d, d,
                gain=(2.0 if nm in ("linear_q", "linear_k") else BRANCH_GAIN if nm == "linear_out" else 1.0))
        lin(L + "self_attn.linear_pos", d, d, bias=False)
        sd[L + "self_attn.pos_bias_u"] = _randn(L + "pos_bias_u", seed, (cfg.n_heads, cfg.head_dim), 0.1)
        sd[L + "self_attn.pos_bias_v"] = _randn(L + "pos_bias_v", seed, (cfg.n_heads, cfg.head_dim), 0.1)
        norm(L + "norm_conv", d)
        sd[L + "conv.pointwise_conv1.weight"] = _randn(L + "pw1.w", seed, (2 * d, d, 1), 1.0 / math.sqrt(d))
        sd[L + "conv.pointwise_conv1.bias"] = _randn(L + "pw1.b", seed, (2 * d,), 0.05)
        sd[L + "conv.depthwise_conv.weight"] = _randn(L + "dw.w", seed, (d, 1, cfg.conv_kernel),
                                                      1.0 / math.sqrt(cfg.conv_kernel))
        sd[L + "conv.depthwise_conv.bias"] = _randn(L + "dw.b", seed, (d,), 0.05)
        sd[L + "conv.batch_norm.weight"] = 1.0 + _randn(L + "bn.w", seed, (d,), 0.05)
        sd[L + "conv.batch_norm.bias"] = _randn(L + "bn.b", seed, (d,), 0.05)
        sd[L + "conv.batch_norm.running_mean"] = _randn(L + "bn.m", seed, (d,), 0.05)
        sd[L + "conv.batch_norm.running_var"] = 1.0 + 0.1 * torch.rand(
            (d,), generator=torch.Generator().manual_seed(_seed_for(L + "bn.v", seed)))
        sd[L + "conv.batch_norm.num_batches_tracked"] = torch.tensor(1, dtype=torch.int64)
        sd[L + "conv.pointwise_conv2.weight"] = _randn(L + "pw2.w", seed, (d, d, 1), BRANCH_GAIN / math.sqrt(d))
        sd[L + "conv.pointwise_conv2.bias"] = _randn(L + "pw2.b", seed, (d,), 0.05)
        norm(L + "norm_out", d)

    emb = _randn("decoder.prediction.embed.weight", seed, (cfg.n_logits, H), 1.0)
    emb[cfg.blank_id].zero_()
    sd["decoder.prediction.embed.weight"] = emb
    for l in range(cfg.pred_layers):
        P = "decoder.prediction.dec_rnn.lstm."
        sd[P + f"weight_ih_l{l}"] = _randn(P + f"weight_ih_l{l}", seed, (4 * H, H), 1.0 / math.sqrt(H))
        sd[P + f"weight_hh_l{l}"] = _randn(P + f"weight_hh_l{l}", seed, (4 * H, H), 1.0 / math.sqrt(H))
        sd[P + f"bias_ih_l{l}"] = _randn(P + f"bias_ih_l{l}", seed, (4 * H,), 0.05)
        sd[P + f"bias_hh_l{l}"] = _randn(P + f"bias_hh_l{l}", seed, (4 * H,), 0.05)
    lin("joint.pred", J, H, gain=dec_gain)   # dec_gain > 1: the prediction network weighs in the joint like a trained one's (beam-search recipes)
    lin("joint.enc", J, d)
    lin("joint.joint_net.2", cfg.n_logits, J, gain=2.0 * out_gain)   # out_gain > 1: peaked posteriors (beam-search recipes)
    if blank_bias is None:
        blank_bias = default_blank_bias(cfg)
    sd["joint.joint_net.2.bias"][cfg.blank_id] += float(blank_bias)
    return sd


def default_blank_bias(cfg: ModelConfig) -> float:
    """Blank-logit offset of the synthetic joint so that greedy emits on the order of
    5 tokens per audio-second (tuned with the CPU oracle, tests/golden/tune_blank_bias.py)."""
    return _BLANK_BIAS.get((cfg.d_model, cfg.n_layers, cfg.vocab_size), 3.0)


_BLANK_BIAS = {(1024, 24, 3000): 6.2, (256, 2, 63): 4.1}


# ------------------------------------------------------------------------------------
# .nemo reader
# ------------------------------------------------------------------------------------

def read_nemo(path: str):
    """Read a `.nemo` archive without NeMo / OmegaConf.

    Returns (ModelConfig, state_dict, tokenizer_model_bytes or None).  [UPSTREAM] layout:
    a (possibly gzip'd) tar with `model_config.yaml`, `model_weights.ckpt` (a pickled
    state dict) and `<hash>_tokenizer.model`.
    """
    import yaml
    cfg_dict, sd, tok = None, None, None
    with tarfile.open(path, "r:*") as tar:
        for member in tar.getmembers():
            name = member.name.split("/")[-1]
            if name == "model_config.yaml":
                cfg_dict = yaml.safe_load(tar.extractfile(member).read())
            elif name == "model_weights.ckpt":
                buf = io.BytesIO(tar.extractfile(member).read())
                sd = torch.load(buf, map_location="cpu", weights_only=True)
                if isinstance(sd, dict) and "state_dict" in sd and all(isinstance(k, str) for k in sd["state_dict"]):
                    sd = sd["state_dict"]                 # a Lightning-style checkpoint wrapped around the weights
            elif name.endswith("tokenizer.model"):        # NeMo prefixes artefacts with a content hash
                tok = tar.extractfile(member).read()
    if cfg_dict is None or sd is None:
        raise ValueError(f"{path}: not a .nemo archive (model_config.yaml / model_weights.ckpt missing)")
    target = str(cfg_dict.get("target", "") or "")
    if target and "RNNT" not in target and "Transducer" not in target and "Hybrid" not in target:
        raise UnsupportedCheckpoint(f"{path}: model class {target!r} is not an RNN-T model")
    return from_nemo_yaml(cfg_dict), sd, tok


def write_nemo(path: str, cfg: ModelConfig, sd, tokenizer_model: bytes = None):
    """Inverse of `read_nemo` (used by tests to round-trip a synthetic checkpoint)."""
    import yaml
    y = {
        "preprocessor": {"sample_rate": cfg.sample_rate, "n_fft": cfg.n_fft,
                         "window_size": cfg.win_length / cfg.sample_rate,
                         "window_stride": cfg.hop_length / cfg.sample_rate,
                         "features": cfg.n_mels, "preemph": cfg.preemph},
        "encoder": {"d_model": cfg.d_model, "n_heads": cfg.n_heads,
                    "ff_expansion_factor": cfg.ff_dim // cfg.d_model, "n_layers": cfg.n_layers,
                    "conv_kernel_size": cfg.conv_kernel,
                    "subsampling_conv_channels": cfg.sub_channels,
                    "subsampling_factor": cfg.sub_factor, "xscaling": cfg.xscaling,
                    "self_attention_model": "rel_pos" if cfg.att_left < 0 else "rel_pos_local_attn",
                    "att_context_size": [cfg.att_left, cfg.att_right],
                    "global_tokens": cfg.n_global},
        "decoder": {"vocab_size": cfg.vocab_size,
                    "prednet": {"pred_hidden": cfg.pred_hidden, "pred_rnn_layers": cfg.pred_layers}},
        "joint": {"num_classes": cfg.vocab_size, "jointnet": {"joint_hidden": cfg.joint_hidden}},
        "decoding": {"strategy": "greedy_batch", "greedy": {"max_symbols": cfg.max_symbols}},
    }
    with tarfile.open(path, "w") as tar:
        def add(name, data: bytes):
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
        add("./model_config.yaml", yaml.safe_dump(y).encode())
        buf = io.BytesIO()
        torch.save({k: v for k, v in sd.items()}, buf)
        add("./model_weights.ckpt", buf.getvalue())
        if tokenizer_model is not None:
            add("./0000_tokenizer.model", tokenizer_model)


# ------------------------------------------------------------------------------------
# weight prep: NeMo state dict -> tensors in the layouts librs_asr.so consumes
# ------------------------------------------------------------------------------------

FB_MAXW = 32          # taps per banded mel filter row (k_frontend.hip)
DEFAULT_POS_CAP = 1024  # rows of relative positions kept resident: T' up to 1024 (~82 s of audio)


def fft_twiddles(n_fft: int = 512) -> np.ndarray:
    """[n_fft/2][2] float32 (cos, -sin)(2*pi*j/n_fft), computed in float64."""
    j = np.arange(n_fft // 2, dtype=np.float64)
    ang = 2.0 * np.pi * j / n_fft
    return np.stack([np.cos(ang), -np.sin(ang)], axis=1).astype(np.float32)


def banded_filterbank(fb: np.ndarray):
    """dense [n_mels][n_bins] -> (idx int32 [n_mels][2] = first bin, taps; w float32 [n_mels][FB_MAXW])"""
    n_mels = fb.shape[0]
    idx = np.zeros((n_mels, 2), np.int32)
    w = np.zeros((n_mels, FB_MAXW), np.float32)
    for m in range(n_mels):
        nz = np.nonzero(fb[m])[0]
        if len(nz) == 0:
            continue
        k0, k1 = int(nz[0]), int(nz[-1]) + 1
        if k1 - k0 > FB_MAXW:
            raise ValueError(f"mel filter {m} has {k1 - k0} taps > {FB_MAXW}")
        idx[m] = (k0, k1 - k0)
        w[m, :k1 - k0] = fb[m, k0:k1]
    return idx, w


def to_fragment_major(w: torch.Tensor) -> torch.Tensor:
    """float32 [N][K] -> [ceil(N/16)][K/16][64 lanes][4]: the 16x16 block (n-tile, k-block) is
    stored in the order the v_mfma_f32_16x16x4_f32 B-operand lanes consume it (lane = 16*kk + li
    holds W[16*tn + li][16*kb + 4*kk + 0..3]), so a wave's operand load is one contiguous 1 KiB
    run instead of 64 row-strided 16-byte pieces.  Rows past N are zero."""
    n, k = w.shape
    assert k % 16 == 0
    nt = (n + 15) // 16
    wp = torch.zeros((nt * 16, k), dtype=torch.float32)
    wp[:n] = w.to(torch.float32)
    return wp.view(nt, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(nt, k // 16, 64, 4)


def glu_interleave_index(d: int) -> torch.Tensor:
    """row order of the registered pointwise_conv1 weight: blocks of 64 rows = 32 value rows (32j ..) followed
    by their 32 gate rows (d + 32j ..)"""
    assert d % 32 == 0, "GLU interleave needs d_model % 32 == 0"
    j = torch.arange(d // 32)[:, None]
    r = torch.arange(32)[None, :]
    return torch.cat([32 * j + r, d + 32 * j + r], dim=1).reshape(-1)


def screen_tensors(weight: torch.Tensor, bias: torch.Tensor) -> Dict[str, torch.Tensor]:
    """operands of the screened joint (k_rnnt.hip) for an output layer [V][J]: a bf16 copy for the screening GEMM (rows padded
    to a multiple of 16 with zeros, bias pad -3e38 so a padded column can never be a candidate), the float32 row-major copy
    the exact re-evaluation reads, and the largest row norm (rounded up: it scales an error BOUND)"""
    wo = weight.detach().to(torch.float32)
    n, j = wo.shape
    vpad = (n + 15) // 16 * 16
    w16 = torch.zeros((vpad, j), dtype=torch.bfloat16)
    w16[:n] = wo.to(torch.bfloat16)
    bpad = torch.full((vpad,), -3.0e38, dtype=torch.float32)
    bpad[:n] = bias.detach().to(torch.float32)
    wmax = float(wo.double().norm(dim=1).max()) * (1.0 + 2.0 ** -10)
    return {"joint.out.w16": w16.contiguous(), "joint.out.wrm": wo.contiguous(), "joint.out.bpad": bpad,
            "joint.out.wmax": torch.tensor([wmax, 0.0, 0.0, 0.0], dtype=torch.float32)}


def prepare_weights(cfg: ModelConfig, sd: Dict[str, torch.Tensor], pos_cap: int = DEFAULT_POS_CAP, f32: bool = False):
    """-> dict name -> CPU torch tensor (float32 / bfloat16 / int32) exactly as registered with
    rs_set_tensor (DESIGN.md "Weights in HBM").  Host-side transforms, all one-off:
      * GEMM weights -> bf16, [N][K] row-major (torch Linear layout already)
      * q,k,v projections concatenated to one [3d][d] GEMM
      * subsampling convs -> tap-major [9][C] float32; output Linear columns permuted from
        (c, f) to (f, c) order to match the channels-last activation layout
      * conv-module BatchNorm folded into the depthwise weights (float64 math, float32 store),
        stored tap-major [k][d]
      * LSTM: W = [W_ih | W_hh] ([4H][2H]) float32, bias = b_ih + b_hh (float32 add); the three
        float32 decode matrices (LSTM, joint.pred, joint output) are stored fragment-major
        (to_fragment_major) for contiguous MFMA operand loads
      * relative position table for T' up to pos_cap, bf16 [2*cap-1][d]
      * `f32=True` (the float32 parity mode, include/rs_asr.h "precision_f32"): every bf16 GEMM weight once more as
        "<name>.f32", unrounded, in the same layout — except conv.pw1, which keeps NeMo's row order (values | gates:
        the float32 conv kernel applies the GLU itself) — and the position table as "pos.table.f32"
    """
    out = {}
    want_f32 = bool(f32)
    bf = lambda t: t.detach().to(torch.float32).to(torch.bfloat16).contiguous()   # noqa: E731
    f32 = lambda t: t.detach().to(torch.float32).contiguous()                      # noqa: E731

    def dense(name, t):
        """a GEMM weight: bf16 for the throughput mode, and unrounded for the parity mode when asked"""
        out[name] = bf(t)
        if want_f32:
            out[name + ".f32"] = f32(t)

    C, d = cfg.sub_channels, cfg.d_model
    raw_sd, used = sd, set()

    class _Tracked:
        """records which checkpoint tensors the prep consumed; a bias the checkpoint does not have is zeros when
        the encoder was built with use_bias=False ([UPSTREAM] ConformerEncoder(use_bias=...)), an error otherwise"""

        def __getitem__(self, key):
            if key not in raw_sd:
                if key.endswith(".bias") and key.startswith("encoder.layers.") and not cfg.use_bias:
                    wkey = key[:-5] + ".weight"
                    used.add(key)
                    return torch.zeros((raw_sd[wkey].shape[0],), dtype=torch.float32)
                raise UnsupportedCheckpoint(f"checkpoint has no tensor {key!r} (architecture differs from model_config.yaml?)")
            used.add(key)
            return raw_sd[key]

        def __contains__(self, key):
            return key in raw_sd

    sd = _Tracked()

    fb = sd["preprocessor.featurizer.fb"].to(torch.float32).reshape(cfg.n_mels, -1).numpy()
    idx, w = banded_filterbank(fb)
    out["fe.window"] = f32(sd["preprocessor.featurizer.window"])
    out["fe.twiddle"] = torch.from_numpy(fft_twiddles(cfg.n_fft))
    out["fe.fb_idx"] = torch.from_numpy(idx)
    out["fe.fb_w"] = torch.from_numpy(w)

    pre = "encoder.pre_encode."
    tap_major = lambda t: f32(t.reshape(C, 9).t())                                 # noqa: E731
    out["sub.conv0.w"] = tap_major(sd[pre + "conv.0.weight"])
    out["sub.conv0.b"] = f32(sd[pre + "conv.0.bias"])
    ci = 2
    for s in range(1, cfg.n_sub_stages):
        out[f"sub.dw{s}.w"] = tap_major(sd[pre + f"conv.{ci}.weight"])
        out[f"sub.dw{s}.b"] = f32(sd[pre + f"conv.{ci}.bias"])
        dense(f"sub.pw{s}.w", sd[pre + f"conv.{ci + 1}.weight"].reshape(C, C))
        out[f"sub.pw{s}.b"] = f32(sd[pre + f"conv.{ci + 1}.bias"])
        ci += 3
    F = cfg.sub_freq
    wo = sd[pre + "out.weight"].reshape(d, C, F).permute(0, 2, 1).reshape(d, F * C)
    dense("sub.out.w", wo)
    out["sub.out.b"] = f32(sd[pre + "out.bias"])

    for i in range(cfg.n_layers):
        L = f"encoder.layers.{i}."
        p = f"L{i}."
        for short, long in (("ln_ff1", "norm_feed_forward1"), ("ln_att", "norm_self_att"),
                            ("ln_conv", "norm_conv"), ("ln_ff2", "norm_feed_forward2"), ("ln_out", "norm_out")):
            out[p + short + ".g"] = f32(sd[L + long + ".weight"])
            out[p + short + ".b"] = f32(sd[L + long + ".bias"])
        for short, long in (("ff1", "feed_forward1"), ("ff2", "feed_forward2")):
            dense(p + short + ".w1", sd[L + long + ".linear1.weight"])
            out[p + short + ".b1"] = f32(sd[L + long + ".linear1.bias"])
            dense(p + short + ".w2", sd[L + long + ".linear2.weight"])
            out[p + short + ".b2"] = f32(sd[L + long + ".linear2.bias"])
        A = L + "self_attn."
        dense(p + "att.qkv.w", torch.cat([sd[A + "linear_q.weight"], sd[A + "linear_k.weight"],
                                          sd[A + "linear_v.weight"]], dim=0))
        out[p + "att.qkv.b"] = f32(torch.cat([sd[A + "linear_q.bias"], sd[A + "linear_k.bias"],
                                              sd[A + "linear_v.bias"]], dim=0))
        dense(p + "att.out.w", sd[A + "linear_out.weight"])
        out[p + "att.out.b"] = f32(sd[A + "linear_out.bias"])
        dense(p + "att.pos.w", sd[A + "linear_pos.weight"])
        out[p + "att.bias_u"] = f32(sd[A + "pos_bias_u"].reshape(-1))
        out[p + "att.bias_v"] = f32(sd[A + "pos_bias_v"].reshape(-1))
        Cm = L + "conv."
        # rows interleaved in blocks of 32 (values 32j.., then their gates d + 32j..): a GLU pair lands in one
        # MFMA lane of the pw1 GEMM and is applied in its epilogue (include/rs_asr.h: RS_GEMM_GLU)
        glu_rows = glu_interleave_index(cfg.d_model)
        out[p + "conv.pw1.w"] = bf(sd[Cm + "pointwise_conv1.weight"].squeeze(-1)[glu_rows])
        out[p + "conv.pw1.b"] = f32(sd[Cm + "pointwise_conv1.bias"][glu_rows])
        if want_f32:
            out[p + "conv.pw1.w.f32"] = f32(sd[Cm + "pointwise_conv1.weight"].squeeze(-1))
            out[p + "conv.pw1.b.f32"] = f32(sd[Cm + "pointwise_conv1.bias"])
        g = sd[Cm + "batch_norm.weight"].double()
        b = sd[Cm + "batch_norm.bias"].double()
        mu = sd[Cm + "batch_norm.running_mean"].double()
        var = sd[Cm + "batch_norm.running_var"].double()
        sc = g / torch.sqrt(var + cfg.bn_eps)
        wdw = sd[Cm + "depthwise_conv.weight"].double().squeeze(1) * sc[:, None]      # [d][k]
        bdw = (sd[Cm + "depthwise_conv.bias"].double() - mu) * sc + b
        out[p + "conv.dw.w"] = f32(wdw.float().t())                                   # [k][d]
        out[p + "conv.dw.b"] = f32(bdw.float())
        dense(p + "conv.pw2.w", sd[Cm + "pointwise_conv2.weight"].squeeze(-1))
        out[p + "conv.pw2.b"] = f32(sd[Cm + "pointwise_conv2.bias"])

    dense("joint.enc.w", sd["joint.enc.weight"])
    out["joint.enc.b"] = f32(sd["joint.enc.bias"])
    out["pred.embed"] = f32(sd["decoder.prediction.embed.weight"])
    P = "decoder.prediction.dec_rnn.lstm."
    for l in range(cfg.pred_layers):
        wl = torch.cat([sd[P + f"weight_ih_l{l}"], sd[P + f"weight_hh_l{l}"]], dim=1)
        out[f"pred.lstm{l}.w"] = to_fragment_major(wl)
        # narrow-tile LSTM kernel: rows regrouped so that a 16-row tile holds the 4 gates of 4 consecutive units
        # (row ug*16 + gate*4 + u  <-  row gate*H + 4*ug + u)
        Hh = cfg.pred_hidden
        perm = torch.arange(4 * Hh).view(4, Hh // 4, 4).permute(1, 0, 2).reshape(-1)
        out[f"pred.lstm{l}.w4"] = to_fragment_major(wl.to(torch.float32)[perm])
        out[f"pred.lstm{l}.b"] = f32(sd[P + f"bias_ih_l{l}"].float() + sd[P + f"bias_hh_l{l}"].float())
    out["joint.pred.w"] = to_fragment_major(sd["joint.pred.weight"])
    out["joint.pred.b"] = f32(sd["joint.pred.bias"])
    # joint_net = [activation, (Dropout if dropout > 0), Linear]: the Linear's index is 1 or 2 — find it
    jkeys = sorted(k for k in raw_sd if k.startswith("joint.joint_net.") and k.endswith(".weight"))
    if len(jkeys) != 1:
        raise UnsupportedCheckpoint(f"expected exactly one joint.joint_net.N.weight, found {jkeys}")
    jout = jkeys[0][:-len(".weight")]
    if tuple(raw_sd[jkeys[0]].shape) != (cfg.n_logits, cfg.joint_hidden):
        raise UnsupportedCheckpoint(f"{jkeys[0]} has shape {tuple(raw_sd[jkeys[0]].shape)}, expected "
                                    f"({cfg.n_logits}, {cfg.joint_hidden}) (vocab_size + blank, joint_hidden)")
    out["joint.out.w"] = to_fragment_major(sd[jout + ".weight"])
    out["joint.out.b"] = f32(sd[jout + ".bias"])
    out.update(screen_tensors(sd[jout + ".weight"], sd[jout + ".bias"]))
    out["pos.table"] = torch.from_numpy(rel_pos_table(cfg, pos_cap)).to(torch.bfloat16).contiguous()
    if want_f32:
        out["pos.table.f32"] = torch.from_numpy(rel_pos_table(cfg, pos_cap)).contiguous()
    # anything under the model's own prefixes that was NOT consumed means the checkpoint holds parameters of a
    # variant this path does not compute (e.g. self_attn.global_q/k/v, conv.layer_norm, a second joint layer):
    # refuse instead of silently ignoring them.  Other top-level modules (ctc_decoder.*, spec_augmentation.*) and
    # non-parameter buffers are not part of the RNN-T inference path.
    core = ("encoder.", "decoder.", "joint.")
    benign = ("num_batches_tracked", "encoder.pos_enc.pe", "encoder.pos_emb_max_len")
    left = [k for k in raw_sd if k.startswith(core) and k not in used and not k.endswith(benign)]
    if left:
        raise UnsupportedCheckpoint(f"{len(left)} checkpoint tensor(s) have no counterpart in this implementation: "
                                    + ", ".join(left[:8]) + (" ..." if len(left) > 8 else ""))
    return out
