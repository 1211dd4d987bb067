"""CPU restatement of the icefall Zipformer2 transducer that `reazonspeech.k2.asr` runs through sherpa-onnx
(pkg/k2-asr/src/huggingface.py:73-83, transcribe.py:36-45): kaldi-style fbank features, encoder_embed, the Zipformer2 stacks,
the stateless decoder, the joiner and sherpa-onnx's offline greedy search.  TEST INFRASTRUCTURE — see oracle/__init__.py.

**PARITY UNPINNED.**  sherpa-onnx, onnxruntime, kaldi-native-fbank, icefall and the three ONNX files of
`reazon-research/reazonspeech-k2-v2` are neither vendored under /root/reference nor installable here, and the reference holds no
test vectors for this path.  Every function below restates the PUBLISHED upstream algorithm from the icefall / sherpa-onnx
sources as of 2024 and names the module it follows; each detail that a maintainer with those packages can falsify in minutes
is listed in DESIGN.md's table of [UPSTREAM] bets (window / mel-bank formulas, Conv2dSubsampling padding, the rel-shift
direction of the position scores, `-1` context tokens, the unk rule, the timestamp unit).

One utterance per call, without padding: exactly how the reference drives sherpa-onnx (one stream, decode_stream).  The
optional "bf16" recipe rounds where the HIP path stores / feeds bf16 (GEMM operands and weights, stored activations, attention
weights) so that the comparison tolerance measures the kernels and not the precision recipe.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

FLT_EPSILON = 1.1920928955078125e-07


# ------------------------------------------------------------------------------------------------------------------
# constant tables, from the published formulas in float64.  The oracle builds its OWN window, mel banks and relative-position
# rows: importing the product's arrays (reazonspeech_amd/runtime/k2_weights.py, what the HIP path uploads) would compare the
# device with itself on these three pieces.  tests/test_k2_host.py asserts the two sets agree to float32 resolution.
# ------------------------------------------------------------------------------------------------------------------
def layer_prefix(cfg, s: int, j: int) -> str:
    """[UPSTREAM] zipformer.py: a stack that runs at the full rate is a bare Zipformer2Encoder (`encoders.S.layers.J`), a
    down-sampled one is a DownsampledZipformer2Encoder wrapping it (`encoders.S.encoder.layers.J`)"""
    inner = "layers" if cfg.downsampling[s] == 1 else "encoder.layers"
    return f"encoder.encoders.{s}.{inner}.{j}."


def povey_window(n: int) -> np.ndarray:
    """[UPSTREAM] kaldi-native-fbank feature-window.cc, window_type "povey": w[i] = (0.5 - 0.5 cos(2 pi i / (n - 1))) ** 0.85.
    -> float64 [n]"""
    phase = 2.0 * math.pi / (n - 1) * np.arange(n, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(phase), 0.85)


def mel_scale(hz):
    """[UPSTREAM] kaldi MelBanks::MelScale: 1127 ln(1 + f / 700)"""
    return 1127.0 * np.log1p(np.asarray(hz, dtype=np.float64) / 700.0)


def kaldi_mel_banks(cfg) -> np.ndarray:
    """[UPSTREAM] kaldi-native-fbank mel-computations.cc MelBanks (no vtln): n_mels triangles, equally spaced IN MEL between
    low_freq and high_freq (a non-positive high_freq is an offset from Nyquist); the weight of FFT bin i (frequency i * sr / n_fft,
    i < n_fft / 2: the Nyquist bin is not used) is its position on the rising or falling edge, bins ON an outer edge weigh 0.
    -> float64 [n_mels][n_fft / 2 + 1] (last column zero)"""
    nyquist = cfg.sample_rate / 2.0
    high = cfg.high_freq if cfg.high_freq > 0 else nyquist + cfg.high_freq
    lo_m, hi_m = float(mel_scale(cfg.low_freq)), float(mel_scale(high))
    step = (hi_m - lo_m) / (cfg.n_mels + 1)
    edges = lo_m + step * np.arange(cfg.n_mels + 2, dtype=np.float64)              # left edge of bank b = edges[b]
    n_bins = cfg.n_fft // 2
    m = mel_scale(np.arange(n_bins, dtype=np.float64) * (cfg.sample_rate / cfg.n_fft))[None, :]      # [1][bins]
    left, centre, right = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    rising = (m - left) / (centre - left)
    falling = (right - m) / (right - centre)
    inside = (m > left) & (m < right)
    banks = np.where(inside, np.where(m <= centre, rising, falling), 0.0)
    return np.concatenate([banks, np.zeros((cfg.n_mels, 1))], axis=1)


def compact_rel_pos_table(cfg, T: int) -> np.ndarray:
    """[UPSTREAM] zipformer.py CompactRelPositionalEncoding.extend_pe (length_factor 1.0) for relative positions -(T-1) .. T-1
    (row n <-> key index minus query index = n - (T - 1)):  the offset is compressed logarithmically,
    x_c = sqrt(D) sign(x) (ln(|x| + sqrt(D)) - ln sqrt(D)), squashed by atan(x_c 2 pi / D), and the angle feeds D / 2 harmonics:
    even columns cos(k a), odd columns sin(k a), k = 1 .. D / 2; the LAST column is overwritten with 1 (a bias input).
    -> float64 [2 T - 1][D]"""
    D = cfg.pos_dim
    x = np.arange(-(T - 1), T, dtype=np.float64)
    c = math.sqrt(D)
    xc = c * np.sign(x) * (np.log(np.abs(x) + c) - math.log(c))
    angle = np.arctan(xc / (D / (2.0 * math.pi)))
    k = np.arange(1, D // 2 + 1, dtype=np.float64)
    pe = np.empty((x.shape[0], D), np.float64)
    pe[:, 0::2] = np.cos(angle[:, None] * k[None, :])
    pe[:, 1::2] = np.sin(angle[:, None] * k[None, :])
    pe[:, D - 1] = 1.0
    return pe


def _rb(x, recipe):
    return x.to(torch.bfloat16).to(torch.float32) if recipe == "bf16" else x


def swoosh_l(x):
    """[UPSTREAM] scaling.py SwooshL: log(1 + exp(x - 4)) - 0.08 x - 0.035"""
    return F.softplus(x - 4.0) - 0.08 * x - 0.035


def swoosh_r(x):
    """[UPSTREAM] scaling.py SwooshR: log(1 + exp(x - 1)) - 0.08 x - 0.313261687"""
    return F.softplus(x - 1.0) - 0.08 * x - 0.313261687


def bias_norm(x, bias, log_scale):
    """[UPSTREAM] scaling.py BiasNorm: x * (mean((x - bias)^2, channel) ** -0.5 * exp(log_scale)) — x itself is scaled, not x - bias"""
    scales = ((x - bias) ** 2).mean(dim=-1, keepdim=True) ** -0.5 * log_scale.exp()
    return x * scales


# ------------------------------------------------------------------------------------------------------------------
# features
# ------------------------------------------------------------------------------------------------------------------
def fbank(cfg, wav: torch.Tensor) -> torch.Tensor:
    """wav f32[L] in [-1, 1] -> log-mel filterbank energies f32[T][n_mels], T = (L + shift / 2) // shift.

    [UPSTREAM] sherpa-onnx FeatureExtractor over kaldi-native-fbank with sherpa's defaults (normalize_samples: the floats go
    in unscaled; dither 0; snip_edges false; remove_dc_offset; preemphasis 0.97; povey window; power spectrum; 80 mel bins
    20 .. 7600 Hz; log with a floor of FLT_EPSILON; no energy, no CMVN):
      frame f covers samples [160 f - 120, 160 f + 280); positions outside the signal are REFLECTED (-1 -> 0, -2 -> 1, ..);
      per frame: subtract the mean; x[i] -= 0.97 x[i - 1] from the back, x[0] -= 0.97 x[0]; multiply by the window; zero-pad
      to 512; |FFT|^2; mel filters over the bins below the Nyquist bin."""
    L = wav.numel()
    T = cfg.fbank_frames(L)
    N, S = cfg.frame_length, cfg.frame_shift
    if T == 0:
        return torch.zeros((0, cfg.n_mels), dtype=torch.float32)
    idx = (torch.arange(T) * S + S // 2 - N // 2).unsqueeze(1) + torch.arange(N).unsqueeze(0)
    for _ in range(4):                                     # reflect until inside (a signal shorter than a window needs more than one bounce)
        idx = torch.where(idx < 0, -idx - 1, idx)
        idx = torch.where(idx >= L, 2 * L - 1 - idx, idx)
    fr = wav.to(torch.float32)[idx]                        # [T][N]
    fr = fr - fr.mean(dim=1, keepdim=True)
    prev = torch.cat([fr[:, :1], fr[:, :-1]], dim=1)
    fr = fr - cfg.preemph * prev
    fr = fr * torch.from_numpy(povey_window(N).astype(np.float32))
    spec = torch.fft.rfft(F.pad(fr, (0, cfg.n_fft - N)), dim=1)
    power = spec.real ** 2 + spec.imag ** 2                # [T][257]
    mel = power @ torch.from_numpy(kaldi_mel_banks(cfg).astype(np.float32)).t()
    return torch.log(torch.clamp(mel, min=FLT_EPSILON))


# ------------------------------------------------------------------------------------------------------------------
# encoder_embed
# ------------------------------------------------------------------------------------------------------------------
def encoder_embed(cfg, sd, feats: torch.Tensor, recipe="fp32", taps=None) -> torch.Tensor:
    """[UPSTREAM] subsampling.py Conv2dSubsampling: Conv2d(1, 8, 3, padding (0, 1)) SwooshR, Conv2d(8, 32, 3, stride 2) SwooshR,
    Conv2d(32, 128, 3, stride (1, 2)) SwooshR, one ConvNeXt block (depthwise 7x7, 1x1 to 3x channels, SwooshL, 1x1 back,
    residual), Linear over the (channel, frequency)-flattened map, BiasNorm.  feats [T][80] -> [(T - 7) // 2][encoder_dim[0]]"""
    E = "encoder_embed."
    rb = lambda t: _rb(t, recipe)        # noqa: E731
    x = feats[None, None]                # (1, 1, T, F)
    x = rb(swoosh_r(F.conv2d(x, sd[E + "conv.0.weight"], sd[E + "conv.0.bias"], padding=(0, 1))))
    x = rb(swoosh_r(F.conv2d(x, rb(sd[E + "conv.4.weight"]), sd[E + "conv.4.bias"], stride=2)))
    x = swoosh_r(F.conv2d(x, rb(sd[E + "conv.7.weight"]), sd[E + "conv.7.bias"], stride=(1, 2)))
    if taps is not None:
        taps["embed_conv"] = x[0].permute(1, 2, 0).clone()          # (T3, F, C)
    c3 = x.shape[1]
    y = rb(F.conv2d(x, sd[E + "convnext.depthwise_conv.weight"], sd[E + "convnext.depthwise_conv.bias"], padding=3, groups=c3))
    y = rb(swoosh_l(F.conv2d(y, rb(sd[E + "convnext.pointwise_conv1.weight"]), sd[E + "convnext.pointwise_conv1.bias"])))
    y = F.conv2d(y, rb(sd[E + "convnext.pointwise_conv2.weight"]), sd[E + "convnext.pointwise_conv2.bias"])
    x = x + y
    b, c, t, f = x.shape
    flat = rb(x.transpose(1, 2).reshape(t, c * f))
    out = flat @ rb(sd[E + "out.weight"]).t() + sd[E + "out.bias"]
    out = bias_norm(out, sd[E + "out_norm.bias"], sd[E + "out_norm.log_scale"])
    assert out.shape[0] == cfg.embed_frames(feats.shape[0])
    return out


# ------------------------------------------------------------------------------------------------------------------
# Zipformer2
# ------------------------------------------------------------------------------------------------------------------
def _lin(x, sd, name, recipe, act=None, store=True):
    """Linear on bf16-rounded operands (recipe "bf16"): y = act(rb(x) @ rb(W)^T + b), stored rounded unless it is a residual branch"""
    y = _rb(x, recipe) @ _rb(sd[name + ".weight"], recipe).t()
    if name + ".bias" in sd:
        y = y + sd[name + ".bias"]
    if act is not None:
        y = act(y)
    return _rb(y, recipe) if store else y


def attention_weights(cfg, sd, L, x, pos_proj_fn, heads, recipe):
    """[UPSTREAM] zipformer.py RelPositionMultiheadAttentionWeights.forward (eval, no masks): in_proj -> (q | k | p);
    scores[h, i, j] = q_i . k_j + p_i . (linear_pos pe)[rel = j - i] (the as_strided rel-shift: column (T - 1) - i + j of the
    (T, 2T - 1) position scores); softmax over j.  No 1 / sqrt(d) factor (it lives in the learned in_proj scale).
    x [T][d] -> [heads][T][T]"""
    T = x.shape[0]
    qd, pd = cfg.query_head_dim, cfg.pos_head_dim
    u = _lin(x, sd, L + "self_attn_weights.in_proj", recipe)
    q = u[:, :heads * qd].reshape(T, heads, qd).permute(1, 0, 2)
    k = u[:, heads * qd:2 * heads * qd].reshape(T, heads, qd).permute(1, 0, 2)
    p = u[:, 2 * heads * qd:].reshape(T, heads, pd).permute(1, 0, 2)
    pos = pos_proj_fn(T).reshape(2 * T - 1, heads, pd).permute(1, 2, 0)          # [h][pd][2T-1]
    pos_scores = p @ pos                                                            # [h][T][2T-1]
    i = torch.arange(T).unsqueeze(1)
    j = torch.arange(T).unsqueeze(0)
    pos_scores = pos_scores.gather(2, (j - i + T - 1).unsqueeze(0).expand(heads, T, T))
    scores = q @ k.transpose(1, 2) + pos_scores
    return _rb(scores.softmax(dim=-1), recipe)


def feed_forward(sd, name, x, recipe):
    """[UPSTREAM] FeedforwardModule: Linear -> SwooshL -> Linear"""
    h = _lin(x, sd, name + ".in_proj", recipe, act=swoosh_l)
    return _lin(h, sd, name + ".out_proj", recipe, store=False)


def nonlin_attention(sd, name, x, w0, recipe):
    """[UPSTREAM] NonlinAttention: in_proj -> (s | v | y); v * tanh(s); attention with the FIRST head's weights; * y; out_proj"""
    u = _lin(x, sd, name + ".in_proj", recipe)
    s, v, y = u.chunk(3, dim=1)
    v = _rb(v * torch.tanh(s), recipe)
    o = _rb((w0 @ v) * y, recipe)
    return _lin(o, sd, name + ".out_proj", recipe, store=False)


def self_attention(cfg, sd, name, x, w, recipe):
    """[UPSTREAM] SelfAttention: in_proj -> per-head values (12 wide), attention, out_proj"""
    T, heads = x.shape[0], w.shape[0]
    v = _lin(x, sd, name + ".in_proj", recipe).reshape(T, heads, -1).permute(1, 0, 2)     # [h][T][vd]
    o = _rb((w @ v).permute(1, 0, 2).reshape(T, -1), recipe)
    return _lin(o, sd, name + ".out_proj", recipe, store=False)


def conv_module(sd, name, x, recipe):
    """[UPSTREAM] ConvolutionModule (non-causal): in_proj -> (x | s); x * sigmoid(s); depthwise Conv1d(k, padding k // 2);
    SwooshR; out_proj"""
    u = _rb(x, recipe) @ _rb(sd[name + ".in_proj.weight"], recipe).t() + sd[name + ".in_proj.bias"]
    a, s = u.chunk(2, dim=1)
    g = _rb(a * torch.sigmoid(s), recipe)
    w = sd[name + ".depthwise_conv.weight"]
    k = w.shape[-1]
    c = F.conv1d(g.t()[None], w, sd[name + ".depthwise_conv.bias"], padding=k // 2, groups=w.shape[0])[0].t()
    c = _rb(swoosh_r(c), recipe)
    return _lin(c, sd, name + ".out_proj", recipe, store=False)


def bypass(x_orig, x, scale):
    """[UPSTREAM] BypassModule (eval): x_orig + (x - x_orig) * bypass_scale"""
    return x_orig + (x - x_orig) * scale


def encoder_layer(cfg, sd, L, x, pos_proj_fn, heads, recipe):
    """[UPSTREAM] Zipformer2EncoderLayer.forward (eval): the attention weights are computed once from the layer's input and
    shared by the non-linear attention (head 0) and both self-attention modules"""
    x0 = x
    w = attention_weights(cfg, sd, L, x, pos_proj_fn, heads, recipe)
    x = x + feed_forward(sd, L + "feed_forward1", x, recipe)
    x = x + nonlin_attention(sd, L + "nonlin_attention", x, w[0], recipe)
    x = x + self_attention(cfg, sd, L + "self_attn1", x, w, recipe)
    x = x + conv_module(sd, L + "conv_module1", x, recipe)
    x = x + feed_forward(sd, L + "feed_forward2", x, recipe)
    x = bypass(x0, x, sd[L + "bypass_mid.bypass_scale"])
    x = x + self_attention(cfg, sd, L + "self_attn2", x, w, recipe)
    x = x + conv_module(sd, L + "conv_module2", x, recipe)
    x = x + feed_forward(sd, L + "feed_forward3", x, recipe)
    x = bias_norm(x, sd[L + "norm.bias"], sd[L + "norm.log_scale"])
    return bypass(x0, x, sd[L + "bypass.bypass_scale"])


def simple_downsample(x, bias, ds):
    """[UPSTREAM] SimpleDownsample: pad to a multiple of ds by repeating the LAST frame, softmax(bias)-weighted sum of each group"""
    T = x.shape[0]
    Td = (T + ds - 1) // ds
    pad = Td * ds - T
    if pad:
        x = torch.cat([x, x[-1:].expand(pad, -1)], dim=0)
    w = torch.softmax(bias.to(torch.float32), dim=0)
    return (x.reshape(Td, ds, -1) * w[None, :, None]).sum(dim=1)


def convert_channels(x, d):
    if d <= x.shape[1]:
        return x[:, :d]
    return F.pad(x, (0, d - x.shape[1]))


def zipformer(cfg, sd, x: torch.Tensor, recipe="fp32", taps=None) -> torch.Tensor:
    """[UPSTREAM] Zipformer2.forward (eval, non-causal): stacks at 1/1, 1/2, 1/4, 1/8, 1/4, 1/2 of the 50 Hz rate, each fed the
    previous stack's output cut or zero-padded to its width; output = the last stack's channels extended by the extra channels
    of earlier, wider stacks; SimpleDownsample by 2.  x [T][encoder_dim[0]] -> [(T + 1) // 2][max(encoder_dim)]"""
    pe_cache = {}

    def pos_rows(T):
        if T not in pe_cache:
            pe_cache[T] = torch.from_numpy(compact_rel_pos_table(cfg, T).astype(np.float32))
        return pe_cache[T]

    outputs = []
    for s in range(cfg.n_stacks):
        d, ds, heads = cfg.encoder_dim[s], cfg.downsampling[s], cfg.num_heads[s]
        x = convert_channels(x, d)
        src_orig = x
        if ds > 1:
            x = simple_downsample(x, sd[f"encoder.encoders.{s}.downsample.bias"], ds)
        for j in range(cfg.num_layers[s]):
            L = layer_prefix(cfg, s, j)
            wp = sd[L + "self_attn_weights.linear_pos.weight"].to(torch.float32)
            x = encoder_layer(cfg, sd, L, x, lambda T, wp=wp: pos_rows(T) @ wp.t(), heads, recipe)
            if taps is not None:
                taps[f"S{s}.L{j}"] = x.clone()
        if ds > 1:
            up = x.unsqueeze(1).expand(-1, ds, -1).reshape(-1, d)[:src_orig.shape[0]]       # SimpleUpsample, cut to the input length
            x = bypass(src_orig, up, sd[f"encoder.encoders.{s}.out_combiner.bypass_scale"])
        outputs.append(x)
        if taps is not None:
            taps[f"S{s}"] = x.clone()
    pieces = [outputs[-1]]
    cur = cfg.encoder_dim[-1]
    for s in range(cfg.n_stacks - 2, -1, -1):
        d = cfg.encoder_dim[s]
        if d > cur:
            pieces.append(outputs[s][:, cur:d])
            cur = d
    assert cur == cfg.out_dim
    x = torch.cat(pieces, dim=1)
    return simple_downsample(x, sd["encoder.downsample_output.bias"], cfg.output_downsampling)


def forward(cfg, sd, wav, recipe="fp32", taps=None):
    """wav f32[L] -> dict(feats, enc [T'][out_dim], joint_enc = joiner.encoder_proj(enc) [T'][joiner_dim]) — what the ONNX
    encoder returns ([UPSTREAM] export-onnx.py OnnxEncoder folds encoder_proj into the encoder graph)"""
    with torch.no_grad():
        wav = torch.as_tensor(wav, dtype=torch.float32).reshape(-1)
        feats = fbank(cfg, wav)
        if taps is not None:
            taps["feats"] = feats.clone()
        x = encoder_embed(cfg, sd, feats, recipe, taps)
        if taps is not None:
            taps["embed"] = x.clone()
        enc = zipformer(cfg, sd, x, recipe, taps)
        f = _rb(enc, recipe) @ _rb(sd["joiner.encoder_proj.weight"], recipe).t() + sd["joiner.encoder_proj.bias"]
    assert enc.shape[0] == cfg.enc_frames(feats.shape[0])
    return {"feats": feats, "enc": enc, "joint_enc": f}


# ------------------------------------------------------------------------------------------------------------------
# stateless decoder, joiner, greedy search
# ------------------------------------------------------------------------------------------------------------------
def decoder_out(cfg, sd, context):
    """[UPSTREAM] decoder.py Decoder.forward(need_pad=False) + joiner.decoder_proj (export-onnx.py OnnxDecoder): embedding of the
    last `context_size` tokens (a token of -1 embeds to zero), grouped Conv1d over them (groups = decoder_dim // 4, no bias),
    ReLU, decoder_proj.  context: list of context_size ints -> [joiner_dim]"""
    y = torch.tensor(context, dtype=torch.long)
    emb = sd["decoder.embedding.weight"][y.clamp(min=0)] * (y >= 0).unsqueeze(-1)      # [ctx][D]
    D = emb.shape[1]
    h = F.conv1d(emb.t()[None], sd["decoder.conv.weight"], None, groups=D // 4)[0, :, 0]
    h = F.relu(h)
    return h @ sd["joiner.decoder_proj.weight"].t() + sd["joiner.decoder_proj.bias"]


def greedy_search(cfg, sd, f: torch.Tensor):
    """[UPSTREAM] sherpa-onnx OfflineTransducerGreedySearchDecoder::Decode: tokens start as [-1, .., -1, blank]; per encoder
    frame ONE joiner evaluation logits = output_linear(tanh(enc_proj_t + dec_proj)), y = argmax; y is emitted unless it is the
    blank (0) or `<unk>`; after an emission the decoder runs on the last context_size tokens.  timestamps = frame indices.
    f = joint_enc [T'][J] -> (ids, frames)"""
    wo, bo = sd["joiner.output_linear.weight"], sd["joiner.output_linear.bias"]
    hist = [-1] * (cfg.context_size - 1) + [cfg.blank_id]
    ids, frames = [], []
    with torch.no_grad():
        g = decoder_out(cfg, sd, hist[-cfg.context_size:])
        for t in range(f.shape[0]):
            y = int(torch.argmax(torch.tanh(f[t] + g) @ wo.t() + bo))
            if y != cfg.blank_id and y != cfg.unk_id:
                ids.append(y)
                frames.append(t)
                hist.append(y)
                g = decoder_out(cfg, sd, hist[-cfg.context_size:])
    return ids, frames
