"""CPU restatement of the AV-HuBERT encoder-decoder that `reazonspeech.avsr` runs (pkg/avsr/src/avhubert/modeling_avhubert.py,
modeling_resnet.py, decoder.py; the encoder stack is transformers' HubertEncoder, the search transformers' GenerationMixin).
TEST INFRASTRUCTURE — see oracle/__init__.py.

**PARITY PINNED to the reference itself**: tests/golden/avsr_ref_{tiny,base}.npz are outputs of the reference's own modules
(imported unchanged by oracle/_ref_avsr.py, generator tests/golden/make_avsr_golden.py) on this repo's synthetic weights and
inputs; tests/test_oracle_avsr.py holds every function below to them (encoder taps and output <= 2e-4, teacher-forced logits
<= 5e-4, greedy and beam-search ids identical, beam scores <= 1e-4), and — in the build container, where the reference tree is
present — to the reference directly on fresh inputs.

Written functionally on the state dict (no nn.Module), one statement per reference statement; every function names the lines it
follows.  float32 throughout, like the reference."""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(sd, name, x):
    """nn.BatchNorm2d / 3d in eval mode: (x - running_mean) / sqrt(running_var + eps) * weight + bias over the channel axis 1"""
    shape = [1, -1] + [1] * (x.dim() - 2)
    return ((x - sd[name + ".running_mean"].view(shape)) / torch.sqrt(sd[name + ".running_var"].view(shape) + BN_EPS)
            * sd[name + ".weight"].view(shape) + sd[name + ".bias"].view(shape))


def _act(cfg, sd, name, x):
    """modeling_resnet.py:46-52, :146-150: PReLU with one slope per channel, or ReLU"""
    if cfg.resnet_relu_type == "prelu":
        return F.prelu(x, sd[name + ".weight"])
    return F.relu(x)


def basic_block(cfg, sd, P, x, stride):
    """modeling_resnet.py:60-75 BasicBlock.forward: conv1 -> bn1 -> relu1 -> conv2 -> bn2 -> (+ downsample(x) or x) -> relu2"""
    out = F.conv2d(x, sd[P + "conv1.weight"], None, stride=stride, padding=1)
    out = _act(cfg, sd, P + "relu1", _bn(sd, P + "bn1", out))
    out = _bn(sd, P + "bn2", F.conv2d(out, sd[P + "conv2.weight"], None, stride=1, padding=1))
    if P + "downsample.0.weight" in sd:          # :12-16 downsample_basic_block: 1 x 1 conv with the block's stride + BatchNorm
        x = _bn(sd, P + "downsample.1", F.conv2d(x, sd[P + "downsample.0.weight"], None, stride=stride))
    return _act(cfg, sd, P + "relu2", out + x)


def video_frontend(cfg, sd, pixel_values):
    """modeling_avhubert.py:52-65 VideoFeatureExtractor + modeling_resnet.py:140-178 ResEncoder: Conv3d(1, 64, (5, 7, 7), stride
    (1, 2, 2), padding (2, 3, 3)) -> BatchNorm3d -> PReLU -> MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)); every frame through the
    ResNet-18 trunk (:99-105, layers [2, 2, 2, 2]) and a global average pool; Linear(512, d).  pixel_values [B][T][1][H][W] -> [B][T][d]"""
    R = "avhubert.feature_extractor_video.resnet."
    B, T = pixel_values.shape[:2]
    x = pixel_values.permute(0, 2, 1, 3, 4)                                # b t c h w -> b c t h w
    x = F.conv3d(x, sd[R + "frontend3D.0.weight"], None, stride=(1, 2, 2), padding=(2, 3, 3))
    x = _act(cfg, sd, R + "frontend3D.2", _bn(sd, R + "frontend3D.1", x))
    x = F.max_pool3d(x, kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
    x = x.transpose(1, 2).reshape(B * T, x.shape[1], x.shape[3], x.shape[4])      # :174-178 threeD_to_2D_tensor
    for layer, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for b in range(2):
            x = basic_block(cfg, sd, R + f"trunk.layer{layer}.{b}.", x, stride if b == 0 else 1)
    x = x.mean(dim=(2, 3)).view(B, T, -1)                                  # AdaptiveAvgPool2d(1)
    return F.linear(x, sd["avhubert.feature_extractor_video.proj.weight"], sd["avhubert.feature_extractor_video.proj.bias"])


def pos_conv_weight(sd, prefix):
    """transformers HubertPositionalConvEmbedding: weight_norm(conv, name="weight", dim=2): w[:, :, k] = g[k] v[:, :, k] / ||v[:, :, k]||"""
    g, v = sd[prefix + "parametrizations.weight.original0"], sd[prefix + "parametrizations.weight.original1"]
    return g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()


def attention(sd, P, heads, x, kv, mask):
    """decoder.py:153-266 AVHubertAttention / transformers HubertAttention with the eager kernel (:121-150): softmax((q k^T)
    * head_dim^-0.5 + mask) v, out_proj.  x [B][Tq][d], kv [B][Tk][d], mask additive [B][1][Tq or 1][Tk] or None"""
    B, Tq, d = x.shape
    hd = d // heads
    q = F.linear(x, sd[P + "q_proj.weight"], sd[P + "q_proj.bias"]).view(B, Tq, heads, hd).transpose(1, 2)
    k = F.linear(kv, sd[P + "k_proj.weight"], sd[P + "k_proj.bias"]).view(B, -1, heads, hd).transpose(1, 2)
    v = F.linear(kv, sd[P + "v_proj.weight"], sd[P + "v_proj.bias"]).view(B, -1, heads, hd).transpose(1, 2)
    w = torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5
    if mask is not None:
        w = w + mask
    o = torch.matmul(F.softmax(w, dim=-1), v).transpose(1, 2).reshape(B, Tq, d)
    return F.linear(o, sd[P + "out_proj.weight"], sd[P + "out_proj.bias"])


def _ln(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def _ffn(sd, P, x):
    """transformers HubertFeedForward: intermediate_dense -> GELU (exact, erf) -> output_dense"""
    return F.linear(F.gelu(F.linear(x, sd[P + "intermediate_dense.weight"], sd[P + "intermediate_dense.bias"])),
                    sd[P + "output_dense.weight"], sd[P + "output_dense.bias"])


def encode(cfg, sd, input_values, pixel_values, padding_mask, taps=None):
    """modeling_avhubert.py:162-213 AVHubertModel.forward + transformers HubertEncoder.forward (post-LayerNorm variant).
    input_values [B][T][104], pixel_values [B][T][1][H][W], padding_mask [B][T] (1 = padding) -> [B][T][d]"""
    A, E = "avhubert.", "avhubert.encoder."
    eps = cfg.layer_norm_eps
    if input_values is None and pixel_values is None:
        raise ValueError("Either `input_values` or `pixel_values` must be passed")                                               # :181
    fa = fv = None
    if input_values is not None:
        fa = F.linear(input_values, sd[A + "feature_extractor_audio.proj.weight"], sd[A + "feature_extractor_audio.proj.bias"]) # :40-47
    if pixel_values is not None:
        fv = video_frontend(cfg, sd, pixel_values)
    if fa is None:
        fa = torch.zeros_like(fv)                                                                                              # :172-177: a missing modality = zero FEATURES
    if fv is None:
        fv = torch.zeros_like(fa)
    feats = torch.cat([fa, fv], dim=2) if cfg.modality_fuse == "concat" else fa + fv                                          # :183-187
    feats = F.layer_norm(feats, (feats.shape[-1],), sd[A + "layer_norm.weight"], sd[A + "layer_norm.bias"], 1e-5)              # :190 (nn.LayerNorm default eps)
    keep = ~padding_mask.bool()                                                                                                # :192-195 (T frames, one mask entry each)
    fused = feats
    if A + "post_extract_proj.weight" in sd:
        feats = F.linear(feats, sd[A + "post_extract_proj.weight"], sd[A + "post_extract_proj.bias"])                          # :197-198
    post = feats
    x = feats * keep.unsqueeze(-1)                                              # HubertEncoder: hidden_states[~mask] = 0
    w = pos_conv_weight(sd, E + "pos_conv_embed.conv.")
    pos = F.conv1d(x.transpose(1, 2), w, sd[E + "pos_conv_embed.conv.bias"], padding=cfg.conv_pos // 2, groups=cfg.conv_pos_groups)
    if cfg.conv_pos % 2 == 0:
        pos = pos[:, :, :-1]                                                    # HubertSamePadLayer
    x = x + F.gelu(pos).transpose(1, 2)
    x = _ln(sd, E + "layer_norm", x, eps)
    if taps is not None:
        taps.update(tap_video=fv[0], tap_audio=fa[0], tap_fused_ln=fused[0], tap_post_proj=post[0], tap_enc_ln=x[0])
    mask = torch.zeros(keep.shape, dtype=torch.float32).masked_fill(~keep, torch.finfo(torch.float32).min)[:, None, None, :]
    mid = cfg.encoder_layers // 2
    for i in range(cfg.encoder_layers):
        P = E + f"layers.{i}."
        x = _ln(sd, P + "layer_norm", x + attention(sd, P + "attention.", cfg.encoder_attention_heads, x, x, mask), eps)       # HubertEncoderLayer
        x = _ln(sd, P + "final_layer_norm", x + _ffn(sd, P + "feed_forward.", x), eps)
        if taps is not None and i in (0, mid):
            taps["tap_layer0" if i == 0 else "tap_layer_mid"] = x[0]
    return x


def decode_logits(cfg, sd, enc, padding_mask, decoder_input_ids):
    """modeling_avhubert.py:276-293 + decoder.py:488-617 AVHubertDecoder.forward (no cache: the reference re-feeds the whole
    prefix every step, :372-391): embed_tokens + sinusoidal positions; per layer self-attention under the causal mask ->
    LayerNorm, cross-attention over the encoder frames under the padding mask -> LayerNorm, FFN -> LayerNorm (:332-369);
    decoder.layer_norm; lm_head.  -> logits [B][L][V]"""
    eps = cfg.layer_norm_eps
    B, L = decoder_input_ids.shape
    x = sd["embed_tokens.weight"][decoder_input_ids] + sd["decoder.pos_embed.position_embeddings"][:L][None]
    neg = torch.finfo(torch.float32).min
    causal = torch.full((L, L), neg).triu(1)[None, None]
    cross = torch.zeros(padding_mask.shape, dtype=torch.float32).masked_fill(padding_mask.bool(), neg)[:, None, None, :]
    for i in range(cfg.decoder_layers):
        P = f"decoder.layers.{i}."
        x = _ln(sd, P + "layer_norm", x + attention(sd, P + "attention.", cfg.decoder_attention_heads, x, x, causal), eps)
        x = _ln(sd, P + "encoder_layer_norm", x + attention(sd, P + "encoder_attn.", cfg.decoder_attention_heads, x, enc, cross), eps)
        x = _ln(sd, P + "final_layer_norm", x + _ffn(sd, P + "feed_forward.", x), eps)
    x = _ln(sd, "decoder.layer_norm", x, eps)
    w = sd["embed_tokens.weight"] if cfg.share_decoder_input_output_embed else sd["lm_head.weight"]
    return F.linear(x, w)


def greedy_generate(cfg, sd, enc, padding_mask, max_new_tokens):
    """[UPSTREAM] transformers GenerationMixin._sample with do_sample=False: the prompt is one bos token; argmax of the last
    position; a sequence that produced eos is padded with pad_token_id from then on; stops when every sequence finished or after
    max_new_tokens.  -> int64 [B][<= 1 + max_new_tokens]"""
    B = enc.shape[0]
    ids = torch.full((B, 1), cfg.bos_token_id, dtype=torch.long)
    unfinished = torch.ones((B,), dtype=torch.bool)
    for _ in range(max_new_tokens):
        nxt = decode_logits(cfg, sd, enc, padding_mask, ids)[:, -1].argmax(dim=-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, cfg.pad_token_id))
        ids = torch.cat([ids, nxt[:, None]], dim=1)
        unfinished = unfinished & (nxt != cfg.eos_token_id)
        if not bool(unfinished.any()):
            break
    return ids


def beam_generate(cfg, sd, enc, padding_mask, num_beams, max_new_tokens, length_penalty=1.0):
    """[UPSTREAM] transformers GenerationMixin._beam_search (the vectorised form of v4.50+), early_stopping False, one eos token:
    per step the 2 K best continuations of the K running beams by accumulated log-probability; those ending in eos or reaching the
    maximum length, if among the K best, compete (score / generated length ** length_penalty) for the K finished slots; the K best
    others run on; the search stops when no running beam can beat the worst finished one (score of the best running beam
    / current length ** length_penalty) or when nothing can continue.  -> (sequences [B][..] int64, scores [B] float32)"""
    B, K, V = enc.shape[0], num_beams, cfg.vocab_size
    max_len = 1 + max_new_tokens
    pad = cfg.pad_token_id
    run_seq = torch.full((B, K, max_len), pad, dtype=torch.long)
    run_seq[:, :, 0] = cfg.bos_token_id
    fin_seq = run_seq.clone()
    run_score = torch.zeros((B, K))
    run_score[:, 1:] = -1e9
    fin_score = torch.full((B, K), -1e9)
    is_fin = torch.zeros((B, K), dtype=torch.bool)
    unsat = torch.ones((B, 1), dtype=torch.bool)
    fin_len = torch.zeros((B, K), dtype=torch.long)
    run_len = 1
    enc_k = enc.repeat_interleave(K, dim=0)
    mask_k = padding_mask.repeat_interleave(K, dim=0)
    top_mask = torch.cat([torch.ones(K, dtype=torch.bool), torch.zeros(K, dtype=torch.bool)])
    gather = lambda t, idx: torch.gather(t, 1, idx.view(B, -1, *([1] * (t.dim() - 2))).expand(-1, -1, *t.shape[2:]))      # noqa: E731
    cur = 1
    while True:
        logits = decode_logits(cfg, sd, enc_k, mask_k, run_seq[:, :, :cur].reshape(B * K, cur))[:, -1]
        logp = F.log_softmax(logits.float(), dim=-1).view(B, K, V) + run_score[:, :, None]
        top_lp, top_idx = torch.topk(logp.view(B, K * V), k=2 * K)
        cand = gather(run_seq, top_idx // V)
        cand[:, :, cur] = top_idx % V
        hits = (cand[:, :, cur] == cfg.eos_token_id) | (cur + 1 >= max_len)
        # running beams of the next step
        lp_run = top_lp + hits.float() * -1e9
        keep = torch.topk(lp_run, k=K)[1]
        run_seq, run_score = gather(cand, keep), gather(lp_run, keep)
        # finished beams
        just = hits & top_mask[None, :]
        lp_fin = top_lp / ((cur + 1 - 1) ** length_penalty)
        lp_fin = lp_fin + (~unsat).float() * -1e9 + (~just).float() * -1e9
        m_seq, m_score = torch.cat([fin_seq, cand], dim=1), torch.cat([fin_score, lp_fin], dim=1)
        m_fin = torch.cat([is_fin, just], dim=1)
        m_len = torch.cat([fin_len, torch.full((B, 2 * K), cur + 1, dtype=torch.long)], dim=1)
        best = torch.topk(m_score, k=K)[1]
        fin_seq, fin_score, is_fin, fin_len = gather(m_seq, best), gather(m_score, best), gather(m_fin, best), gather(m_len, best)
        cur += 1
        best_run = run_score[:, :1] / ((cur - 1) ** length_penalty)
        worst_fin = torch.where(is_fin, fin_score.min(dim=1, keepdim=True)[0], torch.tensor(-1e9))
        unsat = unsat & (best_run > worst_fin).any(dim=-1, keepdim=True)
        if not (bool(unsat.any()) and not bool(hits.all())):
            break
    seqs, scores = fin_seq[:, 0], fin_score[:, 0]
    out_len = int(fin_len[:, 0].max())
    return seqs[:, :out_len], scores
