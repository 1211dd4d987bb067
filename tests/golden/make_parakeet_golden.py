"""Generate tests/golden/parakeet_tiny.npz — run in the BUILD container only.

Pins the oracle (oracle/model.py, oracle/rnnt_greedy.c) against an independent
implementation of the same architecture that ships in this image:
`transformers.models.parakeet` (ParakeetFeatureExtractor, ParakeetEncoder,
ParakeetForRNNT.generate).  NeMo itself is not installable here, so this
is the strongest anchor available; the fixture stores the *inputs* and the HF outputs so
the comparison can be re-run anywhere without transformers.

    python tests/golden/make_parakeet_golden.py            # toy geometry
    python tests/golden/make_parakeet_golden.py --wide     # the 619M layer geometry, 2 layers, 3 seeds
    python tests/golden/make_parakeet_golden.py --full     # the 619M model itself: 24 layers, the benchmark's weights
"""
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime.config import TINY, WIDE2, ModelConfig   # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict, slaney_mel_filterbank  # noqa: E402

SEED = 7
BLANK_BIAS = 3.9


def _fake(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__version__ = "0.10.0"
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _install_librosa_stub():
    """HF's extractor needs librosa only for the Slaney filterbank (HF feature_extraction
    _parakeet.py:96-98); provide that one function from our closed form, which equals
    transformers.audio_utils.mel_filter_bank to 1e-9."""
    def mel(sr, n_fft, n_mels, fmin, fmax, norm):
        return slaney_mel_filterbank(ModelConfig(sample_rate=sr, n_fft=n_fft, n_mels=n_mels))
    filt = _fake("librosa.filters", mel=mel)
    _fake("librosa", filters=filt)
    if "soxr" not in sys.modules:
        try:
            import soxr  # noqa: F401
        except ImportError:
            _fake("soxr")


def hf_state_dict(cfg, sd):
    out = {}
    for i in (0, 2, 3, 5, 6):
        for p in ("weight", "bias"):
            out[f"encoder.subsampling.layers.{i}.{p}"] = sd[f"encoder.pre_encode.conv.{i}.{p}"]
    for p in ("weight", "bias"):
        out[f"encoder.subsampling.linear.{p}"] = sd[f"encoder.pre_encode.out.{p}"]
    amap = {"q_proj": "linear_q", "k_proj": "linear_k", "v_proj": "linear_v", "o_proj": "linear_out"}
    for i in range(cfg.n_layers):
        L = f"encoder.layers.{i}."
        for n in ("norm_feed_forward1", "norm_self_att", "norm_conv", "norm_feed_forward2", "norm_out",
                  "feed_forward1.linear1", "feed_forward1.linear2",
                  "feed_forward2.linear1", "feed_forward2.linear2",
                  "conv.pointwise_conv1", "conv.depthwise_conv", "conv.pointwise_conv2"):
            for p in ("weight", "bias"):
                out[L + n + "." + p] = sd[L + n + "." + p]
        for hf, ne in amap.items():
            for p in ("weight", "bias"):
                out[L + f"self_attn.{hf}.{p}"] = sd[L + f"self_attn.{ne}.{p}"]
        out[L + "self_attn.relative_k_proj.weight"] = sd[L + "self_attn.linear_pos.weight"]
        out[L + "self_attn.bias_u"] = sd[L + "self_attn.pos_bias_u"]
        out[L + "self_attn.bias_v"] = sd[L + "self_attn.pos_bias_v"]
        for p in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            out[L + "conv.norm." + p] = sd[L + "conv.batch_norm." + p]
    out["encoder_projector.weight"] = sd["joint.enc.weight"]
    out["encoder_projector.bias"] = sd["joint.enc.bias"]
    out["decoder.embedding.weight"] = sd["decoder.prediction.embed.weight"]
    for l in range(cfg.pred_layers):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            out[f"decoder.lstm.{n}_l{l}"] = sd[f"decoder.prediction.dec_rnn.lstm.{n}_l{l}"]
    out["decoder.decoder_projector.weight"] = sd["joint.pred.weight"]
    out["decoder.decoder_projector.bias"] = sd["joint.pred.bias"]
    out["joint.head.weight"] = sd["joint.joint_net.2.weight"]
    out["joint.head.bias"] = sd["joint.joint_net.2.bias"]
    return out


def build_hf_model(cfg, sd):
    from transformers.models.parakeet.configuration_parakeet import ParakeetEncoderConfig, ParakeetRNNTConfig
    from transformers.models.parakeet.modeling_parakeet import ParakeetForRNNT
    enc = ParakeetEncoderConfig(
        hidden_size=cfg.d_model, num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads,
        intermediate_size=cfg.ff_dim, conv_kernel_size=cfg.conv_kernel,
        subsampling_factor=cfg.sub_factor, subsampling_conv_channels=cfg.sub_channels,
        num_mel_bins=cfg.n_mels, scale_input=cfg.xscaling, dropout=0.0, layerdrop=0.0,
        activation_dropout=0.0, attention_dropout=0.0)
    rc = ParakeetRNNTConfig(vocab_size=cfg.n_logits, decoder_hidden_size=cfg.pred_hidden,
                            num_decoder_layers=cfg.pred_layers, max_symbols_per_step=cfg.max_symbols,
                            encoder_config=enc, blank_token_id=cfg.blank_id, pad_token_id=cfg.blank_id)
    model = ParakeetForRNNT(rc).eval()
    missing, unexpected = model.load_state_dict(hf_state_dict(cfg, sd), strict=False)
    missing = [m for m in missing if "inv_freq" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model.generation_config.decoder_start_token_id = cfg.blank_id
    return model


def main():
    _install_librosa_stub()
    from transformers.models.parakeet.feature_extraction_parakeet import ParakeetFeatureExtractor
    cfg = TINY
    sd = synthetic_state_dict(cfg, SEED, blank_bias=BLANK_BIAS)
    rng = np.random.default_rng(SEED)
    lens = [24000, 17717]                        # 1.5 s and ~1.1 s incl. padding
    audio = np.zeros((2, max(lens)), dtype=np.float32)
    for b, L in enumerate(lens):
        t = np.arange(L) / 16000.0
        audio[b, :L] = (0.05 * rng.standard_normal(L) +
                        0.2 * np.sin(2 * np.pi * (220.0 * (b + 1)) * t) * np.sin(2 * np.pi * 3.0 * t)
                        ).astype(np.float32)
    fe = ParakeetFeatureExtractor()
    feats = fe([audio[b, :L] for b, L in enumerate(lens)], sampling_rate=16000, return_tensors="pt")
    model = build_hf_model(cfg, sd)
    with torch.no_grad():
        enc_out = model.get_audio_features(input_features=feats["input_features"],
                                           attention_mask=feats["attention_mask"])
        gen = model.generate(input_features=feats["input_features"],
                             attention_mask=feats["attention_mask"])
    seqs = gen.sequences.numpy()
    durs = gen.durations.numpy()
    ids, frames = [], []
    enc_lens = enc_out.attention_mask.sum(-1).numpy()
    for b in range(2):
        fr = np.cumsum(durs[b])                  # frame index AFTER each step
        i_b, f_b = [], []
        for s in range(1, seqs.shape[1]):
            frame_at_emit = fr[s] - durs[b, s]   # pointer value when the token was produced
            if frame_at_emit >= enc_lens[b]:
                break
            if seqs[b, s] != cfg.blank_id:
                i_b.append(int(seqs[b, s]))
                f_b.append(int(frame_at_emit))
        ids.append(i_b)
        frames.append(f_b)
    umax = max(len(x) for x in ids)
    ids_arr = np.full((2, umax), -1, np.int32)
    frm_arr = np.full((2, umax), -1, np.int32)
    for b in range(2):
        ids_arr[b, :len(ids[b])] = ids[b]
        frm_arr[b, :len(frames[b])] = frames[b]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parakeet_tiny.npz")
    np.savez_compressed(
        out, seed=SEED, blank_bias=BLANK_BIAS, audio=audio, lengths=np.array(lens, np.int64),
        hf_feats=feats["input_features"].numpy().astype(np.float32),
        hf_n_frames=feats["attention_mask"].sum(-1).numpy().astype(np.int64),
        hf_enc=enc_out.last_hidden_state.numpy().astype(np.float32),
        hf_joint_enc=enc_out.pooler_output.numpy().astype(np.float32),
        hf_enc_lens=enc_lens.astype(np.int64), hf_ids=ids_arr, hf_frames=frm_arr,
        hf_n_ids=np.array([len(x) for x in ids], np.int32))
    print("wrote", out, os.path.getsize(out), "bytes; tokens per utt:", [len(x) for x in ids],
          "enc lens", enc_lens)


# ---- the 619M geometry with two layers: d = 1024, 8 heads, C = 256, V + 1 = 3001 ----
WIDE_SEEDS = (7, 8, 9)
# per seed: a two-layer random conformer barely moves the joint, so the blank offset that gives a mixed
# blank / non-blank pattern differs by seed (scanned with the oracle)
WIDE_BLANK_BIAS = {7: 3.25, 8: 8.0, 9: 5.5}
WIDE_LENS = (20000, 14717)          # 1.25 s and ~0.92 s incl. padding -> T' = 16 and 12


def wide_audio(seed):
    """the fixture's inputs are regenerated from the seed (the file stores outputs + a checksum only)"""
    rng = np.random.default_rng(1000 + seed)
    audio = np.zeros((2, max(WIDE_LENS)), dtype=np.float32)
    for b, L in enumerate(WIDE_LENS):
        t = np.arange(L, dtype=np.float64) / 16000.0
        tone = 0.2 * np.sin(2 * np.pi * (180.0 * (b + 1) + 10.0 * seed) * t) * np.sin(2 * np.pi * 3.0 * t)
        audio[b, :L] = (0.05 * rng.standard_normal(L) + tone).astype(np.float32)
    return audio, np.array(WIDE_LENS, np.int64)


def hf_outputs(cfg, sd, audio, lens):
    from transformers.models.parakeet.feature_extraction_parakeet import ParakeetFeatureExtractor
    fe = ParakeetFeatureExtractor()
    feats = fe([audio[b, :L] for b, L in enumerate(lens)], sampling_rate=16000, return_tensors="pt")
    model = build_hf_model(cfg, sd)
    with torch.no_grad():
        enc_out = model.get_audio_features(input_features=feats["input_features"],
                                           attention_mask=feats["attention_mask"])
        gen = model.generate(input_features=feats["input_features"], attention_mask=feats["attention_mask"],
                             max_new_tokens=600)
    seqs, durs = gen.sequences.numpy(), gen.durations.numpy()
    enc_lens = enc_out.attention_mask.sum(-1).numpy()
    ids, frames = [], []
    for b in range(len(lens)):
        fr = np.cumsum(durs[b])
        i_b, f_b = [], []
        for s in range(1, seqs.shape[1]):
            at = fr[s] - durs[b, s]
            if at >= enc_lens[b]:
                break
            if seqs[b, s] != cfg.blank_id:
                i_b.append(int(seqs[b, s]))
                f_b.append(int(at))
        ids.append(i_b)
        frames.append(f_b)
    return feats, enc_out, enc_lens, ids, frames


def main_wide():
    import hashlib
    _install_librosa_stub()
    cfg = WIDE2
    store = {"seeds": np.array(WIDE_SEEDS), "blank_bias": np.array([WIDE_BLANK_BIAS[s] for s in WIDE_SEEDS]), "lengths": np.array(WIDE_LENS, np.int64)}
    for seed in WIDE_SEEDS:
        sd = synthetic_state_dict(cfg, seed, blank_bias=WIDE_BLANK_BIAS[seed])
        audio, lens = wide_audio(seed)
        feats, enc_out, enc_lens, ids, frames = hf_outputs(cfg, sd, audio, lens)
        umax = max(1, max(len(x) for x in ids))
        ids_arr = np.full((2, umax), -1, np.int32)
        frm_arr = np.full((2, umax), -1, np.int32)
        for b in range(2):
            ids_arr[b, :len(ids[b])] = ids[b]
            frm_arr[b, :len(frames[b])] = frames[b]
        k = f"s{seed}_"
        store[k + "audio_sha256"] = np.frombuffer(hashlib.sha256(audio.tobytes()).digest(), np.uint8)
        store[k + "hf_n_frames"] = feats["attention_mask"].sum(-1).numpy().astype(np.int64)
        store[k + "hf_enc"] = enc_out.last_hidden_state.numpy().astype(np.float32)
        store[k + "hf_joint_enc"] = enc_out.pooler_output.numpy().astype(np.float32)
        store[k + "hf_enc_lens"] = enc_lens.astype(np.int64)
        store[k + "hf_ids"], store[k + "hf_frames"] = ids_arr, frm_arr
        store[k + "hf_n_ids"] = np.array([len(x) for x in ids], np.int32)
        print("seed", seed, "tokens", [len(x) for x in ids], "enc lens", enc_lens, flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parakeet_wide.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


# ---- the FULL benchmark model: 24 layers, 619M parameters, the synthetic weights bench.py runs (seed 0) ----
FULL_ROWS = 2


def full_audio():
    """rows 0 and 1 of the benchmark batch (256 x 10 s, seed 1234) with the reference's 0.5 s padding
    (pkg/nemo-asr/src/audio.py:70-83) — the same rows tests/golden/bench_fp32.npz keeps the oracle's joint projection of"""
    from reazonspeech_amd.runtime.synth import synthetic_batch
    audio, lens = synthetic_batch(256, 10.0, seed=1234)
    padded = np.zeros((FULL_ROWS, audio.shape[1] + 16000), np.float32)
    for b in range(FULL_ROWS):
        padded[b, 8000:8000 + int(lens[b])] = audio[b, :int(lens[b])]
    return padded, (lens[:FULL_ROWS] + 16000).astype(np.int64)


def main_full():
    """tests/golden/parakeet_full.npz: transformers' ParakeetForRNNT with ALL 24 layers at the 619M geometry, the
    benchmark's weights and two benchmark utterances — anchors the oracle path that checks the benchmark itself
    (the toy and two-layer fixtures cannot see an error that only accumulates over depth)"""
    import hashlib
    from reazonspeech_amd.runtime.config import FASTCONFORMER_619M
    _install_librosa_stub()
    cfg = FASTCONFORMER_619M
    sd = synthetic_state_dict(cfg, seed=0)
    audio, lens = full_audio()
    feats, enc_out, enc_lens, ids, frames = hf_outputs(cfg, sd, audio, lens)
    umax = max(1, max(len(x) for x in ids))
    ids_arr = np.full((FULL_ROWS, umax), -1, np.int32)
    frm_arr = np.full((FULL_ROWS, umax), -1, np.int32)
    for b in range(FULL_ROWS):
        ids_arr[b, :len(ids[b])] = ids[b]
        frm_arr[b, :len(frames[b])] = frames[b]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parakeet_full.npz")
    np.savez_compressed(
        out, weights_seed=np.int64(0), audio_sha256=np.frombuffer(hashlib.sha256(audio.tobytes()).digest(), np.uint8),
        lengths=lens, hf_n_frames=feats["attention_mask"].sum(-1).numpy().astype(np.int64),
        hf_enc=enc_out.last_hidden_state.numpy().astype(np.float32),
        hf_joint_enc=enc_out.pooler_output.numpy().astype(np.float32), hf_enc_lens=enc_lens.astype(np.int64),
        hf_ids=ids_arr, hf_frames=frm_arr, hf_n_ids=np.array([len(x) for x in ids], np.int32))
    print("wrote", out, os.path.getsize(out), "bytes; tokens per utt:", [len(x) for x in ids], "enc lens", enc_lens)


if __name__ == "__main__":
    if "--full" in sys.argv:
        main_full()
    elif "--wide" in sys.argv:
        main_wide()
    else:
        main()
