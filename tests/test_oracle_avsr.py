"""CPU: the AV-HuBERT family's checker chain (SURVEY.md §8f row 4 avsr; BASELINE.json configs[4]).

    reference itself (pkg/avsr/src/avhubert/*.py, imported unchanged)  --make_avsr_golden.py-->  tests/golden/avsr_ref_{tiny,base}.npz
    oracle/avsr.py (CPU restatement)  ==  the goldens                                     (this file; runs anywhere)
    oracle/avsr.py  ==  the reference on FRESH inputs                                      (this file; build container only)
    HIP path  ==  the goldens and the oracle                                               (tests/test_gpu_avsr.py, -m gpu)

Tolerances (float32 on both sides, different summation orders): encoder taps and output 2e-4, logits 5e-4; greedy and beam-search
ids identical; beam scores 1e-4.  Host pieces: the feature extractor's log filterbank / stacking / padding against the published
python_speech_features algorithm on analytic cases."""
import hashlib
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.avsr_config import AVSR_TINY, AVSR_BASE
from reazonspeech_amd.runtime.avsr_synth import synthetic_clips
from reazonspeech_amd.runtime.avsr_weights import synthetic_state_dict_avsr, prepare_weights_avsr, expected_shapes_avsr, sinusoidal_positions
from oracle import avsr as oa, _ref_avsr as ra

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {"tiny": AVSR_TINY, "base": AVSR_BASE}


def load(name):
    g = np.load(os.path.join(HERE, "golden", f"avsr_ref_{name}.npz"))
    cfg = CONFIGS[name]
    B, T = int(g["clips"]), int(g["frames"])
    a, v, mask, lens = synthetic_clips(B, T, seed=int(g["input_seed"]), ragged=True, min_frames=max(8, T // 3))
    assert hashlib.sha256(a.tobytes() + v.tobytes() + mask.tobytes()).digest() == bytes(g["input_sha256"].tolist()), "inputs drifted from the golden's"
    return g, cfg, synthetic_state_dict_avsr(cfg, int(g["weight_seed"])), a, v, mask


def test_oracle_equals_the_reference_golden_tiny():
    g, cfg, sd, a, v, mask = load("tiny")
    with torch.no_grad():
        taps = {}
        enc = oa.encode(cfg, sd, torch.from_numpy(a), torch.from_numpy(v), torch.from_numpy(mask), taps)
        assert (enc - torch.from_numpy(g["enc"])).abs().max() <= 2e-4
        for k, t in taps.items():
            assert (t - torch.from_numpy(g[k])).abs().max() <= 2e-4, k
        R = torch.randn((cfg.encoder_embed_dim, 8), generator=torch.Generator().manual_seed(int(g["proj_seed"]))) / cfg.encoder_embed_dim ** 0.5
        assert ((enc @ R) - torch.from_numpy(g["enc_proj"])).abs().max() <= 2e-4
        m = torch.from_numpy(mask)
        # one modality only (modeling_avhubert.py:172-177: zero features in the other's place)
        assert (oa.encode(cfg, sd, torch.from_numpy(a), None, m) - torch.from_numpy(g["enc_audio_only"])).abs().max() <= 2e-4
        assert (oa.encode(cfg, sd, None, torch.from_numpy(v), m) - torch.from_numpy(g["enc_video_only"])).abs().max() <= 2e-4
        greedy = oa.greedy_generate(cfg, sd, enc, m, int(g["new_tokens"]))
        assert np.array_equal(greedy.numpy(), g["greedy"])
        logits = oa.decode_logits(cfg, sd, enc, m, torch.from_numpy(g["greedy"][:, :-1]).long())
        assert (logits[:2] - torch.from_numpy(g["logits"])).abs().max() <= 5e-4
        k = int(g["beam_clips"])
        seq, sc = oa.beam_generate(cfg, sd, enc[:k], m[:k], int(g["beams"]), int(g["new_tokens"]))
        assert np.array_equal(seq.numpy(), g["beam"]) and (sc - torch.from_numpy(g["beam_scores"])).abs().max() <= 1e-4


def test_oracle_equals_the_reference_golden_base_encoder_clip0():
    """the 161M geometry (12 x 768 encoder, ResNet-18 front-end at 88 x 88): clip 0 of the golden batch through the oracle's encoder
    alone (a clip's frames only attend to its own frames; the Conv3d sees its own frames) against the reference's taps and output"""
    g, cfg, sd, a, v, mask = load("base")
    with torch.no_grad():
        taps = {}
        enc = oa.encode(cfg, sd, torch.from_numpy(a[:1]), torch.from_numpy(v[:1]), torch.from_numpy(mask[:1]), taps)
    assert (enc[0] - torch.from_numpy(g["enc"][0])).abs().max() <= 2e-4
    for k, t in taps.items():
        assert (t - torch.from_numpy(g[k])).abs().max() <= 2e-4, k


@pytest.mark.skipif(not ra.available(), reason="the reference tree is only present in the build container")
def test_oracle_equals_the_reference_on_fresh_inputs():
    cfg = AVSR_TINY.with_(encoder_layers=3, decoder_layers=1, vocab_size=40)
    sd = synthetic_state_dict_avsr(cfg, 17)
    model = ra.build(cfg, sd)
    a, v, mask, _ = synthetic_clips(3, 17, seed=99, ragged=True)
    kw = dict(input_values=torch.from_numpy(a), pixel_values=torch.from_numpy(v), padding_mask=torch.from_numpy(mask))
    with torch.no_grad():
        ref = model.avhubert(**kw).last_hidden_state
        enc = oa.encode(cfg, sd, kw["input_values"], kw["pixel_values"], kw["padding_mask"])
        assert (enc - ref).abs().max() <= 2e-4
        ids = torch.tensor([[0, 5, 7, 9], [0, 3, 3, 2], [0, 11, 2, 1]])
        want = model(**kw, decoder_input_ids=ids, decoder_attention_mask=torch.ones_like(ids)).logits
        assert (oa.decode_logits(cfg, sd, enc, kw["padding_mask"], ids) - want).abs().max() <= 5e-4
        for beams in (1, 4):
            out = model.generate(**kw, num_beams=beams, do_sample=False, max_new_tokens=7, use_cache=False)
            mine = oa.greedy_generate(cfg, sd, enc, kw["padding_mask"], 7) if beams == 1 else oa.beam_generate(cfg, sd, enc, kw["padding_mask"], beams, 7)[0]
            assert np.array_equal(out.numpy(), mine.numpy()), beams


def test_weights_round_trip_and_layouts():
    cfg = AVSR_TINY
    sd = synthetic_state_dict_avsr(cfg, 1)
    assert set(expected_shapes_avsr(cfg)) == set(sd) and abs(AVSR_BASE.n_params() / 1e6 - 160.9) < 0.2
    w = prepare_weights_avsr(cfg, sd)
    assert w["fe.audio.w"].shape == (cfg.encoder_embed_dim, 128) and torch.all(w["fe.audio.w"][:, 104:] == 0)
    assert w["v.conv3d.w"].shape == (245, 64) and w["v.l2.0.conv1.w"].shape == (128, 9 * 64) and w["v.l2.0.ds.w"].shape == (128, 64)
    # the positional convolution's effective kernel equals torch's own weight_norm parametrisation
    conv = torch.nn.Conv1d(cfg.encoder_embed_dim, cfg.encoder_embed_dim, cfg.conv_pos, padding=cfg.conv_pos // 2, groups=cfg.conv_pos_groups)
    conv = torch.nn.utils.parametrizations.weight_norm(conv, name="weight", dim=2)
    conv.load_state_dict({k.split("conv.", 1)[1]: v for k, v in sd.items() if "pos_conv_embed.conv." in k})
    cg, G = cfg.encoder_embed_dim // cfg.conv_pos_groups, cfg.conv_pos_groups
    assert torch.allclose(w["enc.pos.w"], conv.weight.detach().reshape(G, cg, cg, cfg.conv_pos).permute(0, 3, 2, 1), atol=1e-6)
    assert torch.allclose(oa.pos_conv_weight(sd, "avhubert.encoder.pos_conv_embed.conv."), conv.weight.detach(), atol=1e-6)
    # BatchNorm folded the way torch's inference kernel does: alpha = w / sqrt(var + eps), beta = b - mean * alpha
    P = "avhubert.feature_extractor_video.resnet.frontend3D.1."
    x = torch.randn(4, 64)
    want = torch.nn.functional.batch_norm(x, sd[P + "running_mean"], sd[P + "running_var"], sd[P + "weight"], sd[P + "bias"], False, 0.0, 1e-5)
    assert torch.allclose(x * w["v.bn0.alpha"] + w["v.bn0.beta"], want, atol=1e-6)
    p = sinusoidal_positions(16, 8)
    assert p[0, 0] == 0 and p[0, 1] == 1 and abs(float(p[3, 2]) - np.sin(3 / 10000 ** (2 / 8))) < 1e-6
    bad = dict(sd)
    bad["avhubert.extra.weight"] = torch.zeros(1)
    with pytest.raises(ValueError, match="no counterpart"):
        prepare_weights_avsr(cfg, bad)


def test_feature_extractor_follows_python_speech_features_and_the_reference_batching():
    from reazonspeech_amd.avsr.feature_extraction import AVHubertFeatureExtractor, logfbank
    # a pure 1 kHz tone: frames of 400 samples every 160; the filter around 1 kHz carries the energy; zero signal -> log(eps)
    t = np.arange(16000) / 16000.0
    fb = logfbank(np.sin(2 * np.pi * 1000.0 * t))
    assert fb.shape == (99, 26) and 8 <= int(fb[10].argmax()) <= 12
    assert np.allclose(logfbank(np.zeros(1600)), np.log(np.finfo(float).eps))
    assert logfbank(np.zeros(400)).shape[0] == 1 and logfbank(np.zeros(401)).shape[0] == 2      # ceil((n - 400) / 160) + 1
    ex = AVHubertFeatureExtractor()
    rng = np.random.default_rng(0)
    audio = [rng.standard_normal(16000).astype(np.float32) * 0.1, rng.standard_normal(9000).astype(np.float32) * 0.1]
    video = [rng.integers(0, 255, size=(25, 96, 96), dtype=np.uint8), rng.integers(0, 255, size=(14, 96, 96), dtype=np.uint8)]
    out = ex(raw_audio=audio, raw_video=video)
    T = 25                                                   # 99 fbank frames -> 100 after padding to a multiple of 4 -> 25 stacked
    assert out["input_values"].shape == (2, T, 104) and out["pixel_values"].shape == (2, T, 1, 88, 88) and out["padding_mask"].shape == (2, T)
    n2 = int(np.ceil((1 + np.ceil((9000 - 400) / 160)) / 4))
    assert out["padding_mask"][0].sum() == 0 and out["padding_mask"][1].sum() == T - n2
    assert np.allclose(out["input_values"][0].mean(-1), 0, atol=1e-5) and np.allclose(out["input_values"][1, n2:], 0)
    assert np.allclose(out["pixel_values"][1, n2:], (0 - 0.421) / 0.165)                       # padded frames: a black image, normalised
    crop = video[0][:, 4:92, 4:92].astype(np.float32) / 255.0
    assert np.allclose(out["pixel_values"][0, :, 0], (crop - 0.421) / 0.165, atol=1e-6)
    with pytest.raises(RuntimeError, match="mediapipe"):
        ex(raw_audio=audio[0], raw_video=video[0], extract_mouth=True)
