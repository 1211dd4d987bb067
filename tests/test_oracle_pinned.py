"""CPU: the oracle against the committed golden fixture generated from the independent
`transformers.models.parakeet` implementation (tests/golden/make_parakeet_golden.py), and the C
greedy restatement against torch's own LSTM / greedy loop."""
import math
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.config import TINY
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import model as om, greedy as og

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parakeet_tiny.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def sd(gold):
    return synthetic_state_dict(TINY, int(gold["seed"]), blank_bias=float(gold["blank_bias"]))


def test_frontend_vs_hf(gold, sd):
    feats, n = om.frontend(TINY, sd, torch.from_numpy(gold["audio"]), torch.from_numpy(gold["lengths"]))
    assert n.tolist() == gold["hf_n_frames"].tolist()
    hf = torch.from_numpy(gold["hf_feats"])[:, :feats.shape[1]]
    assert (feats - hf).abs().max() <= 1e-4


def test_encoder_and_joint_projection_vs_hf(gold, sd):
    taps = {}
    f, el = om.forward_to_joint(TINY, sd, torch.from_numpy(gold["audio"]), torch.from_numpy(gold["lengths"]),
                                "fp32", taps)
    assert el.tolist() == gold["hf_enc_lens"].tolist()
    for b in range(2):
        n = int(el[b])
        assert (taps["enc"][b, :n] - torch.from_numpy(gold["hf_enc"])[b, :n]).abs().max() <= 1e-4
        assert (f[b, :n] - torch.from_numpy(gold["hf_joint_enc"])[b, :n]).abs().max() <= 1e-4


def test_greedy_c_vs_hf_and_torch(gold, sd):
    f, el = om.forward_to_joint(TINY, sd, torch.from_numpy(gold["audio"]), torch.from_numpy(gold["lengths"]), "fp32")
    c_out = og.rnnt_greedy(TINY, sd, f.numpy(), el.numpy())
    t_out = om.greedy_torch(TINY, sd, f, el)
    assert c_out == t_out
    for b in range(2):
        n = int(gold["hf_n_ids"][b])
        assert c_out[b][0] == gold["hf_ids"][b, :n].tolist()
        assert c_out[b][1] == gold["hf_frames"][b, :n].tolist()


def test_greedy_c_vs_torch_random_inputs():
    """diverse emissions: random joint-encoder tensors, ragged lengths, an empty utterance"""
    sd = synthetic_state_dict(TINY, 11, blank_bias=4.0)
    g = torch.Generator().manual_seed(2)
    B, Tp = 9, 30
    f = torch.randn((B, Tp, TINY.joint_hidden), generator=g) * 1.5
    lens = torch.randint(1, Tp + 1, (B,), generator=g)
    lens[2] = 0
    c_out = og.rnnt_greedy(TINY, sd, f.numpy(), lens.numpy())
    t_out = om.greedy_torch(TINY, sd, f, lens)
    assert c_out == t_out
    assert c_out[2] == ([], [])
    assert sum(len(i) for i, _ in c_out) > 20
    runs = max(sum(1 for x in fr if x == t) for _, fr in c_out for t in set(fr))
    assert runs <= TINY.max_symbols


def test_greedy_overflow_reported():
    sd = synthetic_state_dict(TINY, 11, blank_bias=-5.0)     # emits max_symbols on every frame
    f = torch.zeros((1, 4, TINY.joint_hidden))
    with pytest.raises(RuntimeError):
        og.rnnt_greedy(TINY, sd, f.numpy(), np.array([4]), u_max=7)
    out = og.rnnt_greedy(TINY, sd, f.numpy(), np.array([4]))
    assert len(out[0][0]) == 4 * TINY.max_symbols


def test_exact_math_functions():
    L = og.lib()
    for x in np.linspace(-30, 30, 601):
        x = float(np.float32(x))
        assert abs(L.rs_oracle_expf(x) - math.exp(x)) <= 2e-7 * math.exp(x)
        assert abs(L.rs_oracle_sigmoidf(x) - 1 / (1 + math.exp(-x))) <= 2e-7
        assert abs(L.rs_oracle_tanhf(x) - math.tanh(x)) <= 3e-7
    assert L.rs_oracle_expf(-1000.0) > 0 and math.isfinite(L.rs_oracle_expf(1000.0))


def test_bf16_recipe_stays_close_to_fp32(gold, sd):
    a, l = torch.from_numpy(gold["audio"]), torch.from_numpy(gold["lengths"])
    f32, el = om.forward_to_joint(TINY, sd, a, l, "fp32")
    f16, _ = om.forward_to_joint(TINY, sd, a, l, "bf16")
    f16g, _ = om.forward_to_joint(TINY, sd, a, l, "bf16-fused-glu")      # GLU rounded after, not before
    for b in range(2):
        n = int(el[b])
        assert (f32[b, :n] - f16[b, :n]).abs().max() <= 0.1
        assert (f32[b, :n] - f16g[b, :n]).abs().max() <= 0.1
        assert 0 < (f16[b, :n] - f16g[b, :n]).abs().max() <= 0.05


def test_attention_window_predicate():
    cfg = TINY.with_(att_left=2, att_right=1, n_global=1)
    allowed = om.attention_allowed(cfg, 6, torch.tensor([6, 4]))
    a = allowed[0]
    assert a[3, 1] and a[3, 4] and not a[3, 5] and not a[4, 1]      # |i-j| window
    assert a[5, 0] and a[0, 5]                                        # global token 0
    assert not allowed[1][:, 4:].any() and not allowed[1][4:, :].any()  # padding


# ---- the 619M geometry (d = 1024, 8 heads, C = 256, V + 1 = 3001, 640-wide LSTM / joint) with two layers ----
WIDE_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parakeet_wide.npz")


def wide_case(gold, seed):
    """inputs of one seed of tests/golden/parakeet_wide.npz, regenerated and checksum-verified"""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_parakeet_golden import wide_audio
    from reazonspeech_amd.runtime.config import WIDE2
    audio, lens = wide_audio(seed)
    want = bytes(gold[f"s{seed}_audio_sha256"].tolist())
    assert hashlib.sha256(audio.tobytes()).digest() == want, "fixture inputs drifted from the generator"
    bias = float(gold["blank_bias"][list(gold["seeds"]).index(seed)])
    return WIDE2, synthetic_state_dict(WIDE2, seed, blank_bias=bias), audio, lens


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_wide_geometry_vs_hf(seed):
    """oracle (fp32) == transformers.models.parakeet at the benchmark model's widths, three seeds:
    encoder output and joint projection within 2e-4, greedy ids and emission frames identical"""
    gold = np.load(WIDE_GOLD)
    cfg, sd, audio, lens = wide_case(gold, seed)
    taps = {}
    f, el = om.forward_to_joint(cfg, sd, torch.from_numpy(audio), torch.from_numpy(lens), "fp32", taps)
    k = f"s{seed}_"
    assert taps["n_frames"].tolist() == gold[k + "hf_n_frames"].tolist()
    assert el.tolist() == gold[k + "hf_enc_lens"].tolist()
    for b in range(2):
        n = int(el[b])
        assert (taps["enc"][b, :n] - torch.from_numpy(gold[k + "hf_enc"])[b, :n]).abs().max() <= 2e-4
        assert (f[b, :n] - torch.from_numpy(gold[k + "hf_joint_enc"])[b, :n]).abs().max() <= 2e-4
    out = og.rnnt_greedy(cfg, sd, f.numpy(), el.numpy())
    for b in range(2):
        n = int(gold[k + "hf_n_ids"][b])
        assert out[b][0] == gold[k + "hf_ids"][b, :n].tolist()
        assert out[b][1] == gold[k + "hf_frames"][b, :n].tolist()


# ---- the 619M model itself: 24 layers, the benchmark's weights, benchmark utterances --------------------------------
def test_full_depth_oracle_vs_hf_parakeet_24_layers():
    """The oracle path that checks the BENCHMARK (tests/golden/bench_fp32.npz: the float32 oracle end to end on all 256 rows
    of the benchmark batch, 619M geometry, seed-0 weights) against transformers' ParakeetForRNNT with all 24 layers and the
    same weights on benchmark rows 0 and 1 (tests/golden/parakeet_full.npz, make_parakeet_golden.py --full): joint
    projection within 2e-4 over the whole utterance, greedy ids and emission frames identical.  The toy and two-layer
    fixtures above cannot see an error that only accumulates over depth; this one can."""
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, "golden", "bench_fp32.npz"))
    hf = np.load(os.path.join(here, "golden", "parakeet_full.npz"))
    off = gold["equal_offsets"]
    worst = 0.0
    for b in range(2):
        n = int(hf["hf_enc_lens"][b])
        assert n == int(gold["equal_enc_lens"][b]) == 138
        worst = max(worst, float(np.abs(gold["equal_f_rows"][b, :n] - hf["hf_joint_enc"][b, :n]).max()))
        k = int(hf["hf_n_ids"][b])
        assert gold["equal_ids"][off[b]:off[b + 1]].tolist() == hf["hf_ids"][b, :k].tolist()
        assert gold["equal_frames"][off[b]:off[b + 1]].tolist() == hf["hf_frames"][b, :k].tolist()
    assert worst <= 2e-4, worst
    assert int(hf["hf_n_ids"].sum()) > 100


def test_full_depth_oracle_reproduces_its_own_golden_row():
    """the committed golden is what the oracle computes today: row 1 of the benchmark batch re-run here (one utterance, ~3 s
    of CPU) — joint projection bit-for-bit up to float32 reassociation across thread counts (1e-5), ids and frames identical"""
    from reazonspeech_amd.runtime.config import FASTCONFORMER_619M
    from reazonspeech_amd.runtime.synth import synthetic_batch
    here = os.path.dirname(os.path.abspath(__file__))
    gold = np.load(os.path.join(here, "golden", "bench_fp32.npz"))
    cfg = FASTCONFORMER_619M
    sd = synthetic_state_dict(cfg, seed=0)
    audio, lens = synthetic_batch(256, 10.0, seed=1234)
    b = 1
    wav = np.pad(audio[b, :int(lens[b])], 8000)
    f, el = om.forward_to_joint(cfg, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32")
    n = int(el[0])
    assert np.abs(f[0, :n].numpy() - gold["equal_f_rows"][b, :n]).max() <= 1e-5
    hyp = og.rnnt_greedy(cfg, sd, f.numpy(), el.numpy())[0]
    off = gold["equal_offsets"]
    assert hyp[0] == gold["equal_ids"][off[b]:off[b + 1]].tolist() and hyp[1] == gold["equal_frames"][off[b]:off[b + 1]].tolist()
