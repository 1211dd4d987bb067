"""CPU: the flip audit (oracle/audit.py) on the toy geometry.  The "HIP" side is played by the bf16-recipe oracle: its
joint-encoder projection differs from the fp32 oracle's by bf16 noise, exactly the situation the GPU tests and
bench.py audit at the 619M geometry."""
import numpy as np
import torch

from reazonspeech_amd.runtime.config import TINY
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import model as om, greedy as og, audit


def _case(noise=None):
    cfg = TINY
    sd = synthetic_state_dict(cfg, 5, blank_bias=4.0)
    audio, lens = synthetic_batch(4, 2.5, seed=11, ragged=True, min_seconds=1.0)
    padded = np.zeros((4, audio.shape[1] + 16000), np.float32)
    for b in range(4):
        padded[b, 8000:8000 + lens[b]] = audio[b, :lens[b]]
    L = torch.from_numpy(lens + 16000)
    f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(padded), L, "fp32")
    if noise is None:
        f_hip, _ = om.forward_to_joint(cfg, sd, torch.from_numpy(padded), L, "bf16-fused-glu")
    else:
        f_hip = f_ref + noise * torch.randn(f_ref.shape, generator=torch.Generator().manual_seed(3))
    return cfg, sd, f_ref.numpy(), f_hip.numpy(), el.numpy()


def _audit_all(cfg, sd, f_ref, f_hip, el):
    ref = og.rnnt_greedy(cfg, sd, f_ref, el)
    hip = og.rnnt_greedy(cfg, sd, f_hip, el)
    audits = [audit.flip_audit(cfg, sd, f_ref[b], f_hip[b], el[b], hip[b][0], hip[b][1]) for b in range(len(el))]
    return ref, hip, audits


def test_flip_audit_explains_every_difference():
    cfg, sd, f_ref, f_hip, el = _case()
    ref, hip, audits = _audit_all(cfg, sd, f_ref, f_hip, el)
    eq = [hip[b][0] == ref[b][0] for b in range(len(el))]
    s = audit.summarize(audits, eq)
    assert s["walk_reproduces_hip_path"]
    assert s["every_id_difference_starts_at_a_flip"], s
    assert s["decisions"] == sum(int(n) + len(h[0]) - sum(1 for t in set(h[1]) if h[1].count(t) >= cfg.max_symbols)
                                 for n, h in zip(el, hip))
    for a in audits:
        for f in a["flips"]:        # the Lipschitz bound is a theorem: a violation means the walk (or the decode) is wrong
            assert f["margin_ref"] <= f["bound"] * (1 + 1e-9) + 1e-12, f


def test_flip_audit_sees_flips_under_large_noise_and_none_without():
    cfg, sd, f_ref, f_same, el = _case(noise=0.0)
    ref, hip, audits = _audit_all(cfg, sd, f_ref, f_same, el)
    assert all(len(a["flips"]) == 0 for a in audits) and [h[0] for h in hip] == [r[0] for r in ref]
    cfg, sd, f_ref, f_noisy, el = _case(noise=0.5)
    ref, hip, audits = _audit_all(cfg, sd, f_ref, f_noisy, el)
    eq = [hip[b][0] == ref[b][0] for b in range(len(el))]
    s = audit.summarize(audits, eq)
    assert s["local_flips"] > 0 and not all(eq)
    assert s["every_id_difference_starts_at_a_flip"]
    assert s["flip_margin_over_bound_max"] <= 1.0 + 1e-9


def test_batched_audit_equals_the_per_row_walk():
    """oracle/audit.py: flip_audit_batch (torch float64, rows in lockstep — what the GPU tests and bench.py run over all 256
    rows of the benchmark batch) reports what the per-row numpy walk reports: decisions, margins, flips, bounds"""
    cfg, sd, f_ref, f_hip, el = _case(noise=0.05)
    ref, hip, audits = _audit_all(cfg, sd, f_ref, f_hip, el)
    batch = audit.flip_audit_batch(cfg, sd, torch.from_numpy(f_ref), torch.from_numpy(f_hip), el.tolist(),
                                   [h[0] for h in hip], [h[1] for h in hip])
    assert sum(len(a["flips"]) for a in audits) > 0
    for a, b in zip(audits, batch):
        assert a["decisions"] == b["decisions"] and a["path_ok"] == b["path_ok"]
        assert np.allclose(a["margins"], b["margins"], rtol=1e-9, atol=1e-12)
        assert [(x["frame"], x["k_ref"], x["k_hip"]) for x in a["flips"]] == [(x["frame"], x["k_ref"], x["k_hip"]) for x in b["flips"]]
        for x, y in zip(a["flips"], b["flips"]):
            assert abs(x["margin_ref"] - y["margin_ref"]) < 1e-9 and abs(x["bound"] - y["bound"]) < 1e-9


def test_k2_audit_stateless_decoder_one_decision_per_frame():
    """`flip_audit_batch_k2` (Zipformer family: stateless decoder, one decision per frame, blank and <unk> one class): without
    noise no flips and the walk reproduces the hypothesis; under large noise every difference from the reference's ids starts at a
    flip and every flip obeys the Lipschitz bound (tanh joiner)"""
    import torch
    from reazonspeech_amd.runtime.k2_config import ZIPFORMER_TINY as kcfg
    from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
    from oracle import greedy as og
    sd = synthetic_state_dict_k2(kcfg, 7)
    sd["joiner.output_linear.bias"][kcfg.unk_id] += 4.0            # <unk> wins now and then: it must count as "nothing emitted"
    torch.manual_seed(3)
    B, T = 4, 50
    f_ref = torch.randn(B, T, kcfg.joiner_dim)
    lens = [50, 37, 50, 12]
    ref = og.k2_greedy(kcfg, sd, f_ref.numpy(), lens)
    for noise in (0.0, 0.6):
        f_hip = f_ref + noise * torch.randn(B, T, kcfg.joiner_dim)
        hyp = og.k2_greedy(kcfg, sd, f_hip.numpy(), lens)
        au = audit.flip_audit_batch_k2(kcfg, sd, f_ref, f_hip, lens, [h[0] for h in hyp], [h[1] for h in hyp])
        equal = [hyp[b] == ref[b] for b in range(B)]
        s = audit.summarize(au, equal)
        assert s["every_id_difference_starts_at_a_flip"] and s["walk_reproduces_hip_path"] and s["decisions"] == sum(lens)
        for a in au:
            for fl in a["flips"]:
                assert fl["margin_ref"] <= fl["bound"] * (1 + 1e-9) + 1e-12
        if noise == 0.0:
            assert s["local_flips"] == 0 and all(equal)
        else:
            assert s["local_flips"] > 0 and not all(equal)
