"""-m gpu: END-TO-END id parity of `reazonspeech.k2.asr` (pkg/k2-asr/src/transcribe.py:24-45, huggingface.py:73-83).

The reference's default files are the float32 ONNX graphs and onnxruntime computes them in float32; `K2Model(precision="fp32")` /
`load_model(compute="fp32")` does too (rs_set_option "precision_f32": exact-f32 matrix-core GEMMs, float32 activations through
encoder_embed, BiasNorm / bypass / Swoosh, the attention weights, both attention products, the conv modules, down- and
up-sampling).  The checker is the float32 CPU oracle (oracle/zipformer.py with its OWN window / mel banks / position rows) run
end to end, one utterance per call with the reference's 0.9 s of padding: tests/golden/bench_k2_fp32.npz (generator:
tests/golden/make_k2_golden.py) holds its output for EVERY row of the Zipformer benchmark batch (bench.py
`configs.k2_zipformer_159m`: 256 x 10 s, seed 4242, T = 1180 feature frames -> 586 -> 293 output frames).

  float32 mode vs the float32 oracle      fbank fingerprint <= 5e-3 (log energies), joint projection (fingerprint of all rows,
                                          rows 0-1 in full) <= 2e-4; greedy ids AND frames IDENTICAL on every row (near-tie rows
                                          named by the golden may differ at a decision whose margin is below 1e-3)
  bf16 throughput mode                    on full 10 s rows of the batch of 256: taps vs the bf16-recipe oracle (rows 0-1), the
                                          whole batch flip-audited against the float32 mode (oracle/audit.py: flip_audit_batch_k2):
                                          every flip inside the Lipschitz bound, every row without a flip identical to the golden
  batch invariance at B = 256             row 0 alone == row 0 inside the batch, bit for bit, in both modes
PARITY UNPINNED against icefall / sherpa-onnx themselves (neither can run here): every check is against oracle/zipformer.py.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.k2_config import ZIPFORMER_TINY, ZIPFORMER_159M
from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.k2.asr.model import K2Model, synthetic_tokens
from oracle import zipformer as oz, greedy as og, audit

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "bench_k2_fp32.npz")
REPORT = os.path.join(os.path.dirname(HERE), "gpurun_out", "k2_parity.json")
PAD = int(0.9 * 16000)
TOL_F32 = 2e-4          # float32 mode vs float32 oracle (reassociation, the device FFT, v_rsq in BiasNorm)
TOL_FEAT = 5e-3         # fbank log energies (tests/test_gpu_k2.py states the same bar)
TOL_BF16 = 0.08         # bf16 mode vs the bf16-recipe oracle / the float32 mode (tests/test_gpu_k2.py)


def report(key, value):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        try:
            data = json.load(open(REPORT))
        except Exception:
            data = {}
    data[key] = value
    json.dump(data, open(REPORT, "w"), indent=1, sort_keys=True)


def ragged(gold, name, b):
    off = gold["ids_offsets"]
    return gold[name][off[b]:off[b + 1]].tolist()


def run(model, waves, taps=False):
    am, cfg = model.am, model.cfg
    buf = am.stage(waves, buf=am.new_buffers(len(waves), max(len(w) for w in waves)))
    B, t3 = buf.B, cfg.embed_frames(buf.t_max)
    emb = stacks = None
    if taps:
        emb = torch.zeros((B, t3, cfg.encoder_dim[0]), dtype=torch.float32, device=am.device)
        stacks = torch.zeros((B * t3 * sum(cfg.encoder_dim),), dtype=torch.float32, device=am.device)
        am.ctx.set_k2_taps(emb, stacks)
    enc = torch.zeros((B, buf.tp_max, cfg.out_dim), dtype=torch.float32, device=am.device)
    try:
        am.run_device(buf, want_enc=enc)
        torch.cuda.synchronize()
    finally:
        if taps:
            am.ctx.set_k2_taps(None, None)
    outs, off = [], 0
    if taps:
        for d in cfg.encoder_dim:
            outs.append(stacks[off:off + B * t3 * d].view(B, t3, d).cpu())
            off += B * t3 * d
        emb = emb.cpu()
    return buf, emb, outs, enc, am.collect(buf)


def test_tiny_fp32_mode_vs_fp32_oracle(gpu_device):
    """toy geometry, ragged batch: the float32 mode within 2e-4 of the float32 oracle at every tap (encoder_embed, every stack,
    encoder output, joint projection), greedy ids and frames identical to the oracle's own end-to-end search, batch-invariant bits"""
    cfg = ZIPFORMER_TINY
    sd = synthetic_state_dict_k2(cfg, 3)
    model = K2Model(cfg, sd, synthetic_tokens(cfg.vocab_size, 3), device="cuda:0", precision="fp32")
    audio, lens = synthetic_batch(5, 3.0, seed=5, ragged=True, min_seconds=0.7)
    waves = [np.pad(audio[b, :lens[b]], PAD) for b in range(5)]
    buf, emb, stacks, enc, got = run(model, waves, taps=True)
    worst = {}
    for b, w in enumerate(waves):
        taps = {}
        ref = oz.forward(cfg, sd, w, "fp32", taps)
        nf, n = ref["feats"].shape[0], ref["enc"].shape[0]
        t3 = cfg.embed_frames(nf)
        assert got.enc_lens[b] == n
        assert (buf.feats[b, :nf].cpu() - ref["feats"]).abs().max() <= TOL_FEAT
        pairs = [("embed", emb[b, :t3], taps["embed"])] + [(f"S{s}", stacks[s][b, :t3], taps[f"S{s}"]) for s in range(cfg.n_stacks)]
        pairs += [("enc", enc[b, :n].cpu(), ref["enc"]), ("joint", buf.joint_enc[b, :n].cpu(), ref["joint_enc"])]
        for name, a, r in pairs:
            e = float((a - r).abs().max())
            worst[name] = max(worst.get(name, 0.0), e)
            assert e <= TOL_F32, (name, b, e)
        want = oz.greedy_search(cfg, sd, ref["joint_enc"])
        assert (got.ids[b], got.frames[b]) == want, b
    _, _, _, e1, alone = run(model, waves[2:3])
    n = alone.enc_lens[0]
    assert torch.equal(e1[0, :n], enc[2, :n]) and alone.ids[0] == got.ids[2] and alone.frames[0] == got.frames[2]
    print("k2 tiny fp32:", worst)


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def bench_waves(gold):
    audio, lens = synthetic_batch(256, 10.0, seed=int(gold["seed"]))
    assert hashlib.sha256(audio.tobytes()).digest() == bytes(gold["audio_sha256"].tolist()), "inputs drifted from the golden's"
    return [np.pad(audio[b, :lens[b]], PAD) for b in range(256)]


def run_mode(bench_waves, precision):
    cfg = ZIPFORMER_159M
    sd = synthetic_state_dict_k2(cfg, 0)
    model = K2Model(cfg, sd, synthetic_tokens(cfg.vocab_size, 0), device="cuda:0", precision=precision)
    buf, _, _, enc, got = run(model, bench_waves)
    out = [got, buf.joint_enc.clone(), buf.feats.clone()]
    del buf, enc
    b1, _, _, _, alone = run(model, bench_waves[:1])
    out += [alone, b1.joint_enc.clone()]
    del model, b1
    torch.cuda.empty_cache()
    return out


@pytest.fixture(scope="module")
def run32(gpu_device, gold, bench_waves):
    """the float32 mode over the whole benchmark batch -> (DecodedBatch, joint projection, features on the device, row 0 alone)"""
    return run_mode(bench_waves, "fp32")


def test_159m_fp32_mode_every_row_vs_fp32_oracle_golden(gold, run32):
    check_rows(gold, run32, "k2_fp32_mode")


def test_159m_fp32x3_mode_every_row_vs_fp32_oracle_golden(gpu_device, gold, bench_waves):
    """compute="fp32x3": the float32 mode with every float32 product of its GEMMs formed from three bf16 matrix-core terms
    (csrc/k_f32.hip X3) against the SAME golden and assertions, batch invariance included (round 6 measured 256 / 256)"""
    check_rows(gold, run_mode(bench_waves, "fp32x3"), "k2_fp32x3_mode")


def check_rows(gold, run32, key):
    cfg = ZIPFORMER_159M
    rows = int(gold["rows"])
    got, f_dev, feats, alone, f_alone = run32
    assert got.enc_lens[:rows] == gold["enc_lens"].tolist()
    mk = lambda n, dim, seed: (torch.randn((n, dim), generator=torch.Generator().manual_seed(seed), dtype=torch.float32) / n ** 0.5).to(f_dev.device)  # noqa: E731
    R, Rf = mk(cfg.joiner_dim, 8, int(gold["proj_seed"])), mk(cfg.n_mels, 4, int(gold["proj_seed"]) + 1)
    proj, fproj = (f_dev @ R).cpu().numpy(), (feats @ Rf).cpu().numpy()
    worst_proj = worst_feat = worst_f = 0.0
    nf = gold["feat_proj"].shape[1]
    for b in range(rows):
        n = got.enc_lens[b]
        worst_proj = max(worst_proj, float(np.abs(proj[b, :n] - gold["proj"][b, :n]).max()))
        worst_feat = max(worst_feat, float(np.abs(fproj[b, :nf] - gold["feat_proj"][b]).max()))
    for b in range(2):
        n = got.enc_lens[b]
        worst_f = max(worst_f, float((f_dev[b, :n].cpu() - torch.from_numpy(gold["f_rows"][b, :n])).abs().max()))
    assert worst_feat <= TOL_FEAT and worst_proj <= TOL_F32 and worst_f <= TOL_F32, (worst_feat, worst_proj, worst_f)
    g_ids = [ragged(gold, "ids", b) for b in range(rows)]
    g_frames = [ragged(gold, "frames", b) for b in range(rows)]
    near = set(int(b) for b in np.nonzero(gold["min_margin"] < float(gold["near_tie"]))[0])
    differ = [b for b in range(rows) if got.ids[b] != g_ids[b] or got.frames[b] != g_frames[b]]
    assert not [b for b in differ if b not in near], f"rows {differ} differ from the float32 oracle without a near-tie"
    assert all(cfg.unk_id not in x and cfg.blank_id not in x for x in got.ids)
    # batch invariance at B = 256, bits: row 0 alone == row 0 inside the batch
    n0 = alone.enc_lens[0]
    assert n0 == got.enc_lens[0] and torch.equal(f_alone[0, :n0], f_dev[0, :n0]) and alone.ids[0] == got.ids[0] and alone.frames[0] == got.frames[0]
    report(key, {"rows": rows, "ids_and_frames_exact": f"{rows - len(differ)}/{rows}", "near_tie_rows_in_golden": len(near), "differing_rows": differ,
                            "joint_proj_fingerprint_max_err": worst_proj, "joint_enc_rows01_max_err": worst_f, "fbank_fingerprint_max_err": worst_feat,
                            "decisions": int(sum(got.enc_lens[:rows])), "tokens": int(sum(len(x) for x in g_ids)),
                            "golden_min_margin": float(gold["min_margin"].min()), "alone_equals_inside_b256_bits": True})


def test_159m_throughput_mode_full_rows_taps_and_flip_audit_over_all_256_rows(gold, bench_waves, run32):
    """the bf16 throughput mode on the timed batch itself (256 full 10 s rows, T = 1180): (1) rows 0-1 against the bf16-recipe
    oracle at every tap, (2) row 0 alone == inside the batch (bits), (3) all 256 rows audited against the float32 mode's
    projection: joint projection within the stated bf16 tolerance, every local flip inside the Lipschitz bound, every row without
    a flip identical to the float32 oracle golden"""
    cfg = ZIPFORMER_159M
    rows = int(gold["rows"])
    got32, f32_dev, _, _, _ = run32
    sd = synthetic_state_dict_k2(cfg, 0)
    model = K2Model(cfg, sd, synthetic_tokens(cfg.vocab_size, 0), device="cuda:0")
    buf, emb, stacks, enc, got = run(model, bench_waves, taps=True)
    f16 = buf.joint_enc
    assert got.enc_lens == got32.enc_lens
    worst = {}
    for b in range(2):
        taps = {}
        ref = oz.forward(cfg, sd, bench_waves[b], "bf16", taps)
        n, t3 = ref["enc"].shape[0], cfg.embed_frames(ref["feats"].shape[0])
        assert t3 == 586 and n == 293
        pairs = [("embed", emb[b, :t3], taps["embed"])] + [(f"S{s}", stacks[s][b, :t3], taps[f"S{s}"]) for s in range(cfg.n_stacks)]
        pairs += [("enc", enc[b, :n].cpu(), ref["enc"]), ("joint", f16[b, :n].cpu(), ref["joint_enc"])]
        for name, a, r in pairs:
            e = (a - r).abs()
            worst[name] = max(worst.get(name, 0.0), float(e.max()))
            assert e.max() <= TOL_BF16 and e.mean() <= 0.01, (name, b, float(e.max()), float(e.mean()))
    del emb, stacks
    b1, _, _, _, alone = run(model, bench_waves[:1])
    n0 = alone.enc_lens[0]
    assert torch.equal(b1.joint_enc[0, :n0], f16[0, :n0]) and alone.ids[0] == got.ids[0] and alone.frames[0] == got.frames[0]
    same = og.k2_greedy(cfg, sd, f16[:8].cpu().numpy(), np.asarray(got.enc_lens[:8], np.int32))
    assert got.ids[:8] == [r[0] for r in same] and got.frames[:8] == [r[1] for r in same]
    dj = max(float((f16[b, :got.enc_lens[b]] - f32_dev[b, :got.enc_lens[b]]).abs().max()) for b in range(rows))
    assert dj <= TOL_BF16, dj
    audits = audit.flip_audit_batch_k2(cfg, sd, f32_dev[:rows], f16[:rows], got.enc_lens[:rows], got.ids[:rows], got.frames[:rows], device=f32_dev.device)
    g_ids = [ragged(gold, "ids", b) for b in range(rows)]
    g_frames = [ragged(gold, "frames", b) for b in range(rows)]
    equal = [got.ids[b] == g_ids[b] and got.frames[b] == g_frames[b] for b in range(rows)]
    for a in audits:
        for fl in a["flips"]:
            assert fl["margin_ref"] <= fl["bound"] * (1 + 1e-9) + 1e-12, fl
            assert fl["delta_f"] <= TOL_BF16 * cfg.joiner_dim ** 0.5
    summary = audit.summarize(audits, equal)
    assert summary["every_id_difference_starts_at_a_flip"] and summary["walk_reproduces_hip_path"], summary
    n_tok = sum(len(x) for x in g_ids)
    agree = sum(sum(1 for x, y in zip(got.ids[b], g_ids[b]) if x == y) for b in range(rows)) / max(n_tok, 1)
    summary.update({"rows": rows, "rows_identical_to_fp32_oracle": int(sum(equal)), "joint_enc_max_err_vs_fp32_mode": dj,
                    "token_agreement_positional": agree, "taps_rows01_vs_bf16_recipe_oracle_max_err": worst,
                    "alone_equals_inside_b256_bits": True})
    report("k2_bf16_audit_all_rows", summary)
    del model, buf
    torch.cuda.empty_cache()
