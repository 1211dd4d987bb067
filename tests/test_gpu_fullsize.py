"""-m gpu: parity at the BENCHMARK geometry (d = 1024, 8 heads, C = 256, FFN 4096, V + 1 = 3001, 640-wide LSTM and
joint) — the kernel paths the 2-layer toy configuration never reaches:

  decode   rs_rnnt_greedy vs oracle/rnnt_greedy.c, token ids AND emission frames BIT-EXACT: K slices of five
           16-blocks (the register-prefetch loops), 47 column tiles with the cross-tile argmax merge, the
           ragged last tile (3001 = 46 * 64 + 57, blank id 3000 lives in it), a forced exact tie between two
           column tiles (lowest index must win), a batch that is not a multiple of the 32-row tile
  encoder  619M (24 layers) vs the bf16-recipe oracle with per-stage taps (subsampling output = rows S1-S5 at
           C = 256; layers 0 / 11 / 23 = LayerNorm<4>, the fused norm pair, 8-head attention, the big-tile GEMMs)
           and the WIDE2 (2-layer) HF golden, three seeds

Stated tolerances (LayerNorm-ed O(1) activations, bf16 GEMM operands / stored activations, f32 accumulation):
  subsampling output (x sqrt(d) scaled, |x| ~ 30)   max |err| <= 0.2,  mean <= 0.015   (measured 0.061 / 0.0042)
  layer outputs, 24-layer encoder output            max |err| <= TOL_MAX, mean <= TOL_MEAN (below)
  joint encoder projection                          same class
The measured values of every run are written to gpurun_out/parity_fullsize.json.
"""
import json
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.config import FASTCONFORMER_619M, WIDE2
from reazonspeech_amd.runtime.model import AsrModel
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import model as om, greedy as og
from test_oracle_pinned import WIDE_GOLD, wide_case

pytestmark = pytest.mark.gpu

# 24 layers of bf16-operand GEMMs against the oracle that rounds at the same points: the two sides differ
# by f32 accumulation order and fast exp / rcp, which the next bf16 rounding amplifies to one bf16 ulp
# (2^-8 relative) per flipped rounding; LayerNorm after every layer keeps the error from compounding.
TOL_MAX, TOL_MEAN = 0.06, 0.008      # measured on MI355X: max 0.020, mean 0.0029 (profiles/r02a_parity_fullsize.json)
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_fullsize.json")


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


def run_greedy(model, f, lens, u_max=None):
    B, Tp, _ = f.shape
    u_max = u_max or Tp * model.cfg.max_symbols
    dev = model.device
    ids = torch.zeros((B, u_max), dtype=torch.int32, device=dev)
    frames = torch.zeros_like(ids)
    n_ids = torch.zeros((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((model.ctx.workspace_bytes(B, 16000),), dtype=torch.uint8, device=dev)
    model.ctx.rnnt_greedy(f.to(dev), lens.to(dev), B, Tp, u_max, ids, frames, n_ids, ws,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = n_ids.cpu().numpy()
    return [(ids[b, :n[b]].cpu().tolist(), frames[b, :n[b]].cpu().tolist()) for b in range(B)]


@pytest.fixture(scope="module")
def wide(gpu_device):
    """the 619M decoder / joint (same tensor names and seeds -> the same weights as the 24-layer model) behind a
    2-layer encoder, so the fixture builds in seconds"""
    sd = synthetic_state_dict(WIDE2, 0, blank_bias=7.5)
    return AsrModel(WIDE2, sd, SyntheticTokenizer(WIDE2.vocab_size), device="cuda:0"), sd


def test_decode_bit_exact_at_619m_geometry(wide):
    model, sd = wide
    cfg = model.cfg
    assert (cfg.pred_hidden, cfg.joint_hidden, cfg.n_logits) == (640, 640, 3001)
    g = torch.Generator().manual_seed(2)
    B, Tp = 37, 45
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * (0.8 + 0.4 * torch.rand((B, 1, 1), generator=g))
    lens = torch.randint(1, Tp + 1, (B,), generator=g, dtype=torch.int32)
    lens[3] = 0
    lens[5] = Tp
    got = run_greedy(model, f, lens)
    ref = og.rnnt_greedy(cfg, sd, f.numpy(), lens.numpy())
    n_tok = sum(len(r[0]) for r in ref)
    report("decode_619m", {"tokens": n_tok, "distinct": len({t for r in ref for t in r[0]}),
                           "max_per_utt": max(len(r[0]) for r in ref)})
    assert n_tok > 300 and len({t for r in ref for t in r[0]}) > 40, "fixture should emit a varied token stream"
    for b in range(B):
        assert got[b][0] == ref[b][0], f"ids differ for utterance {b}"
        assert got[b][1] == ref[b][1], f"frames differ for utterance {b}"


def test_decode_tie_across_column_tiles_picks_lowest_index(gpu_device):
    """two identical joint rows in DIFFERENT 64-column tiles (ids 100 and 2900: tiles 1 and 45) produce bit-identical
    logits; whenever they are the maximum the merge across tiles must return the lower id, as torch.argmax and the
    oracle do.  A third copy inside the ragged last tile (id 2999) ties as well."""
    cfg = WIDE2
    sd = synthetic_state_dict(cfg, 0, blank_bias=7.5)
    W, b = sd["joint.joint_net.2.weight"], sd["joint.joint_net.2.bias"]
    for dup in (2900, 2999):
        W[dup] = W[100]
        b[dup] = b[100]
    b[[100, 2900, 2999]] += 3.5                                     # make the tied triple win often
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    g = torch.Generator().manual_seed(5)
    B, Tp = 9, 30
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g)
    lens = torch.randint(8, Tp + 1, (B,), generator=g, dtype=torch.int32)
    got = run_greedy(model, f, lens)
    ref = og.rnnt_greedy(cfg, sd, f.numpy(), lens.numpy())
    toks = [t for r in ref for t in r[0]]
    assert toks.count(100) >= 5, "the tie must actually decide emissions in this fixture"
    assert 2900 not in toks and 2999 not in toks
    assert got == ref


def test_decode_overflow_and_small_umax(wide):
    """u_max smaller than the emission count -> RS_EOVERFLOW, like the oracle's -5"""
    from reazonspeech_amd.runtime import capi
    model, sd = wide
    g = torch.Generator().manual_seed(9)
    f = torch.randn((2, 20, model.cfg.joint_hidden), generator=g) * 1.5     # emits max_symbols on most frames
    lens = torch.tensor([20, 20], dtype=torch.int32)
    with pytest.raises(capi.RsError) as e:
        run_greedy(model, f, lens, u_max=8)
    assert e.value.code == capi.RS_EOVERFLOW


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_wide_geometry_vs_hf_golden(gpu_device, seed):
    """d = 1024 / 8 heads / C = 256 / V + 1 = 3001 through the whole HIP path against the HF outputs of
    tests/golden/parakeet_wide.npz: encoder output within 0.08 of the fp32 HF encoder (measured 0.019 - 0.024), decode bit-exact against
    the C oracle on the HIP joint projection; agreement with the HF token ids is recorded."""
    gold = np.load(WIDE_GOLD)
    cfg, sd, audio, lens = wide_case(gold, seed)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0", pad_seconds=0.0)
    buf = model.stage([audio[b, :int(lens[b])] for b in range(2)])
    enc = torch.zeros((2, buf.tp_max, cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    k = f"s{seed}_"
    assert buf.n_frames.cpu().tolist() == gold[k + "hf_n_frames"].tolist()
    el = buf.enc_lens.cpu().tolist()
    assert el == gold[k + "hf_enc_lens"].tolist()
    worst = 0.0
    for b in range(2):
        d = (enc.cpu()[b, :el[b]] - torch.from_numpy(gold[k + "hf_enc"])[b, :el[b]]).abs()
        dj = (buf.joint_enc.cpu()[b, :el[b]] - torch.from_numpy(gold[k + "hf_joint_enc"])[b, :el[b]]).abs()
        worst = max(worst, d.max().item(), dj.max().item())
        assert d.max() <= 0.08 and dj.max() <= 0.08, (b, d.max().item(), dj.max().item())
    got = model.collect(buf)
    ref = og.rnnt_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref]
    hf_ids = [gold[k + "hf_ids"][b, :gold[k + "hf_n_ids"][b]].tolist() for b in range(2)]
    report(f"wide_hf_seed{seed}", {"enc_max_err_vs_hf_fp32": worst, "ids_equal_hf": got.ids == hf_ids,
                                   "n_ids": [len(x) for x in got.ids], "n_ids_hf": [len(x) for x in hf_ids]})


@pytest.fixture(scope="module")
def full(gpu_device):
    sd = synthetic_state_dict(FASTCONFORMER_619M, 0)
    return AsrModel(FASTCONFORMER_619M, sd, SyntheticTokenizer(FASTCONFORMER_619M.vocab_size), device="cuda:0"), sd


@pytest.mark.parametrize("fuse_glu", [1, 0])
def test_encoder_619m_vs_bf16_oracle_with_taps(full, fuse_glu):
    """the 24-layer model on 3 ragged utterances (1.2 - 3 s + 0.5 s pad each side) against the bf16-recipe oracle:
    subsampling output, layers 0 / 11 / 23, encoder output, joint projection; then decode bit-exact on the HIP joint
    projection and the id agreement with the oracle's own end-to-end greedy"""
    model, sd = full
    cfg = model.cfg
    audio, lens = synthetic_batch(3, 3.0, seed=123, ragged=True, min_seconds=1.2)
    waves = [audio[b, :lens[b]] for b in range(3)]
    buf = model.stage(waves)
    M = buf.B * buf.tp_max
    dev = model.device
    tap_ids = (0, 11, 23)
    sub = torch.zeros((M, cfg.d_model), dtype=torch.float32, device=dev)
    lay = torch.zeros((len(tap_ids), M, cfg.d_model), dtype=torch.float32, device=dev)
    enc = torch.zeros((buf.B, buf.tp_max, cfg.d_model), dtype=torch.float32, device=dev)
    model.ctx.set_taps(sub, lay, tap_ids)
    model.ctx.set_option("fuse_glu", fuse_glu)     # 1 (default): GLU in the pw1 GEMM epilogue; 0: in the depthwise kernel
    try:
        model.run_device(buf, want_enc=enc)
        torch.cuda.synchronize()
    finally:
        model.ctx.set_taps()
        model.ctx.set_option("fuse_glu", 1)
    padded = np.zeros((3, audio.shape[1] + 16000), np.float32)
    for b in range(3):
        padded[b, 8000:8000 + lens[b]] = waves[b]
    taps = {}
    f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(padded), torch.from_numpy(lens + 16000),
                                    "bf16-fused-glu" if fuse_glu == 1 else "bf16", taps)
    assert buf.enc_lens.cpu().tolist() == el.tolist()
    Tp = buf.tp_max
    sub = sub.cpu().view(3, Tp, -1)
    lay = lay.cpu().view(len(tap_ids), 3, Tp, -1)
    enc = enc.cpu()
    stats = {}

    def cmp(name, got, want, tol_max, tol_mean):
        mx, sm, cnt = 0.0, 0.0, 0
        for b in range(3):
            n = int(el[b])
            d = (got[b, :n] - want[b, :n]).abs()
            mx, sm, cnt = max(mx, d.max().item()), sm + d.sum().item(), cnt + d.numel()
        stats[name] = {"max": mx, "mean": sm / cnt}
        return mx <= tol_max and sm / cnt <= tol_mean

    ok = cmp("sub_out", sub, taps["sub_out"], 0.2, 0.015)
    for k, i in enumerate(tap_ids):
        ok &= cmp(f"layer{i}", lay[k], taps[f"layer{i}"], TOL_MAX, TOL_MEAN)
    ok &= cmp("enc", enc, taps["enc"], TOL_MAX, TOL_MEAN)
    ok &= cmp("joint_enc", buf.joint_enc.cpu(), f_ref, TOL_MAX, TOL_MEAN)
    assert torch.equal(lay[len(tap_ids) - 1], enc), "the last layer's tap is the encoder output"
    got = model.collect(buf)
    ref_same = og.rnnt_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    ref_e2e = og.rnnt_greedy(cfg, sd, f_ref.numpy(), el.numpy())
    stats["ids_equal_oracle_e2e"] = [got.ids[b] == ref_e2e[b][0] for b in range(3)]
    stats["n_ids"] = [len(x) for x in got.ids]
    report(f"encoder_619m_fuse_glu{fuse_glu}", stats)
    assert ok, stats
    assert got.ids == [r[0] for r in ref_same] and got.frames == [r[1] for r in ref_same]


@pytest.mark.parametrize("options", [dict(decode_screen=0, decode_narrow=0), dict(decode_screen=0, decode_narrow=1),
                                     dict(decode_screen=1, decode_narrow=0), dict(decode_screen=1, decode_narrow=1)])
def test_every_decode_kernel_family_is_bit_exact(wide, options):
    """the four ways the library can run the greedy loop — wide / narrow LSTM tiles x exact / screened joint — emit
    identical ids and frames (the oracle's)"""
    model, sd = wide
    cfg = model.cfg
    g = torch.Generator().manual_seed(4)
    B, Tp = 21, 33
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * (0.8 + 0.4 * torch.rand((B, 1, 1), generator=g))
    lens = torch.randint(1, Tp + 1, (B,), generator=g, dtype=torch.int32)
    lens[2] = 0
    ref = og.rnnt_greedy(cfg, sd, f.numpy(), lens.numpy())
    ctx = model.ctx.clone()
    try:
        for k, v in options.items():
            ctx.set_option(k, v)
        saved, model.ctx = model.ctx, ctx
        try:
            got = run_greedy(model, f, lens)
        finally:
            model.ctx = saved
    finally:
        ctx.close()
    assert sum(len(r[0]) for r in ref) > 100
    assert got == ref


# ---- the benchmark batch itself: batch invariance and the flip audit -------------------------
@pytest.fixture(scope="module")
def bench_batch(full):
    """the benchmark's first resident batch (bench.py: seed 1234, 256 x 10 s) through the one-stream path with the
    encoder output tapped: the launch geometry of the timed region (256- / 192-row GEMM tiles, GLU in the pw1 epilogue)"""
    model, sd = full
    audio, lens = synthetic_batch(256, 10.0, seed=1234)
    buf = model.stage([audio[b, :lens[b]] for b in range(256)], buf=model.new_buffers(256, 160000))
    enc = torch.zeros((256, buf.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    got = model.collect(buf)
    return audio, lens, enc.cpu(), buf.joint_enc.cpu(), got


def test_619m_alone_equals_inside_b256(full, bench_batch):
    """The reference runs every utterance alone (batch_size=1, pkg/nemo-asr/src/transcribe.py:48-50).  Here an
    utterance ALONE (M = 138 rows: 64-row GEMM tiles) and the same utterance as row k of the benchmark batch of 256
    (M = 35 328: 256- / 192-row tiles) must give the same BITS — encoder output, joint projection, ids, frames.
    Also inside a ragged batch of 256 (lengths U(2 s, 10 s), SURVEY §8d seed 1235), where the batch pads a short
    utterance to 138 frames and alone it is padded to its own length."""
    model, sd = full
    audio, lens, enc, f, got = bench_batch
    for k in (0, 77, 255):
        one = model.stage([audio[k, :lens[k]]])
        e1 = torch.zeros((1, one.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device)
        model.run_device(one, want_enc=e1)
        torch.cuda.synchronize()
        r1 = model.collect(one)
        n = r1.enc_lens[0]
        assert n == got.enc_lens[k]
        assert torch.equal(e1[0, :n].cpu(), enc[k, :n]), f"encoder output of utterance {k} depends on the batch"
        assert torch.equal(one.joint_enc[0, :n].cpu(), f[k, :n])
        assert r1.ids[0] == got.ids[k] and r1.frames[0] == got.frames[k]
    ra, rl = synthetic_batch(256, 10.0, seed=1235, ragged=True, min_seconds=2.0)
    waves = [ra[b, :rl[b]] for b in range(256)]
    buf = model.stage(waves, buf=model.new_buffers(256, 160000))
    enc_r = torch.zeros((256, buf.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc_r)
    torch.cuda.synchronize()
    got_r = model.collect(buf)
    short, long_ = int(np.argmin(rl)), int(np.argmax(rl))
    for k in (short, long_, 100):
        one = model.stage([waves[k]])
        e1 = torch.zeros((1, one.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device)
        model.run_device(one, want_enc=e1)
        torch.cuda.synchronize()
        r1 = model.collect(one)
        n = r1.enc_lens[0]
        assert n == got_r.enc_lens[k]
        assert torch.equal(e1[0, :n].cpu(), enc_r[k, :n].cpu()), f"ragged batch: utterance {k} ({rl[k]} samples)"
        assert r1.ids[0] == got_r.ids[k] and r1.frames[0] == got_r.frames[k]


def test_619m_b256_rows_vs_fp32_oracle_with_flip_audit(full, bench_batch):
    """Rows 0..3 of the benchmark batch (the kernels of the timed region) against the END-TO-END fp32 oracle: encoder /
    joint projection within the stated tolerance, decode bit-exact on the HIP joint projection, and every greedy-id
    difference audited (oracle/audit.py): it must start at a decision where the oracle's own margin between the two
    candidates is below (|w_a| + |w_b|) * |delta f| — a near-tie the stated encoder tolerance can move — and an
    utterance without such a decision must have identical ids."""
    from oracle import audit
    model, sd = full
    cfg = model.cfg
    audio, lens, enc, f, got = bench_batch
    k = 4
    audits, equal, stats = [], [], {"enc_max": 0.0, "joint_max": 0.0}
    wnorm_max = float(sd["joint.joint_net.2.weight"].norm(dim=1).max())
    for b in range(k):
        wav = np.pad(audio[b, :lens[b]], 8000)
        taps = {}
        f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32", taps)
        n = int(el[0])
        assert n == got.enc_lens[b]
        de, dj = (enc[b, :n] - taps["enc"][0, :n]).abs(), (f[b, :n] - f_ref[0, :n]).abs()
        stats["enc_max"], stats["joint_max"] = max(stats["enc_max"], de.max().item()), max(stats["joint_max"], dj.max().item())
        assert de.max() <= TOL_MAX and de.mean() <= TOL_MEAN and dj.max() <= TOL_MAX, (b, de.max().item(), dj.max().item())
        ref = og.rnnt_greedy(cfg, sd, f_ref.numpy(), el.numpy())[0]
        equal.append(got.ids[b] == ref[0])
        a = audit.flip_audit(cfg, sd, f_ref[0, :n].numpy(), f[b, :n].numpy(), n, got.ids[b], got.frames[b])
        audits.append(a)
        for fl in a["flips"]:
            assert fl["margin_ref"] <= fl["bound"] * (1 + 1e-9) + 1e-12, fl            # Lipschitz bound (theorem)
            assert fl["delta_f"] <= TOL_MAX * cfg.joint_hidden ** 0.5                    # ... of a difference within tolerance
            assert fl["margin_ref"] <= 2 * wnorm_max * TOL_MAX * cfg.joint_hidden ** 0.5
    same = og.rnnt_greedy(cfg, sd, f[:k].numpy(), np.asarray(got.enc_lens[:k], np.int32))
    assert [got.ids[b] for b in range(k)] == [r[0] for r in same] and [got.frames[b] for b in range(k)] == [r[1] for r in same]
    s = audit.summarize(audits, equal)
    s.update(stats, ids_equal_oracle_e2e=equal)
    report("b256_rows_flip_audit", s)
    assert s["walk_reproduces_hip_path"]
    assert s["every_id_difference_starts_at_a_flip"], s
    if s["local_flips"]:
        # flips live in the bottom of the oracle's own margin distribution: near-ties, not systematic disagreement
        assert s["flip_margin_percentile_of_all_margins_max"] <= 50.0, s


def test_619m_limited_context_attention_vs_oracle(full):
    """the attention variant the shipped checkpoint is believed to use ([128, 128] + 1 global token) at
    the benchmark geometry (T' = 138, 8 heads): WINDOW kernel vs the bf16-recipe oracle with the same predicate"""
    _, sd = full
    cfg = FASTCONFORMER_619M.with_(att_left=128, att_right=128, n_global=1)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(2, 10.0, seed=321, ragged=True, min_seconds=9.0)      # T' > 129: the window masks some pairs
    waves = [audio[b, :lens[b]] for b in range(2)]
    buf = model.stage(waves)
    enc = torch.zeros((2, buf.tp_max, cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    padded = np.zeros((2, audio.shape[1] + 16000), np.float32)
    for b in range(2):
        padded[b, 8000:8000 + lens[b]] = waves[b]
    taps = {}
    f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(padded), torch.from_numpy(lens + 16000), "bf16-fused-glu", taps)
    assert buf.enc_lens.cpu().tolist() == el.tolist() and int(el.max()) > 130
    worst = 0.0
    for b in range(2):
        n = int(el[b])
        d = (enc.cpu()[b, :n] - taps["enc"][b, :n]).abs()
        worst = max(worst, d.max().item())
        assert d.max() <= TOL_MAX and d.mean() <= TOL_MEAN, (b, d.max().item(), d.mean().item())
    report("encoder_619m_window_128_128_g1", {"max": worst})
    got = model.collect(buf)
    ref = og.rnnt_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref]


# ---- ALL 256 rows: the float32 parity mode against the committed float32-oracle golden, and the throughput mode audited
# ---- against the parity mode on every row ----------------------------------------------------------
BENCH_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_fp32.npz")
TOL_F32 = 1e-4                       # encoder / joint projection of the parity mode vs the float32 oracle (measured ~2e-5)
SETS = {"equal": dict(seed=1234), "ragged": dict(seed=1235, ragged=True, min_seconds=2.0)}


def golden_rows(gold, name):
    off = gold[name + "_offsets"]
    ids = [gold[name + "_ids"][off[b]:off[b + 1]].tolist() for b in range(len(off) - 1)]
    frames = [gold[name + "_frames"][off[b]:off[b + 1]].tolist() for b in range(len(off) - 1)]
    return ids, frames


def projection(J, seed, device):
    g = torch.Generator().manual_seed(int(seed))
    return (torch.randn((J, 8), generator=g, dtype=torch.float32) / J ** 0.5).to(device)


@pytest.fixture(scope="module")
def full32(full):
    _, sd = full
    cfg = FASTCONFORMER_619M
    return AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0", precision="fp32")


@pytest.fixture(scope="module")
def fp32_runs(full32):
    """the parity mode over the benchmark batch and the ragged set -> {name: (audio, lens, DecodedBatch, joint_enc on the device)}"""
    import hashlib
    gold = np.load(BENCH_GOLD)
    out = {}
    for name, kw in SETS.items():
        audio, lens = synthetic_batch(256, 10.0, **kw)
        assert hashlib.sha256(audio.tobytes()).digest() == bytes(gold[name + "_audio_sha256"].tolist()), "inputs drifted from the golden's"
        buf = full32.stage([audio[b, :lens[b]] for b in range(256)], buf=full32.new_buffers(256, 160000))
        full32.run_device(buf)
        torch.cuda.synchronize()
        out[name] = (audio, lens, full32.collect(buf), buf.joint_enc.clone())
    return out


def check_fp32_rows(full32, fp32_runs, name, key):
    """shared by the exact float32 mode and the three-term-product mode: see test_619m_fp32_mode_every_row_vs_fp32_oracle_golden"""
    from oracle import audit
    gold = np.load(BENCH_GOLD)
    rows = int(gold["rows"])
    cfg = FASTCONFORMER_619M
    audio, lens, got, f_dev = fp32_runs[name]
    g_ids, g_frames = golden_rows(gold, name)
    assert got.enc_lens[:rows] == gold[name + "_enc_lens"].tolist()
    R = projection(cfg.joint_hidden, gold["proj_seed"], f_dev.device)
    proj = (f_dev @ R).cpu().numpy()
    worst_proj = worst_f = 0.0
    for b in range(rows):
        n = got.enc_lens[b]
        worst_proj = max(worst_proj, float(np.abs(proj[b, :n] - gold[name + "_proj"][b, :n]).max()))
    for b in range(2):
        n = got.enc_lens[b]
        worst_f = max(worst_f, float((f_dev[b, :n].cpu() - torch.from_numpy(gold[name + "_f_rows"][b, :n])).abs().max()))
    assert worst_proj <= TOL_F32 and worst_f <= TOL_F32, (worst_proj, worst_f)
    near = set(int(b) for b in np.nonzero(gold[name + "_min_margin"] < float(gold["near_tie"]))[0])
    differ = [b for b in range(rows) if got.ids[b] != g_ids[b] or got.frames[b] != g_frames[b]]
    assert not [b for b in differ if b not in near], f"rows {differ} differ from the float32 oracle without a near-tie"
    # a differing near-tie row: walk the GOLDEN hypothesis on this run's joint projection (float64): the first decision where
    # the two sides part must be one whose margin on this side is itself below 1e-4
    explained = []
    if differ:
        sd = full32_sd(full32)
        a = audit.flip_audit_batch(cfg, sd, f_dev[differ], f_dev[differ], [got.enc_lens[b] for b in differ],
                                   [g_ids[b] for b in differ], [g_frames[b] for b in differ], device=f_dev.device)
        for b, r in zip(differ, a):
            first = min(r["flips"], key=lambda x: x["frame"]) if r["flips"] else None
            explained.append({"row": b, "golden_min_margin": float(gold[name + "_min_margin"][b]),
                              "margin_at_first_difference": None if first is None else first["margin_ref"]})
            assert first is not None and first["margin_ref"] <= 1e-4, (b, first)
    report(f"{key}_{name}", {"rows": rows, "ids_and_frames_exact": f"{rows - len(differ)}/{rows}",
                                 "near_tie_rows_in_golden": len(near), "differing_rows": explained,
                                 "joint_proj_fingerprint_max_err": worst_proj, "joint_enc_rows01_max_err": worst_f,
                                 "decisions": int(gold[name + "_n_decisions"].sum())})




@pytest.mark.parametrize("name", ["equal", "ragged"])
def test_619m_fp32_mode_every_row_vs_fp32_oracle_golden(full32, fp32_runs, name):
    """`load_model(precision="fp32")` on ALL 256 rows of the benchmark batch (seed 1234) and of the ragged set (seed 1235)
    against tests/golden/bench_fp32.npz — the float32 oracle run end to end, one utterance per call with the reference's
    padding (pkg/nemo-asr/src/transcribe.py:44-53):
      * encoder lengths identical; the joint projection of EVERY row within 1e-4 (8-dim fingerprint of all rows, the full
        tensor for rows 0 and 1);
      * greedy ids AND emission frames IDENTICAL on every row whose float32-oracle decision margins all exceed the
        generator's near-tie threshold (1e-3, two orders above float32 reassociation noise; the rows below it are named by
        the golden itself, not by this comparison);
      * on the near-tie rows: identical too, or the difference starts at a decision whose oracle margin is below 1e-4."""
    check_fp32_rows(full32, fp32_runs, name, "fp32_mode")


def test_619m_fp32x3_mode_every_row_vs_fp32_oracle_golden(full):
    """`load_model(precision="fp32x3")`: the float32 mode with every float32 product of its GEMMs formed from three bf16 matrix-core
    terms (csrc/k_f32.hip X3; 2x the float32 mode's speed) — the SAME golden, the same assertions as the exact mode, both sets:
    every row whose oracle margins exceed the near-tie threshold identical, a differing near-tie row explained by a margin
    below 1e-4 on this side (round 6 measured 256 / 256 on both sets)."""
    _, sd = full
    cfg = FASTCONFORMER_619M
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0", precision="fp32x3")
    assert model.x3 and model.precision == "fp32"
    gold = np.load(BENCH_GOLD)
    for name, kw in SETS.items():
        audio, lens = synthetic_batch(256, 10.0, **kw)
        buf = model.stage([audio[b, :lens[b]] for b in range(256)], buf=model.new_buffers(256, 160000))
        model.run_device(buf)
        torch.cuda.synchronize()
        runs = {name: (audio, lens, model.collect(buf), buf.joint_enc.clone())}
        check_fp32_rows(model, runs, name, "fp32x3_mode")
        del buf
    del model
    torch.cuda.empty_cache()


def full32_sd(model):
    return synthetic_state_dict(model.cfg, 0)


def test_619m_throughput_mode_flip_audit_over_all_256_rows(full, bench_batch, fp32_runs):
    """The bf16 throughput mode on ALL 256 rows of the benchmark batch, audited against the float32 parity mode's joint
    projection of the same rows (itself pinned to the float32 oracle on every row by the test above): every row WITHOUT a
    local flip has ids and frames identical to the float32 oracle golden's; every flip obeys the Lipschitz bound with a
    joint-projection difference inside the stated encoder tolerance (an ABSOLUTE cap: a regression cannot excuse itself);
    flips sit at the bottom of the margin distribution."""
    from oracle import audit
    model, sd = full
    cfg = model.cfg
    gold = np.load(BENCH_GOLD)
    rows = int(gold["rows"])
    audio, lens, enc, f16, got = bench_batch
    _, _, got32, f32_dev = fp32_runs["equal"]
    g_ids, g_frames = golden_rows(gold, "equal")
    assert got.enc_lens == got32.enc_lens
    dj = 0.0
    for b in range(rows):
        n = got.enc_lens[b]
        dj = max(dj, float((f16[b, :n] - f32_dev[b, :n].cpu()).abs().max()))
    assert dj <= TOL_MAX, dj
    audits = audit.flip_audit_batch(cfg, sd, f32_dev[:rows], f16[:rows], got.enc_lens[:rows], got.ids[:rows], got.frames[:rows],
                                    device=f32_dev.device)
    equal = [got.ids[b] == g_ids[b] and got.frames[b] == g_frames[b] for b in range(rows)]
    wnorm_max = float(sd["joint.joint_net.2.weight"].norm(dim=1).max())
    for a in audits:
        for fl in a["flips"]:
            assert fl["margin_ref"] <= fl["bound"] * (1 + 1e-9) + 1e-12, fl
            assert fl["delta_f"] <= TOL_MAX * cfg.joint_hidden ** 0.5
            assert fl["margin_ref"] <= 2 * wnorm_max * TOL_MAX * cfg.joint_hidden ** 0.5
    s = audit.summarize(audits, equal)
    s.update(rows=rows, ids_and_frames_equal_fp32_oracle=f"{sum(equal)}/{rows}", joint_enc_max_diff_vs_fp32_mode=dj)
    report("b256_all_rows_flip_audit", s)
    assert s["walk_reproduces_hip_path"] and s["every_id_difference_starts_at_a_flip"], s
    if s["local_flips"]:
        assert s["flip_margin_percentile_of_all_margins_max"] <= 50.0, s
