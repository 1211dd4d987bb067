"""-m gpu: the float32 PARITY MODE (`load_model(precision="fp32")`, rs_set_option "precision_f32"; csrc/k_f32.hip).

The reference runs NeMo in float32 without autocast (pkg/nemo-asr/src/transcribe.py:26-28, :48-53).  In this mode the HIP
path does too — float32 weights, activations and arithmetic end to end — so the comparison with the float32 oracle is a
statement about the WHOLE path:

  operators                 float32 GEMM / attention / conv-module middle vs float64 references: |err| <= 2e-5 on O(1) values
  encoder, joint projection max |err| <= 1e-4 against the float32 oracle AND against the float32 HF parakeet golden
  greedy ids, frames        IDENTICAL to the float32 oracle's and to HF `generate`'s, end to end
"""
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import TINY, WIDE2
from reazonspeech_amd.runtime.model import AsrModel
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import model as om, greedy as og
from test_oracle_pinned import WIDE_GOLD, wide_case

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_ENC = 1e-4


@pytest.fixture(scope="module")
def ctx(gpu_device):
    c = capi.Context(TINY, 0)
    yield c
    c.close()


def sync():
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (130, 72, 192), (1, 640, 256), (517, 3072, 256), (4416, 1024, 4096),
                                   (300, 256, 2560)])
def test_gemm_f32_shapes(ctx, gpu_device, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / K ** 0.5
    out = torch.full((M + 3, N), 7.0, dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(A.to(gpu_device), W.to(gpu_device), out[:M])
    sync()
    ref = (A.double() @ W.double().t()).float()
    assert (out[:M].cpu() - ref).abs().max() <= 2e-5
    assert (out[M:] == 7.0).all(), "rows past M were written"


def test_gemm_f32_layout_asymmetric(ctx, gpu_device):
    """A = I against an asymmetric W catches a transposed or permuted accumulator write"""
    M = N = K = 256
    W = torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.001953125
    out = torch.zeros((M, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(torch.eye(M, K).to(gpu_device), W.to(gpu_device), out)
    sync()
    assert torch.equal(out.cpu(), W.t().contiguous())


def test_gemm_f32_epilogues_and_invariance(ctx, gpu_device):
    g = torch.Generator().manual_seed(5)
    B, T, Fq, K, N = 3, 11, 5, 128, 192
    M = B * T * Fq
    A = torch.randn((M, K), generator=g)
    W = torch.randn((N, K), generator=g) / K ** 0.5
    bias = torch.randn((N,), generator=g)
    res = torch.randn((M, N), generator=g)
    lens = torch.tensor([11, 4, 0], dtype=torch.int32)
    dA, dW, db = A.to(gpu_device), W.to(gpu_device), bias.to(gpu_device)
    base = (A.double() @ W.double().t() + bias.double())
    out = torch.zeros((M, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(dA, dW, out, flags=capi.GEMM_BIAS | capi.GEMM_SILU, bias=db)
    sync()
    assert (out.cpu() - torch.nn.functional.silu(base).float()).abs().max() <= 2e-5
    stream = res.clone().to(gpu_device)
    ctx.gemm_f32(dA, dW, stream, flags=capi.GEMM_BIAS | capi.GEMM_RESIDUAL, bias=db, alpha=0.5, residual=stream)
    sync()
    assert (stream.cpu() - (res.double() + 0.5 * base).float()).abs().max() <= 2e-5
    out = torch.ones((M, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(dA, dW, out, flags=capi.GEMM_BIAS | capi.GEMM_RELU | capi.GEMM_ROWMASK, bias=db, mask_lens=lens.to(gpu_device),
                 mask_rows=Fq, mask_steps=T)
    sync()
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None, None].expand(B, T, Fq, N).reshape(M, N)
    got = out.cpu()
    assert (got - (torch.relu(base).float() * mask)).abs().max() <= 2e-5
    assert torch.all(got[mask == 0] == 0)
    # a row's bits do not depend on M or on where it sits in a tile (batch invariance of the parity mode)
    full = torch.zeros((M, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(dA, dW, full, flags=capi.GEMM_BIAS, bias=db)
    part = torch.zeros((7, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm_f32(dA[130:137].contiguous(), dW, part, flags=capi.GEMM_BIAS, bias=db)
    sync()
    assert torch.equal(part.cpu(), full[130:137].cpu())
    with pytest.raises(capi.RsError):
        ctx.gemm_f32(dA[:, :48].contiguous(), dW[:, :48].contiguous(), full)


@pytest.mark.parametrize("T,lens,window,heads", [(19, [19, 14], None, 2), (138, [138, 97, 5], None, 2), (70, [70, 33], None, 2),
                                                 (138, [138, 60], (32, 16, 1), 2), (300, [300, 160], (40, 100, 0), 2)])
def test_attention_f32(gpu_device, T, lens, window, heads):
    """full, windowed and global-token attention against the oracle's predicate, float64 reference"""
    cfg = TINY if window is None else TINY.with_(att_left=window[0], att_right=window[1], n_global=window[2])
    c = capi.Context(cfg, 0)
    g = torch.Generator().manual_seed(T + len(lens))
    B, d = len(lens), cfg.d_model
    H, dh = heads, d // heads
    qkv = torch.randn((B, T, 3 * d), generator=g)
    p = torch.randn((2 * T - 1, d), generator=g)
    bu, bv = 0.3 * torch.randn((H, dh), generator=g), 0.3 * torch.randn((H, dh), generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, T, H, dh).double() for i in range(3))
    ref = om.attention_core(cfg, q, k, v, p.view(2 * T - 1, H, dh).double(), bu.double(), bv.double(), lens_t, "fp32").float()
    out = torch.full((B * T, d), 3.0, dtype=torch.float32, device=gpu_device)
    c.attention_f32(qkv.reshape(B * T, 3 * d).to(gpu_device), p.to(gpu_device), bu.reshape(-1).to(gpu_device),
                    bv.reshape(-1).to(gpu_device), lens_t.to(torch.int32).to(gpu_device), B, T, out)
    sync()
    got = out.cpu().view(B, T, d)
    for b in range(B):
        n = lens[b]
        err = (got[b, :n] - ref[b, :n]).abs().max().item()
        assert err <= 2e-5, (b, err)
        assert torch.all(got[b, n:] == 0)
    c.close()


@pytest.mark.parametrize("T,lens,k", [(19, [19, 14], 9), (70, [70, 33, 1], 9), (50, [50, 20], 31)])
def test_glu_dwconv_f32(ctx, gpu_device, T, lens, k):
    g = torch.Generator().manual_seed(T)
    B, d = len(lens), 256
    x = torch.randn((B, T, 2 * d), generator=g)
    w = torch.randn((d, k), generator=g) / 3
    b = 0.1 * torch.randn((d,), generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    xd = x.double()
    u = xd[..., :d] * torch.sigmoid(xd[..., d:]) * (torch.arange(T)[None, :] < lens_t[:, None])[:, :, None]
    z = torch.nn.functional.conv1d(u.transpose(1, 2), w.double()[:, None, :], b.double(), padding=(k - 1) // 2, groups=d).transpose(1, 2)
    ref = torch.nn.functional.silu(z).float()
    out = torch.zeros((B * T, d), dtype=torch.float32, device=gpu_device)
    ctx.glu_dwconv_f32(x.reshape(B * T, 2 * d).to(gpu_device), w.t().contiguous().to(gpu_device), b.to(gpu_device),
                       lens_t.to(gpu_device), B, T, d, k, out)
    sync()
    assert (out.cpu().view(B, T, d) - ref).abs().max() <= 2e-5


# ---- end to end --------------------------------------------------------------------------------------------------
def _run(model, audio, lens):
    waves = [audio[b, :int(lens[b])] for b in range(audio.shape[0])]
    buf = model.stage(waves)
    enc = torch.zeros((buf.B, buf.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    return buf, enc.cpu(), model.collect(buf)


def _fp32_cases():
    gold = np.load(os.path.join(GOLDEN, "parakeet_tiny.npz"))
    sd = synthetic_state_dict(TINY, int(gold["seed"]), blank_bias=float(gold["blank_bias"]))
    yield "tiny", TINY, sd, gold["audio"], gold["lengths"], gold, ""
    wide = np.load(WIDE_GOLD)
    for seed in (int(s) for s in wide["seeds"]):
        cfg, sd, audio, lens = wide_case(wide, seed)
        yield f"wide-{seed}", cfg, sd, audio, lens, wide, f"s{seed}_"


def test_fp32_mode_matches_oracle_and_hf_golden(gpu_device):
    """toy geometry and the 619M layer geometry (2 layers, three seeds): the parity mode against the float32 oracle run end
    to end and against the committed HF parakeet goldens — encoder / joint projection within 1e-4, ids and frames IDENTICAL
    to both"""
    worst = 0.0
    for name, cfg, sd, audio, lens, gold, pre in _fp32_cases():
        model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0", pad_seconds=0.0, precision="fp32")
        buf, enc, got = _run(model, audio, lens)
        taps = {}
        f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(audio), torch.from_numpy(lens), "fp32", taps)
        assert got.enc_lens == el.tolist() == gold[pre + "hf_enc_lens"].tolist()
        f = buf.joint_enc.cpu()
        hf, hfj = torch.from_numpy(gold[pre + "hf_enc"]), torch.from_numpy(gold[pre + "hf_joint_enc"])
        for b in range(len(el)):
            n = int(el[b])
            errs = [(enc[b, :n] - taps["enc"][b, :n]).abs().max().item(), (f[b, :n] - f_ref[b, :n]).abs().max().item(),
                    (enc[b, :n] - hf[b, :n]).abs().max().item(), (f[b, :n] - hfj[b, :n]).abs().max().item()]
            worst = max(worst, *errs)
            assert max(errs) <= TOL_ENC, (name, b, errs)
        ref = og.rnnt_greedy(cfg, sd, f_ref.numpy(), el.numpy())
        assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref], name
        for b in range(len(el)):
            n = int(gold[pre + "hf_n_ids"][b])
            assert got.ids[b] == gold[pre + "hf_ids"][b, :n].tolist(), (name, b)
            assert got.frames[b] == gold[pre + "hf_frames"][b, :n].tolist(), (name, b)
        del model
    print(f"fp32 mode: worst |err| vs oracle / HF over the toy and wide goldens = {worst:.3g}")


def test_fp32_mode_is_batch_invariant_and_survives_table_growth(gpu_device):
    """an utterance alone == the same utterance inside a ragged batch, bit for bit (encoder rows, ids, frames); a long
    utterance grows the float32 position table on the fly"""
    sd = synthetic_state_dict(TINY, 21, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0", precision="fp32", pos_cap=32)
    audio, lens = synthetic_batch(5, 3.0, seed=9, ragged=True, min_seconds=0.5)
    buf, enc, together = _run(model, audio, lens)
    assert model.pos_cap >= buf.tp_max > 32
    for b in (0, 3):
        _, e1, alone = _run(model, audio[b:b + 1], lens[b:b + 1])
        n = alone.enc_lens[0]
        assert torch.equal(e1[0, :n], enc[b, :n])
        assert alone.ids[0] == together.ids[b] and alone.frames[0] == together.frames[b]
    # the two precisions of one set of weights agree to bf16 noise (they are the same model)
    m16 = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    _, e16, _ = _run(m16, audio, lens)
    for b in range(5):
        n = together.enc_lens[b]
        assert (e16[b, :n] - enc[b, :n]).abs().max() <= 6e-2


def test_precision_option_needs_the_f32_tensors(gpu_device):
    sd = synthetic_state_dict(TINY, 3)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    with pytest.raises(capi.RsError):
        model.ctx.set_option("precision_f32", 1)
    with pytest.raises(ValueError):
        AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0", precision="fp16")
