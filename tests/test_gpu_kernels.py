"""-m gpu: per-operator parity of the HIP kernels (through the C ABI) against the CPU oracle.

Tolerances (bf16 operands, f32 accumulation):
  GEMM            |err| <= 2e-3 * (|a|.|w| row/col norms)  (f32 reassociation only: operands are
                  rounded to bf16 identically on both sides)
  LayerNorm       1e-5 relative (f32), bf16 output within 1 bf16 ulp
  attention ctx   2e-2 absolute on O(1) values (probabilities rounded to bf16 against a running
                  instead of the final row maximum, fast exp)
  front-end       2e-3 absolute on normalised log-mel (f32 FFT vs pocketfft)
"""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import TINY, FASTCONFORMER_619M
from oracle import model as om

pytestmark = pytest.mark.gpu

import contextlib
import ctypes


@contextlib.contextmanager
def gemm_knobs(ctx, tile=0, pairs=2, breg=0):
    """force the GEMM tile height (256 / 192 / 128 / 64; 0 = by shape) and the tiles-per-workgroup mode (2 = pairs with
    the LDS ring carried from the first tile into the second, the default; 1 = pairs, ring restarted; 0 = one tile per
    workgroup) for the calls inside"""
    lib = ctx.lib
    for f in (lib.rs_debug_set_gemm_tile, lib.rs_debug_set_gemm_pairs, lib.rs_debug_set_gemm_breg):
        f.argtypes = [ctypes.c_int]
        f.restype = None
    try:
        lib.rs_debug_set_gemm_tile(tile)
        lib.rs_debug_set_gemm_pairs(pairs)
        lib.rs_debug_set_gemm_breg(breg)      # 1 = the weight operand stays out of LDS (fragment-major copy, global -> VGPR)
        yield
    finally:
        lib.rs_debug_set_gemm_tile(0)
        lib.rs_debug_set_gemm_pairs(2)
        lib.rs_debug_set_gemm_breg(0)


@pytest.fixture(scope="module")
def ctx(gpu_device):
    c = capi.Context(TINY, 0)
    yield c
    c.close()


def bf(t):
    return t.to(torch.bfloat16)


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------
def test_mfma_layout_asymmetric(ctx, gpu_device):
    """A = I against an asymmetric W catches a transposed or permuted C-write (guide G9)."""
    M = N = K = 128
    A = torch.eye(M, K)
    W = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.001953125)   # exact in bf16? keep small
    W = rb(W)
    out = torch.zeros((M, N), dtype=torch.float32, device=gpu_device)
    ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out, flags=capi.GEMM_OUT_F32)
    sync()
    assert torch.equal(out.cpu(), W.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (130, 72, 192), (1, 640, 256),
                                   (517, 3072, 256), (4416, 512, 256)])
def test_gemm_shapes(ctx, gpu_device, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    ref = A @ W.t()
    out = torch.full((M, N), 7.0, dtype=torch.float32, device=gpu_device)
    ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out, flags=capi.GEMM_OUT_F32)
    sync()
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 2e-3, err


@pytest.mark.parametrize("tile", [0, 256, 192, 128, 64])
@pytest.mark.parametrize("N,K,residual", [(4096, 1024, False), (1024, 4096, True), (1024, 256, True)])
def test_gemm_big_tiles(ctx, gpu_device, N, K, residual, tile):
    """The benchmark-geometry launches (M = 35 328 - 37 rows: ragged last tile) at every tile height of the kernel (0 =
    what the launcher picks: 256 rows for the bf16 shapes, 192 for the N = 1024 residual family), checked on three row
    bands (first, middle, ragged tail) against a float32 reference of the bf16-rounded operands; the rows past M must
    stay untouched."""
    M = 35328 - 37
    g = torch.Generator().manual_seed(N + K)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    res = torch.randn((M, N), generator=g) if residual else None
    rows = torch.cat([torch.arange(0, 256), torch.arange(17000, 17256), torch.arange(M - 300, M)])
    ref = A[rows] @ W.t() + bias
    if residual:
        ref = 0.5 * ref + res[rows]
        flags, alpha = capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, 0.5
        out_full = torch.full((M + 64, N), 7.0, dtype=torch.float32, device=gpu_device)
    else:
        ref = torch.nn.functional.silu(ref)
        flags, alpha = capi.GEMM_BIAS | capi.GEMM_SILU, 1.0
        out_full = torch.full((M + 64, N), 7.0, dtype=torch.bfloat16, device=gpu_device)
    with gemm_knobs(ctx, tile=tile):
        ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out_full[:M], flags=flags, bias=bias.to(gpu_device),
                 alpha=alpha, residual=res.to(gpu_device) if residual else None)
        sync()
    got = out_full[rows.to(gpu_device)].float().cpu()
    tol = 2e-3 + (0.0 if residual else 2.0 ** -8) * ref.abs()      # bf16 output: one rounding of the result
    bad = ((got - ref).abs() > tol)
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))
    assert (out_full[M:].float() == 7.0).all(), "rows past M were written"


@pytest.mark.parametrize("tile", [256, 192, 128, 64])
def test_gemm_residual_in_place(ctx, gpu_device, tile):
    """residual update IN PLACE (out aliases the residual, as in the encoder's f32 stream) at every tile height, K = 4096
    (64 K tiles through the five-slot ring), ragged last tile"""
    M, N, K = 35328 - 37, 1024, 4096
    g = torch.Generator().manual_seed(tile)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    rows = torch.cat([torch.arange(0, 200), torch.arange(17000, 17200), torch.arange(M - 300, M)])
    x = torch.randn((M, N), generator=g)
    ref = 0.5 * (A[rows] @ W.t() + bias) + x[rows]
    out_full = torch.full((M + 64, N), 7.0, dtype=torch.float32, device=gpu_device)
    out_full[:M] = x.to(gpu_device)
    with gemm_knobs(ctx, tile=tile):
        ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out_full[:M], bias=bias.to(gpu_device),
                 flags=capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, alpha=0.5, residual=out_full[:M])
        sync()
    got = out_full[rows.to(gpu_device)].float().cpu()
    bad = (got - ref).abs() > 2e-3
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))
    assert (out_full[M:].float() == 7.0).all(), "rows past M were written"


@pytest.mark.parametrize("tile", [256, 192, 128, 64])
@pytest.mark.parametrize("M,N,K", [(4416, 512, 128), (5000, 256, 192), (70, 1024, 320), (2049, 768, 1024), (300, 640, 64)])
def test_gemm_split_ring_short_k(ctx, gpu_device, M, N, K, tile):
    """the split ring at the edges of its schedule: ONE K tile (nothing is issued in the loop), two (only B(1)), three
    (one A(t+2)), five (the slot sequence wraps), ragged rows / columns; f32 output"""
    g = torch.Generator().manual_seed(M + N + K)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    ref = A @ W.t() + bias
    out = torch.full((M + 8, N), 7.0, dtype=torch.float32, device=gpu_device)
    with gemm_knobs(ctx, tile=tile):
        ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out[:M], flags=capi.GEMM_BIAS | capi.GEMM_OUT_F32, bias=bias.to(gpu_device))
        sync()
    assert (out[:M].cpu() - ref).abs().max() <= 2e-3
    assert (out[M:] == 7.0).all()


def test_gemm_is_tile_and_batch_invariant(ctx, gpu_device):
    """The batch-invariance contract of the encoder at the operator: an output row is the same BITS whatever the tile
    height and whether a workgroup runs one tile or two, wherever the row sits in the matrix and however many rows ride along (an utterance alone vs
    inside a batch of 256).  Every epilogue: bf16 + SiLU, f32 residual, GLU."""
    M, K, d = 35328 - 37, 1024, 1024
    g = torch.Generator().manual_seed(99)
    A = bf(torch.randn((M, K), generator=g)).to(gpu_device)
    W = bf(torch.randn((2 * d, K), generator=g) / K ** 0.5).to(gpu_device)
    bias = torch.randn((2 * d,), generator=g).to(gpu_device)
    x = torch.randn((M, d), generator=g).to(gpu_device)
    lo, n = 20010, 138                       # "one utterance": 138 rows out of the middle of the batch

    def run(a, res, tile, pairs=2):
        outs = []
        with gemm_knobs(ctx, tile=tile, pairs=pairs):
            o = torch.zeros((a.shape[0], d), dtype=torch.bfloat16, device=gpu_device)
            ctx.gemm(a, W[:d], o, flags=capi.GEMM_BIAS | capi.GEMM_SILU, bias=bias[:d])
            outs.append(o)
            o = res.clone()
            ctx.gemm(a, W[d:], o, flags=capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, bias=bias[d:], alpha=0.5, residual=o)
            outs.append(o)
            o = torch.zeros((a.shape[0], d), dtype=torch.bfloat16, device=gpu_device)
            ctx.gemm(a, W, o, flags=capi.GEMM_BIAS | capi.GEMM_GLU, bias=bias)
            outs.append(o)
            sync()
        return outs

    base = run(A, x, 0)
    for tile, pairs in ((256, 2), (192, 2), (128, 2), (64, 2), (0, 1), (256, 1), (192, 1), (0, 0), (256, 0), (192, 0)):
        for got, want in zip(run(A, x, tile, pairs), base):
            assert torch.equal(got, want), (tile, pairs)
    for pairs in (1, 0):                     # 128-row tiles (three-stage ring), ring restarted / one tile per workgroup
        for got, want in zip(run(A, x, 128, pairs), base):
            assert torch.equal(got, want), ("three-stage ring", pairs)
    alone = run(A[lo:lo + n].contiguous(), x[lo:lo + n].contiguous(), 0)           # picks the 64-row tile on its own
    for got, want in zip(alone, base):
        assert torch.equal(got, want[lo:lo + n])


def test_gemm_register_resident_weights_are_bit_identical(ctx, gpu_device):
    """$RS_GEMM_BREG (round 6's A/B form: the weight operand as fragment-major global -> VGPR loads, only A in the LDS ring) does
    the same per-element arithmetic as the LDS form — ascending 32-deep k-steps into one accumulator — so every epilogue gives
    the same BITS: 256- and 192-row tiles, pairs with the ring carried / restarted / single tiles, K = 64 (one K tile) ..
    4096, ragged M, a registered weight (cached copy) and an unregistered one (scratch)."""
    g = torch.Generator().manual_seed(321)
    for M, N, K in ((35328 - 37, 1024, 1024), (5000, 2048, 64), (9000, 1024, 4096), (3000, 256, 128), (4444, 3072, 320)):
        A = bf(torch.randn((M, K), generator=g)).to(gpu_device)
        W = bf(torch.randn((N, K), generator=g) / K ** 0.5).to(gpu_device)
        bias = torch.randn((N,), generator=g).to(gpu_device)
        x = torch.randn((M, N), generator=g).to(gpu_device)
        if K == 4096:
            ctx.set_tensor("test.breg.w", W)          # registered: the fragment-major copy is made once and cached

        def run(tile, pairs, breg):
            outs = []
            with gemm_knobs(ctx, tile=tile, pairs=pairs, breg=breg):
                o = torch.zeros((M, N), dtype=torch.bfloat16, device=gpu_device)
                ctx.gemm(A, W, o, flags=capi.GEMM_BIAS | capi.GEMM_SILU, bias=bias)
                outs.append(o)
                o = x.clone()
                ctx.gemm(A, W, o, flags=capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, bias=bias, alpha=0.5, residual=o)
                outs.append(o)
                o = torch.zeros((M, N), dtype=torch.float32, device=gpu_device)
                ctx.gemm(A, W, o, flags=capi.GEMM_BIAS | capi.GEMM_OUT_F32, bias=bias)
                outs.append(o)
                o = torch.zeros((M, N // 2), dtype=torch.bfloat16, device=gpu_device)
                ctx.gemm(A, W, o, flags=capi.GEMM_BIAS | capi.GEMM_GLU, bias=bias)
                outs.append(o)
                sync()
            return outs
        base = run(256, 2, 0)
        for tile, pairs in ((256, 2), (192, 2), (256, 1), (192, 0), (256, 0)):
            for k, (got, want) in enumerate(zip(run(tile, pairs, 1), base)):
                assert torch.equal(got, want), (M, N, K, tile, pairs, k)
        for got, want in zip(run(0, 2, 1), base):          # twice through the cache / scratch
            assert torch.equal(got, want)


@pytest.mark.parametrize("K", [64, 128, 320])
@pytest.mark.parametrize("tiles_m", [1, 2, 7, 8, 9, 16, 33, 64, 65, 257, 511, 520, 771])
def test_gemm_pairs_cover_every_tile_once(ctx, gpu_device, tiles_m, K):
    """two tiles per workgroup: every XCD run length around the pair / single split (empty runs, a lone tile, odd and
    even runs, exactly one round, one round + 1, several rounds + a short one) writes every output tile exactly once —
    with the ring carried across the pair (K >= 128: two and five K tiles, the slot sequence wraps inside the second
    tile) and restarted (K = 64: a single K tile cannot carry)"""
    bm, N = 64, 256
    M = tiles_m * bm - 5
    g = torch.Generator().manual_seed(tiles_m)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    ref = A @ W.t()
    outs = []
    for pairs in (2, 1, 0):
        out = torch.full((M + 8, N), 7.0, dtype=torch.float32, device=gpu_device)
        with gemm_knobs(ctx, tile=bm, pairs=pairs):
            ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out[:M], flags=capi.GEMM_OUT_F32)
            sync()
        assert (out[:M].cpu() - ref).abs().max() <= 2e-3, pairs
        assert (out[M:] == 7.0).all()
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("K", [64, 128, 192, 320, 1024])
@pytest.mark.parametrize("tiles_m", [1, 2, 9, 65, 520])
@pytest.mark.parametrize("bm", [128, 64])
def test_gemm_three_stage_ring(ctx, gpu_device, bm, tiles_m, K):
    """Tiles of 128 / 64 rows run on a three-stage LDS ring (all of K tile t+2 issued during K tile t), the taller ones on
    the five-slot split ring: one, two, three, five and 16 K tiles (the stage index wraps; with pairs the ring is carried
    into the second tile), every run-length class of the pair split; bit-equal to the 192-row tiles of the split ring"""
    N = 512
    M = tiles_m * bm - 7
    g = torch.Generator().manual_seed(tiles_m + K)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    ref = A @ W.t() + bias
    outs = []
    for tile, pairs in ((192, 2), (bm, 2), (bm, 1), (bm, 0)):
        out = torch.full((M + 8, N), 7.0, dtype=torch.float32, device=gpu_device)
        with gemm_knobs(ctx, tile=tile, pairs=pairs):
            ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out[:M], flags=capi.GEMM_BIAS | capi.GEMM_OUT_F32, bias=bias.to(gpu_device))
            sync()
        assert (out[:M].cpu() - ref).abs().max() <= 2e-3, (tile, pairs)
        assert (out[M:] == 7.0).all()
        outs.append(out)
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("tile", [0, 64, 256])
@pytest.mark.parametrize("M", [35328 - 37, 300])
def test_gemm_glu_epilogue(ctx, gpu_device, M, tile):
    """RS_GEMM_GLU: value / gate columns interleaved in blocks of 32 (the loader's pw1 row order), GLU applied to the
    f32 accumulators, bf16 [M][N/2] out."""
    from reazonspeech_amd.runtime.weights import glu_interleave_index
    d, K = 1024, 1024
    g = torch.Generator().manual_seed(M)
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((2 * d, K), generator=g) / K ** 0.5)
    bias = torch.randn((2 * d,), generator=g)
    rows = torch.cat([torch.arange(0, min(256, M)), torch.arange(M - 44, M)])
    y = A[rows] @ W.t() + bias
    ref = y[:, :d] * torch.sigmoid(y[:, d:])
    idx = glu_interleave_index(d)
    out = torch.full((M + 64, d), 7.0, dtype=torch.bfloat16, device=gpu_device)
    with gemm_knobs(ctx, tile=tile):
        ctx.gemm(bf(A).to(gpu_device), bf(W[idx]).to(gpu_device), out[:M], flags=capi.GEMM_BIAS | capi.GEMM_GLU,
                 bias=bias[idx].to(gpu_device))
        sync()
    got = out[rows.to(gpu_device)].float().cpu()
    bad = (got - ref).abs() > 2e-3 + 2.0 ** -8 * ref.abs()
    assert not bad.any(), (int(bad.sum()), float((got - ref).abs().max()))
    assert (out[M:].float() == 7.0).all(), "rows past M were written"
    with pytest.raises(capi.RsError):        # GLU takes a bias only
        ctx.gemm(bf(A).to(gpu_device), bf(W[idx]).to(gpu_device), out[:M], flags=capi.GEMM_GLU | capi.GEMM_SILU)


def test_gemm_big_rowmask(ctx, gpu_device):
    """the subsampling pointwise GEMM at its benchmark-like size (256-row tiles, N = one tile wide): bias + ReLU + the
    per-utterance row mask (rows of frames at or past an utterance's length are zeroed), ragged last tile"""
    g = torch.Generator().manual_seed(17)
    B, T, Fq, K, N = 37, 275, 20, 256, 256
    M = B * T * Fq
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    lens = torch.randint(0, T + 1, (B,), generator=g, dtype=torch.int32)
    lens[0], lens[1] = T, 0
    out = torch.full((M + 64, N), 7.0, dtype=torch.bfloat16, device=gpu_device)
    ctx.gemm(bf(A).to(gpu_device), bf(W).to(gpu_device), out[:M], flags=capi.GEMM_BIAS | capi.GEMM_RELU | capi.GEMM_ROWMASK,
             bias=bias.to(gpu_device), mask_lens=lens.to(gpu_device), mask_rows=Fq, mask_steps=T)
    sync()
    for b in (0, 1, 5, 36):
        rows = slice(b * T * Fq, (b + 1) * T * Fq)
        ref = torch.relu(A[rows] @ W.t() + bias).view(T, Fq, N)
        ref[int(lens[b]):] = 0
        got = out[rows].float().cpu().view(T, Fq, N)
        assert ((got - ref).abs() <= 2e-3 + 2.0 ** -8 * ref.abs()).all(), b
    assert (out[M:].float() == 7.0).all(), "rows past M were written"


def test_gemm_epilogues(ctx, gpu_device):
    g = torch.Generator().manual_seed(5)
    B, T, Fq, K, N = 3, 11, 5, 128, 192
    M = B * T * Fq
    A = rb(torch.randn((M, K), generator=g))
    W = rb(torch.randn((N, K), generator=g) / K ** 0.5)
    bias = torch.randn((N,), generator=g)
    res = torch.randn((M, N), generator=g)
    lens = torch.tensor([11, 4, 0], dtype=torch.int32)
    dA, dW, dbias = bf(A).to(gpu_device), bf(W).to(gpu_device), bias.to(gpu_device)
    base = A @ W.t() + bias
    # bias + SiLU -> bf16
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=gpu_device)
    ctx.gemm(dA, dW, out, flags=capi.GEMM_BIAS | capi.GEMM_SILU, bias=dbias)
    sync()
    ref = torch.nn.functional.silu(base)
    assert (out.cpu().float() - ref).abs().max() <= 2e-2
    # bias, *0.5, + residual, in place on an f32 stream
    stream = res.clone().to(gpu_device)
    ctx.gemm(dA, dW, stream, flags=capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, bias=dbias, alpha=0.5,
             residual=stream)
    sync()
    assert (stream.cpu() - (res + 0.5 * base)).abs().max() <= 2e-3
    # bias + ReLU + per-utterance row mask -> bf16
    out = torch.ones((M, N), dtype=torch.bfloat16, device=gpu_device)
    ctx.gemm(dA, dW, out, flags=capi.GEMM_BIAS | capi.GEMM_RELU | capi.GEMM_ROWMASK, bias=dbias,
             mask_lens=lens.to(gpu_device), mask_rows=Fq, mask_steps=T)
    sync()
    mask = (torch.arange(T)[None, :] < lens[:, None]).float()[:, :, None, None].expand(B, T, Fq, N).reshape(M, N)
    ref = torch.relu(base) * mask
    got = out.cpu().float()
    assert (got - ref).abs().max() <= 2e-2
    assert torch.all(got[mask == 0] == 0)


def test_gemm_rejects_bad_k(ctx, gpu_device):
    A = torch.zeros((4, 48), dtype=torch.bfloat16, device=gpu_device)
    W = torch.zeros((8, 48), dtype=torch.bfloat16, device=gpu_device)
    out = torch.zeros((4, 8), dtype=torch.float32, device=gpu_device)
    with pytest.raises(capi.RsError):
        ctx.gemm(A, W, out, flags=capi.GEMM_OUT_F32)


@pytest.mark.parametrize("d", [256, 1024])
def test_layernorm(ctx, gpu_device, d):
    g = torch.Generator().manual_seed(d)
    M = 77
    x = torch.randn((M, d), generator=g) * 3 + 1.5
    gamma = 1 + 0.1 * torch.randn((d,), generator=g)
    beta = 0.1 * torch.randn((d,), generator=g)
    ref = torch.nn.functional.layer_norm(x, (d,), gamma, beta, 1e-5)
    o32 = torch.zeros((M, d), dtype=torch.float32, device=gpu_device)
    o16 = torch.zeros((M, d), dtype=torch.bfloat16, device=gpu_device)
    dx = x.to(gpu_device)
    ctx.layernorm(dx, gamma.to(gpu_device), beta.to(gpu_device), 1e-5, out_bf16=o16, out_f32=o32)
    sync()
    assert (o32.cpu() - ref).abs().max() <= 2e-5
    assert (o16.cpu().float() - ref).abs().max() <= 2e-2
    # in place (how the encoder applies norm_out to the residual stream)
    ctx.layernorm(dx, gamma.to(gpu_device), beta.to(gpu_device), 1e-5, out_f32=dx)
    sync()
    assert (dx.cpu() - ref).abs().max() <= 2e-5


@pytest.mark.parametrize("T,lens", [(19, [19, 14]), (70, [70, 33, 1]), (138, [138, 100])])
def test_glu_dwconv_silu(ctx, gpu_device, T, lens):
    g = torch.Generator().manual_seed(T)
    B, d, k = len(lens), 256, 9
    x = rb(torch.randn((B, T, 2 * d), generator=g))
    w = torch.randn((d, k), generator=g) / 3
    b = 0.1 * torch.randn((d,), generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int32)
    a, gate = x[..., :d], x[..., d:]
    u = a * torch.sigmoid(gate) * (torch.arange(T)[None, :] < lens_t[:, None])[:, :, None]
    z = torch.nn.functional.conv1d(u.transpose(1, 2), w[:, None, :], b, padding=4, groups=d).transpose(1, 2)
    ref = torch.nn.functional.silu(z)
    out = torch.zeros((B * T, d), dtype=torch.bfloat16, device=gpu_device)
    ctx.glu_dwconv(bf(x).reshape(B * T, 2 * d).to(gpu_device), w.t().contiguous().to(gpu_device), b.to(gpu_device),
                   lens_t.to(gpu_device), B, T, d, k, out)
    sync()
    assert (out.cpu().float().view(B, T, d) - ref).abs().max() <= 2e-2
    # the other input layouts: value / gate columns interleaved in blocks of 32 (what the pw1 GEMM produces from the
    # loader's row order) and GLU already applied by the GEMM epilogue (bf16 [B*T][d])
    from reazonspeech_amd.runtime.weights import glu_interleave_index
    out1 = torch.zeros_like(out)
    ctx.glu_dwconv(bf(x[..., glu_interleave_index(d)]).reshape(B * T, 2 * d).to(gpu_device), w.t().contiguous().to(gpu_device),
                   b.to(gpu_device), lens_t.to(gpu_device), B, T, d, k, out1, layout=capi.GLU_BLOCK32)
    sync()
    assert torch.equal(out1, out)
    ug = rb(a * torch.sigmoid(gate))
    um = ug * (torch.arange(T)[None, :] < lens_t[:, None])[:, :, None]
    ref2 = torch.nn.functional.silu(torch.nn.functional.conv1d(um.transpose(1, 2), w[:, None, :], b, padding=4, groups=d).transpose(1, 2))
    out2 = torch.zeros_like(out)
    ctx.glu_dwconv(bf(ug).reshape(B * T, d).to(gpu_device), w.t().contiguous().to(gpu_device), b.to(gpu_device),
                   lens_t.to(gpu_device), B, T, d, k, out2, layout=capi.GLU_APPLIED)
    sync()
    assert (out2.cpu().float().view(B, T, d) - ref2).abs().max() <= 2e-2


@pytest.mark.parametrize("T,lens,window", [(19, [19, 14], None), (138, [138, 97, 5], None), (64, [64, 33], None),
                                           (300, [300, 160], None), (138, [138, 60], (32, 16, 1)),
                                           # limited context over many key chunks: block / chunk skipping (O(T * W)), with
                                           # and without global tokens, asymmetric windows, a window wider than a chunk
                                           (1400, [1400, 777], (128, 128, 1)), (1000, [1000, 333], (40, 200, 0)),
                                           (900, [900, 650], (0, 0, 3)), (700, [700, 512], (300, 17, 0))])
@pytest.mark.parametrize("heads", [2, 4])
def test_relpos_attention(gpu_device, T, lens, window, heads):
    """heads = 2: head_dim 128 (FastConformer-XL), heads = 4: head_dim 64 (the ESPnet Conformer's 512 / 8) — one kernel
    template, two geometries"""
    cfg = TINY.with_(n_heads=heads)
    if window is not None:
        cfg = cfg.with_(att_left=window[0], att_right=window[1], n_global=window[2])
    c = capi.Context(cfg, 0)
    g = torch.Generator().manual_seed(T + len(lens))
    B, H, dh, d = len(lens), cfg.n_heads, cfg.head_dim, cfg.d_model
    qkv = rb(torch.randn((B, T, 3 * d), generator=g))
    p = rb(torch.randn((2 * T - 1, d), generator=g))
    bu = 0.3 * torch.randn((H, dh), generator=g)
    bv = 0.3 * torch.randn((H, dh), generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, T, H, dh) for i in range(3))
    ref = om.attention_core(cfg, q, k, v, p.view(2 * T - 1, H, dh), bu, bv, lens_t, "bf16")
    out = torch.full((B * T, d), 3.0, dtype=torch.bfloat16, device=gpu_device)
    c.attention(bf(qkv).reshape(B * T, 3 * d).to(gpu_device), bf(p).to(gpu_device), bu.reshape(-1).to(gpu_device),
                bv.reshape(-1).to(gpu_device), lens_t.to(torch.int32).to(gpu_device), B, T, out)
    sync()
    got = out.cpu().float().view(B, T, d)
    for b in range(B):
        n = lens[b]
        err = (got[b, :n] - ref[b, :n]).abs().max().item()
        assert err <= 2e-2, (b, err)
        assert torch.all(got[b, n:] == 0)      # padded queries: NeMo zero-fills
    c.close()


@pytest.mark.parametrize("form", ["streaming", "persistent"])
@pytest.mark.parametrize("T,lens", [(19, [19, 14]), (138, [138, 97, 5, 0]), (300, [300, 161, 32])])
@pytest.mark.parametrize("heads", [2, 4])
def test_relpos_attention_alternative_forms(gpu_device, T, lens, heads, form):
    """the two forms of full attention that round 6 built and measured SLOWER than the default (csrc/k_attention.hip: the
    streaming kernel — 16-query waves, K / V through a DMA ring, V^T by ds_read_b64_tr_b16 — and the staged kernel on resident
    workgroups; profiles/r06_10_*): off by default, kept correct.  streaming: the oracle's tolerance (another summation
    order); persistent: bit-identical to the default"""
    cfg = TINY.with_(n_heads=heads)
    c = capi.Context(cfg, 0)
    g = torch.Generator().manual_seed(T + len(lens))
    B, H, dh, d = len(lens), cfg.n_heads, cfg.head_dim, cfg.d_model
    qkv = rb(torch.randn((B, T, 3 * d), generator=g))
    p = rb(torch.randn((2 * T - 1, d), generator=g))
    bu = 0.3 * torch.randn((H, dh), generator=g)
    bv = 0.3 * torch.randn((H, dh), generator=g)
    lens_t = torch.tensor(lens, dtype=torch.int64)
    q, k, v = (qkv[..., i * d:(i + 1) * d].reshape(B, T, H, dh) for i in range(3))
    ref = om.attention_core(cfg, q, k, v, p.view(2 * T - 1, H, dh), bu, bv, lens_t, "bf16")
    args = (bf(qkv).reshape(B * T, 3 * d).to(gpu_device), bf(p).to(gpu_device), bu.reshape(-1).to(gpu_device),
            bv.reshape(-1).to(gpu_device), lens_t.to(torch.int32).to(gpu_device), B, T)
    base = torch.full((B * T, d), 3.0, dtype=torch.bfloat16, device=gpu_device)
    c.attention(*args, base)
    out = torch.full((B * T, d), 3.0, dtype=torch.bfloat16, device=gpu_device)
    hook = c.lib.rs_debug_set_attn_stream if form == "streaming" else c.lib.rs_debug_set_attn_persist
    hook(1 if form == "streaming" else 2)             # persistent: two resident workgroups walk all the items
    try:
        c.attention(*args, out)
        sync()
    finally:
        hook(0)
    if form == "persistent":
        assert torch.equal(out, base)
    got = out.cpu().float().view(B, T, d)
    for b in range(B):
        n = lens[b]
        if n:
            assert (got[b, :n] - ref[b, :n]).abs().max().item() <= 2e-2, b
        assert torch.all(got[b, n:] == 0)
    c.close()


def test_rejects_wrong_head_dim(gpu_device):
    with pytest.raises(capi.RsError):
        capi.Context(TINY.with_(n_heads=8), 0)            # head_dim 32: 128 and 64 are built
