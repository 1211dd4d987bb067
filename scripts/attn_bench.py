"""Rel-pos attention, three forms on one box at the geometries of the two encoders that use it (FastConformer: 8 heads of 128,
T' = 138; ESPnet Conformer: 8 heads of 64, T' = 358), full and ragged batches:
  persistent   the staged kernel on resident workgroups that walk their items (default)
  classic      the staged kernel, one workgroup per (query group, head, utterance)
  streaming    the round-6 rewrite (16-query waves, K / V through a DMA ring, V^T by ds_read_b64_tr_b16; $RS_ATTN_STREAM=1)

    python scripts/attn_bench.py

Per geometry: median launch time of each form (interleaved, 9 repetitions of 8 launches), whether persistent == classic bit for
bit, the largest difference streaming - classic (a different summation order: rounding, not bits) and each form's error against
a float32 torch reference of the same inputs on three utterances.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi                       # noqa: E402
from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M, FASTCONFORMER_619M  # noqa: E402

dev = torch.device("cuda", 0)


def reference(qkv, pos, bu, bv, n, T, H, dh, b):
    """float32, the kernel's rounding points: (q + u), (q + v) and the probabilities to bf16"""
    d = H * dh
    x = qkv.view(-1, T, 3 * d)[b].float()
    q, k, v = (x[:, i * d:(i + 1) * d].view(T, H, dh) for i in range(3))
    p = pos.float().view(2 * T - 1, H, dh)
    qu = (q + bu.view(H, dh)).bfloat16().float()
    qv = (q + bv.view(H, dh)).bfloat16().float()
    ac = torch.einsum("ihd,jhd->hij", qu, k)
    bd_full = torch.einsum("ihd,nhd->hin", qv, p)
    idx = (torch.arange(T, device=x.device)[None, :] - torch.arange(T, device=x.device)[:, None] + T - 1)
    bd = torch.gather(bd_full, 2, idx[None].expand(H, T, T))
    s = (ac + bd) / dh ** 0.5
    s[:, :, n:] = float("-inf")
    w = torch.softmax(s, dim=-1)
    w = w / 1.0                                           # (the kernel rounds exp(s - m) to bf16, not the normalised weight)
    o = torch.einsum("hij,jhd->ihd", w, v).reshape(T, d)
    o[n:] = 0
    return o


def run(cfg, B, T, ragged, label):
    d, H = cfg.d_model, cfg.n_heads
    ctx = capi.Context(cfg, 0)
    g = torch.Generator().manual_seed(B + T)
    qkv = torch.randn((B * T, 3 * d), generator=g).to(torch.bfloat16).to(dev)
    pos = torch.randn((2 * T - 1, d), generator=g).to(torch.bfloat16).to(dev)
    bu = (0.3 * torch.randn(d, generator=g)).to(dev)
    bv = (0.3 * torch.randn(d, generator=g)).to(dev)
    if ragged:
        lens = torch.randint(T // 5, T + 1, (B,), generator=g).to(torch.int32)
        lens[0] = T
    else:
        lens = torch.full((B,), T, dtype=torch.int32)
    lens_d = lens.to(dev)
    forms = {"persistent": (0, 1), "classic": (0, 0), "streaming": (1, 0)}

    def select(name):
        ctx.lib.rs_debug_set_attn_stream(forms[name][0])
        ctx.lib.rs_debug_set_attn_persist(forms[name][1])

    outs, ts = {}, {f: [] for f in forms}
    for f in forms:
        select(f)
        outs[f] = torch.full((B * T, d), 3.0, dtype=torch.bfloat16, device=dev)
        ctx.attention(qkv, pos, bu, bv, lens_d, B, T, outs[f])
    torch.cuda.synchronize()
    scratch = torch.empty_like(outs["classic"])
    for rep in range(9):
        for f in forms:
            select(f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ctx.attention(qkv, pos, bu, bv, lens_d, B, T, scratch)
            e1.record()
            torch.cuda.synchronize()
            ts[f].append(e0.elapsed_time(e1) / 8 * 1e3)
    select("persistent")
    same = torch.equal(outs["persistent"], outs["classic"])
    diff = (outs["streaming"].float() - outs["classic"].float()).abs().max().item()
    errs = {f: 0.0 for f in forms}
    for b in sorted({0, B // 2, B - 1}):
        ref = reference(qkv, pos, bu, bv, int(lens[b]), T, H, d // H, b)
        for f in forms:
            errs[f] = max(errs[f], (outs[f].view(B, T, d)[b].float() - ref).abs().max().item())
    med = {f: sorted(ts[f])[len(ts[f]) // 2] for f in ts}
    print(f"{label}: " + " | ".join(f"{f} {med[f]:7.1f} us (min {min(ts[f]):.1f})" for f in forms) +
          f" | persistent == classic bit for bit: {same}; max |streaming - classic| {diff:.4f}; max error vs float32 torch on 3 utterances: "
          + ", ".join(f"{f} {errs[f]:.4f}" for f in forms), flush=True)
    ctx.close()
    return same and max(errs.values()) <= 2e-2 and diff <= 3e-2


ok = True
ok &= run(FASTCONFORMER_619M, 256, 138, False, "FastConformer 8 x 128, B = 256, T' = 138")
ok &= run(FASTCONFORMER_619M, 256, 138, True, "FastConformer 8 x 128, B = 256, T' <= 138 ragged")
ok &= run(FASTCONFORMER_619M, 32, 138, False, "FastConformer 8 x 128, B = 32, T' = 138")
ok &= run(FASTCONFORMER_619M, 8, 1500, True, "FastConformer 8 x 128, B = 8, T' <= 1500 ragged")
ok &= run(ESPNET_CONFORMER_120M, 256, 358, False, "ESPnet Conformer 8 x 64, B = 256, T' = 358")
ok &= run(ESPNET_CONFORMER_120M, 256, 358, True, "ESPnet Conformer 8 x 64, B = 256, T' <= 358 ragged")
assert ok
