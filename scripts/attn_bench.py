"""Time the rel-pos attention kernel at the benchmark geometry (B = 256, T' = 138, 8 heads; run on the GPU box).

    RS_ATTN_SKEW_CYCLES=<n> python scripts/attn_bench.py      (the knob is read once per process)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi                       # noqa: E402
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M as cfg  # noqa: E402

B, T, d, H = 256, 138, cfg.d_model, cfg.n_heads
dev = torch.device("cuda", 0)
ctx = capi.Context(cfg, 0)
g = torch.Generator().manual_seed(0)
qkv = torch.randn((B * T, 3 * d), generator=g).to(torch.bfloat16).to(dev)
pos = torch.randn((2 * T - 1, d), generator=g).to(torch.bfloat16).to(dev)
bu = (0.3 * torch.randn(d, generator=g)).to(dev)
bv = (0.3 * torch.randn(d, generator=g)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
out = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)
ref = None
ts = []
for rep in range(9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        ctx.attention(qkv, pos, bu, bv, lens, B, T, out)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 8 * 1e3)
ts.sort()
print(f"RS_ATTN_SKEW_CYCLES={os.environ.get('RS_ATTN_SKEW_CYCLES', '0'):>6}: attention {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})  checksum {out.float().sum().item():.3f}")
