"""Same-box A/B of the HBM-bound encoder kernels at the benchmark geometry (B=256, T'=138, d=1024).

    python scripts/elementwise_bench.py
Prints microseconds and achieved HBM rate (algorithmic bytes / time) for the conv-module middle
(generic vs register-window kernel) and LayerNorm.
"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M as cfg

B, T, d, k = 256, 138, cfg.d_model, cfg.conv_kernel
dev = torch.device("cuda", 0)
ctx = capi.Context(cfg, 0)
lib = ctx.lib
lib.rs_debug_set_glu_generic.argtypes = [ctypes.c_int]
g = torch.Generator().manual_seed(0)


def timed(fn, n=7, inner=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / inner)
    return sorted(ts)[len(ts) // 2]


x = torch.randn((B * T, 2 * d), generator=g).to(torch.bfloat16).to(dev)
w = (0.3 * torch.randn((k, d), generator=g)).to(dev)
bias = (0.1 * torch.randn(d, generator=g)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
outs = []
for generic in (1, 0, 1, 0):
    lib.rs_debug_set_glu_generic(generic)
    out = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)
    us = timed(lambda: ctx.glu_dwconv(x, w, bias, lens, B, T, d, k, out))
    outs.append(out.float().cpu())
    print(f"glu_dwconv_silu {'generic' if generic else 'window '}: {us:7.1f} us  {B * T * d * 6 / us / 1e6:5.2f} TB/s")
print("max |generic - window| =", (outs[0] - outs[1]).abs().max().item())

# the gated layout (GLU applied by the pw1 GEMM: the product default): DMA-staged bf16 tile vs the f32-tile kernel
xg = torch.randn((B * T, d), generator=g).to(torch.bfloat16).to(dev)
outs = []
for mode in (2, 0, 2, 0):
    lib.rs_debug_set_glu_generic(mode)
    out = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)
    us = timed(lambda: ctx.glu_dwconv(xg, w, bias, lens, B, T, d, k, out, layout=capi.GLU_APPLIED))
    outs.append(out.float().cpu())
    print(f"dwconv_silu (gated in) {'f32 tile ' if mode else 'dma bf16 '}: {us:7.1f} us  {B * T * d * 4 / us / 1e6:5.2f} TB/s")
print("max |f32 tile - dma| =", (outs[0] - outs[1]).abs().max().item())
lib.rs_debug_set_glu_generic(0)

xf = torch.randn((B * T, d), generator=g).to(dev)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
ob = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)
of = torch.empty((B * T, d), dtype=torch.float32, device=dev)
us = timed(lambda: ctx.layernorm(xf, gamma, beta, 1e-5, out_bf16=ob))
print(f"layernorm f32->bf16      : {us:7.1f} us  {B * T * d * 6 / us / 1e6:5.2f} TB/s")
us = timed(lambda: ctx.layernorm(xf, gamma, beta, 1e-5, out_bf16=ob, out_f32=of))
print(f"layernorm f32->f32+bf16  : {us:7.1f} us  {B * T * d * 10 / us / 1e6:5.2f} TB/s")
