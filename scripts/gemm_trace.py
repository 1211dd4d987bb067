"""Per-tile timeline of one GEMM launch (debug build of the kernel with s_memtime stamps).

    python scripts/gemm_trace.py [variant] -> phase statistics in microseconds (2.1 GHz assumed for
    cycles -> us; only ratios matter)
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M

M = 35328
ctx = capi.Context(FASTCONFORMER_619M, 0)
lib = ctx.lib
lib.rs_debug_set_gemm_variant.argtypes = [ctypes.c_int]
lib.rs_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
lib.rs_debug_set_gemm_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
dev = torch.device("cuda", 0)
for name, n, k, flags in [("ffn_up silu->bf16", 4096, 1024, capi.GEMM_BIAS | capi.GEMM_SILU),
                          ("plain->bf16 K64", 4096, 64, 0),
                          ("ffn_down res->f32", 1024, 4096, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32)]:
    A = torch.randn((M, k), device=dev).to(torch.bfloat16)
    W = (torch.randn((n, k), device=dev) / k ** 0.5).to(torch.bfloat16)
    bias = torch.randn((n,), device=dev)
    res = torch.randn((M, n), device=dev) if flags & capi.GEMM_RESIDUAL else None
    out = torch.empty((M, n), dtype=torch.float32 if flags & capi.GEMM_OUT_F32 else torch.bfloat16, device=dev)
    for _ in range(3):
        ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
    torch.cuda.synchronize()
    ntiles = ((M + 255) // 256) * ((n + 255) // 256)
    tr = torch.zeros((ntiles, 8), dtype=torch.int64, device=dev)
    lib.rs_debug_set_gemm_trace(ctypes.c_void_p(tr.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res); e1.record()
    torch.cuda.synchronize()
    lib.rs_debug_set_gemm_trace(None)
    t = tr.cpu().numpy().astype(np.float64)
    t = t[t[:, 7] > 0]
    total_us = e0.elapsed_time(e1) * 1e3
    w0 = (t[:, 6] - t[:, 6].min()) / 100.0          # us (100 MHz wall clock)
    w1 = (t[:, 7] - t[:, 6].min()) / 100.0
    cyc_per_us = (t[:, 4] / np.maximum(w1 - w0, 1e-3)).mean()   # shader cycles per wall us
    print(f"== {name}: {len(t)} tiles, launch {total_us:.1f} us, shader clock ~{cyc_per_us / 1e3:.2f} GHz")
    prev = 0
    for label, b in (("wait first stage", 1), ("main loop", 2), ("epilogue issue", 3), ("store drain", 4)):
        x = (t[:, b] - t[:, prev if b > 1 else 0]) / cyc_per_us if b > 1 else t[:, 1] / cyc_per_us
        x = (t[:, b] - (t[:, b - 1] if b > 1 else 0)) / cyc_per_us
        print(f"   {label:18s} mean {x.mean():7.2f}  p10 {np.percentile(x, 10):7.2f}  p50 {np.percentile(x, 50):7.2f}  p90 {np.percentile(x, 90):7.2f} us")
    tot = w1 - w0
    print(f"   tile wall time     mean {tot.mean():7.2f}  p10 {np.percentile(tot, 10):7.2f}  p50 {np.percentile(tot, 50):7.2f}  p90 {np.percentile(tot, 90):7.2f} us")
    print(f"   sum of tile times {tot.sum():.0f} us = {tot.sum() / (w1.max()):.1f} resident tiles on average; kernel span {w1.max():.1f} us")
    # how synchronised are the CUs?  histogram of tile start times
    hist, edges = np.histogram(w0, bins=20, range=(0, w1.max()))
    print("   tile starts per 5% of the span:", hist.tolist())
