"""Measure the fixed (prologue + epilogue) cost of the GEMM kernel: K=64 problems per epilogue kind."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M
M = 35328
ctx = capi.Context(FASTCONFORMER_619M, 0)
setv = ctx.lib.rs_debug_set_gemm_variant; setv.argtypes = [ctypes.c_int]; setv.restype = None
dev = torch.device("cuda", 0)
sets = ctx.lib.rs_debug_set_gemm_skew; sets.argtypes = [ctypes.c_int]; sets.restype = None
for spec in sys.argv[1:] or ["2"]:
    v, sk = (spec.split(":") + ["-1"])[:2]
    v = int(v); setv(v); sets(int(sk)); v = spec
    for name, n, flags in [("plain->bf16", 4096, 0), ("silu->bf16", 4096, capi.GEMM_BIAS | capi.GEMM_SILU),
                           ("res->f32", 1024, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32)]:
        for k in (64, 1024, 4096):
            A = torch.randn((M, k), device=dev).to(torch.bfloat16)
            W = torch.randn((n, k), device=dev).to(torch.bfloat16)
            bias = torch.randn((n,), device=dev)
            res = torch.randn((M, n), device=dev) if flags & capi.GEMM_RESIDUAL else None
            out = torch.empty((M, n), dtype=torch.float32 if flags & capi.GEMM_OUT_F32 else torch.bfloat16, device=dev)
            for _ in range(2): ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"v{v} {name:12s} N{n} K{k:5d}: {us:8.1f} us  {2.0*M*n*k/us/1e6:7.1f} TF", flush=True)
