"""Per-tile timeline of the GEMM kernel (gemm_smf16_kernel, TRACE build with s_memtime stamps).

    python scripts/gemm_trace.py [--shape=qkv]

For every output tile, wave 0 (wave group 0) and wave 4 (wave group 1) record: prologue (launch -> first K tile
landed), main loop, epilogue issue, store drain (s_waitcnt vmcnt(0) after the last store), the cycles spent inside
the per-K-tile `s_waitcnt vmcnt(0)` of the main loop, and the chip-wide 100 MHz wall clock at start / end.
Printed in microseconds (shader cycles converted with the clock measured from the two time bases).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi                       # noqa: E402
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M  # noqa: E402

M = 35328
SHAPES = [
    ("qkv     bias->bf16", 3072, 1024, capi.GEMM_BIAS, 256),
    ("ffn_up  silu->bf16", 4096, 1024, capi.GEMM_BIAS | capi.GEMM_SILU, 256),
    ("out/pw2 res->f32  ", 1024, 1024, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, 192),
    ("ffn_down res->f32 ", 1024, 4096, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32, 192),
]


def main():
    scheds = [0]
    only = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--shape=")]
    ctx = capi.Context(FASTCONFORMER_619M, 0)
    lib = ctx.lib
    lib.rs_debug_set_gemm_tile.argtypes = [ctypes.c_int]
    lib.rs_debug_set_gemm_trace.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda", 0)
    for name, n, k, flags, bm in SHAPES:
        if only and not any(o in name for o in only):
            continue
        for v in scheds:
            lib.rs_debug_set_gemm_tile(bm)
            A = torch.randn((M, k), device=dev).to(torch.bfloat16)
            W = (torch.randn((n, k), device=dev) / k ** 0.5).to(torch.bfloat16)
            bias = torch.randn((n,), device=dev)
            res = torch.randn((M, n), device=dev) if flags & capi.GEMM_RESIDUAL else None
            out = torch.empty((M, n), dtype=torch.float32 if flags & capi.GEMM_OUT_F32 else torch.bfloat16, device=dev)
            for _ in range(3):
                ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
            torch.cuda.synchronize()
            ntiles = ((M + bm - 1) // bm) * ((n + 255) // 256)
            tr = torch.zeros((ntiles * 2, 8), dtype=torch.int64, device=dev)
            lib.rs_debug_set_gemm_trace(ctypes.c_void_p(tr.data_ptr()))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
            e1.record()
            torch.cuda.synchronize()
            lib.rs_debug_set_gemm_trace(None)
            t = tr.cpu().numpy().astype(np.float64).reshape(ntiles, 2, 8)
            ok = t[:, 0, 7] > 0
            t = t[ok]
            total_us = e0.elapsed_time(e1) * 1e3
            w_min = t[:, :, 6].min()
            w0 = (t[:, 0, 6] - w_min) / 100.0
            w1 = (t[:, 0, 7] - w_min) / 100.0
            cyc = t[:, 0, 0] + t[:, 0, 1] + t[:, 0, 2] + t[:, 0, 3]
            cyc_per_us = (cyc / np.maximum(w1 - w0, 1e-3)).mean()
            mfma_floor = 2.0 * bm * 256 * k / (2.5e15 / 256) * 1e6
            print(f"== {name}: {len(t)} tiles of {bm}x256, traced launch {total_us:.1f} us, shader clock ~{cyc_per_us / 1e3:.2f} GHz, "
                  f"MFMA floor per tile {mfma_floor:.1f} us @2.4 GHz")
            for g in (0, 1):
                print(f"   wave group {g}:")
                for label, col in (("prologue", 0), ("main loop", 1), ("epilogue issue", 2), ("store drain", 3), ("  K-tile wait stall", 4)):
                    x = t[:, g, col] / cyc_per_us
                    print(f"      {label:20s} mean {x.mean():7.2f}  p10 {np.percentile(x, 10):7.2f}  p50 {np.percentile(x, 50):7.2f}  p90 {np.percentile(x, 90):7.2f} us")
            tot = w1 - w0
            print(f"   tile wall time          mean {tot.mean():7.2f}  p10 {np.percentile(tot, 10):7.2f}  p50 {np.percentile(tot, 50):7.2f}  p90 {np.percentile(tot, 90):7.2f} us")
            print(f"   sum of tile times {tot.sum():.0f} us = {tot.sum() / w1.max():.1f} resident tiles on average; kernel span {w1.max():.1f} us")
            hist, _ = np.histogram(w0, bins=20, range=(0, w1.max()))
            print("   tile starts per 5% of the span:", hist.tolist())
    lib.rs_debug_set_gemm_tile(0)


if __name__ == "__main__":
    main()
