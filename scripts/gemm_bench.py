"""A/B the GEMM tile heights / main-loop schedules on the shapes of the 619M encoder (run on the GPU box).

    python scripts/gemm_bench.py [TILE[p0|p1] ...] [--batch=32] [--shape=ffn] [--group-m=N] [--quick]
TILE = 0 (what the launcher picks), 256, 192, 128, 64; suffix p0 = one tile per workgroup, p1 = pairs (ring restarted), p2 (default) = pairs with the ring carried over;
a trailing b = the register-resident weight form ($RS_GEMM_BREG: fragment-major weights global -> VGPR, only A in the LDS ring), e.g. "0 0b".  Prints per shape and variant: correctness vs a torch bf16 matmul, median
microseconds, TFLOP/s.  The variants are interleaved per shape inside one process (guide §5.4 rule 24).
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi                     # noqa: E402
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M  # noqa: E402

BATCH = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--batch=")] or [256])[0]
M = BATCH * 138
SHAPES = [  # name, M, N, K, flags
    ("ffn_up  silu->bf16", M, 4096, 1024, capi.GEMM_BIAS | capi.GEMM_SILU),
    ("ffn_down res->f32 ", M, 1024, 4096, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32),
    ("qkv     bias->bf16", M, 3072, 1024, capi.GEMM_BIAS),
    ("out/pw2 res->f32  ", M, 1024, 1024, capi.GEMM_BIAS | capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32),
    ("pw1     bias->bf16", M, 2048, 1024, capi.GEMM_BIAS),
    ("pw1     glu ->bf16", M, 2048, 1024, capi.GEMM_BIAS | capi.GEMM_GLU),
    ("sub_pw1 relu->bf16", BATCH * 275 * 20, 256, 256, capi.GEMM_BIAS | capi.GEMM_RELU),
    ("sub_pw2 relu->bf16", BATCH * 138 * 10, 256, 256, capi.GEMM_BIAS | capi.GEMM_RELU),
    ("sub_out ->f32     ", M, 1024, 2560, capi.GEMM_BIAS | capi.GEMM_OUT_F32),
]


def main():
    quick = "--quick" in sys.argv
    def parse(v):
        breg = v.endswith("b")
        v = v.rstrip("b")
        return (int(v.split("p")[0]), int(v.split("p")[1]) if "p" in v else 2, int(breg))
    variants = [parse(v) for v in sys.argv[1:] if not v.startswith("--")] or [(0, 2, 0)]
    groups = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--group-m=")] or [None]
    # row pitch of A / W in elements beyond K (power-of-two pitches can camp on a few L2 / HBM channels)
    pad_a = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--pad-a=")] or [0])[0]
    pad_w = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--pad-w=")] or [0])[0]
    pad_c = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--pad-c=")] or [0])[0]
    only = [a.split("=")[1] for a in sys.argv[1:] if a.startswith("--shape=")]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(FASTCONFORMER_619M, 0)
    sett, setp, pick = ctx.lib.rs_debug_set_gemm_tile, ctx.lib.rs_debug_set_gemm_pairs, ctx.lib.rs_debug_gemm_tile_height
    setb = ctx.lib.rs_debug_set_gemm_breg
    for f in (sett, setp, setb):
        f.argtypes = [ctypes.c_int]
        f.restype = None
    pick.argtypes = [ctypes.c_int] * 5

    def setv(v):
        sett(v[0])
        setp(v[1])
        setb(v[2] if len(v) > 2 else 0)
    setg = ctx.lib.rs_debug_set_gemm_group_m
    setg.argtypes = [ctypes.c_int]
    setg.restype = None
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, m, n, k, flags in SHAPES:
        if only and not any(o in name for o in only):
            continue
        A = torch.randn((m, k), generator=g).to(torch.bfloat16).to(dev)
        W = (torch.randn((n, k), generator=g) / k ** 0.5).to(torch.bfloat16).to(dev)
        if pad_a:
            Ap = torch.zeros((m, k + pad_a), dtype=torch.bfloat16, device=dev)
            Ap[:, :k] = A
            A = Ap[:, :k]
        if pad_w:
            Wp = torch.zeros((n, k + pad_w), dtype=torch.bfloat16, device=dev)
            Wp[:, :k] = W
            W = Wp[:, :k]
        if any(v[2] for v in variants):
            ctx.set_tensor("bench." + name.split()[0], W)     # a registered weight: the fragment-major copy is made once, like the model's weights
        bias = torch.randn((n,), generator=g).to(dev)
        res = torch.randn((m, n), generator=g).to(dev) if flags & capi.GEMM_RESIDUAL else None
        glu = bool(flags & capi.GEMM_GLU)
        out = torch.empty((m, n // 2 if glu else n), dtype=torch.float32 if flags & capi.GEMM_OUT_F32 else torch.bfloat16, device=dev)
        if pad_c:
            out = torch.empty((m, out.shape[1] + pad_c), dtype=out.dtype, device=dev)[:, :out.shape[1]]
            if res is not None:
                rp = torch.empty((m, n + pad_c), dtype=res.dtype, device=dev)
                rp[:, :n] = res
                res = rp[:, :n]
        R = min(4096, m)
        ref = A[:R].float() @ W.float().t() + bias
        if glu:      # value / gate columns interleaved in blocks of 32
            r3 = ref.view(R, n // 64, 2, 32)
            ref = (r3[:, :, 0] * torch.sigmoid(r3[:, :, 1])).reshape(R, n // 2)
        if flags & capi.GEMM_SILU:
            ref = torch.nn.functional.silu(ref)
        if flags & capi.GEMM_RELU:
            ref = torch.relu(ref)
        if res is not None:
            ref = ref + res[:R]
        for v, gm in [(v, gm) for v in variants for gm in groups]:
            setv(v)
            if gm is not None:
                setg(gm)
            try:
                ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
                torch.cuda.synchronize()
            except capi.RsError as e:
                print(f"{name} {v}: {e}")
                continue
            err = (out[:R].float() - ref).abs().max().item()
            ts = []
            for _ in range(1 if quick else 7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(1 if quick else 4):
                    ctx.gemm(A, W, out, flags=flags, bias=bias, residual=res)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / (1 if quick else 4))
            ts.sort()
            us = ts[len(ts) // 2] * 1e3
            v = f"{v[0] or pick(m, n, k, 256, flags)}{'*' if not v[0] else ''} p{v[1]}{' breg' if v[2] else ''}"
            print(f"{name} M{m} N{n} K{k} tile {v}{'' if gm is None else f' gm{gm}'}{f' padA{pad_a}' if pad_a else ''}{f' padW{pad_w}' if pad_w else ''}{f' padC{pad_c}' if pad_c else ''}: err {err:.3g}  {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
        if "--yardstick" in sys.argv:
            # the vendor library on the same operands, same box, same process: torch.mm -> hipBLASLt / rocBLAS, plain bf16
            # product without any epilogue (so it does less work than the fused launches it stands next to)
            Wt = W.t()
            y = torch.mm(A, Wt)
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    torch.mm(A, Wt, out=y)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4)
            ts.sort()
            us = ts[len(ts) // 2] * 1e3
            print(f"{name} M{m} N{n} K{k} yardstick torch.mm (hipBLASLt, no epilogue): {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TF", flush=True)
            del y
        del A, W, out, res
    setv((0, 2, 0))


if __name__ == "__main__":
    main()
