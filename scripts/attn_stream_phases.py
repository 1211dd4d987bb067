"""Where the streaming attention kernel's time goes: the same launch with phases left out ($RS_ATTN_STREAM = 1 + 16 x mask; the
results are wrong then, only the time means something).  Mask bits: 1 PV (ds_read_b64_tr_b16 + MFMA), 2 BD^T MFMAs, 4 the skew
through LDS, 8 the K / V DMAs, 16 S^T (K fragment reads + MFMA), 32 the position loads, 64 the per-block workgroup barrier,
128 the whole key loop (prologue + ctx store only).       python scripts/attn_stream_phases.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M, FASTCONFORMER_619M
dev = torch.device("cuda", 0)
def run(cfg, B, T, label):
    d = cfg.d_model
    ctx = capi.Context(cfg, 0)
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn((B * T, 3 * d), generator=g).to(torch.bfloat16).to(dev)
    pos = torch.randn((2 * T - 1, d), generator=g).to(torch.bfloat16).to(dev)
    bu = (0.3 * torch.randn(d, generator=g)).to(dev); bv = (0.3 * torch.randn(d, generator=g)).to(dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    out = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)
    res = []
    for dbg in (0, 31, 31|32, 31|32|64, 128, 32|2, 8|16|1, 64):
        ctx.lib.rs_debug_set_attn_stream(1 + 16 * dbg)
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): ctx.attention(qkv, pos, bu, bv, lens, B, T, out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 8 * 1e3)
        res.append(f"dbg={dbg:2d}: {sorted(ts)[2]:6.1f}")
    print(label, " | ".join(res), flush=True)
    ctx.close()
run(FASTCONFORMER_619M, 256, 138, "hd128 T138")
run(FASTCONFORMER_619M, 8, 1500, "hd128 T1500 B8")
run(ESPNET_CONFORMER_120M, 256, 358, "hd64 T358")
