"""Phase timeline of the rel-pos attention kernel (debug build with per-phase drains + cycle stamps).

    python scripts/attn_trace.py [B] [T]
Wave 1 of every workgroup stamps: start, K/V staged, barrier, Q fragments, first position block, then per key
block (S^T, position rows + BD^T, skew + scores, softmax update), end of PV, ctx stored.  Draining the counters
at every stamp serialises what little the wave overlaps, so the SUM is an upper bound; the shares are the point.
"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M as cfg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 138
dev = torch.device("cuda", 0)
ctx = capi.Context(cfg, 0)
lib = ctx.lib
lib.rs_debug_set_attn_trace.argtypes = [ctypes.c_void_p]
d, H = cfg.d_model, cfg.n_heads
g = torch.Generator().manual_seed(0)
qkv = torch.randn((B * T, 3 * d), generator=g).to(torch.bfloat16).to(dev)
pos = torch.randn((2 * T - 1, d), generator=g).to(torch.bfloat16).to(dev)
bu = (0.1 * torch.randn(d, generator=g)).to(dev)
bv = (0.1 * torch.randn(d, generator=g)).to(dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
out = torch.empty((B * T, d), dtype=torch.bfloat16, device=dev)


def timed(n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.attention(qkv, pos, bu, bv, lens, B, T, out); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


print(f"B={B} T={T}: production kernel {timed():.1f} us")
nwg = B * H * ((((T + 31) // 32) + 4) // 5 if (T + 31) // 32 > 6 else 1)
tr = torch.full((B * H * 8, 40), -1, dtype=torch.int64, device=dev)
lib.rs_debug_set_attn_trace(ctypes.c_void_p(tr.data_ptr()))
print(f"traced kernel {timed(3):.1f} us")
lib.rs_debug_set_attn_trace(None)
t = tr.cpu().numpy().astype(np.float64)
t = t[t[:, 1] >= 0]
n = int((t[0] >= 0).sum())
names = ["start", "Q/K/V loads, K/V staged", "barrier", "Q fragments", "first position block"]
k = 0
while len(names) < n - 2:
    names += [f"blk{k} S^T", f"blk{k} pos rows + BD^T", f"blk{k} skew + scores", f"blk{k} softmax"]
    k += 1
names = names[:n - 2] + ["last PV", "ctx stored"]
clk = 2.4e3   # cycles per us (nominal)
print(f"{len(t)} workgroups traced, {n} stamps; phase = time since the previous stamp (us @2.4 GHz)")
agg = {}
for i in range(1, n):
    dt = (t[:, i] - t[:, i - 1]) / clk
    print(f"   {names[i]:28s} mean {dt.mean():7.2f}  p10 {np.percentile(dt, 10):7.2f}  p90 {np.percentile(dt, 90):7.2f}")
    key = names[i].split(" ", 1)[1] if names[i].startswith("blk") else names[i]
    agg[key] = agg.get(key, 0.0) + dt.mean()
tot = (t[:, n - 1] - t[:, 0]) / clk
print(f"   workgroup total              mean {tot.mean():7.2f} us")
print("   by phase:", ", ".join(f"{k} {v:.2f}" for k, v in agg.items()))
