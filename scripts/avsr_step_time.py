"""Where an AV-HuBERT generate() call spends its time on the device: encoder, one decoding step (16 and 80 hypothesis rows, with /
without re-parenting), steps back to back without a host sync, the step + log-softmax + top-k + D2H round trip, whole generate()
calls (cold, then warm).     python scripts/avsr_step_time.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from reazonspeech_amd.avsr import synthetic_model
from reazonspeech_amd.runtime.avsr_config import AVSR_BASE
from reazonspeech_amd.runtime.avsr_synth import synthetic_clips
m = synthetic_model(AVSR_BASE, 0, device="cuda:0")
B, T, K = 16, 250, 5
a, v, mask, lens = synthetic_clips(B, T, seed=1, ragged=False)
t0 = time.perf_counter(); enc = m.avhubert(input_values=a, pixel_values=v, padding_mask=mask).last_hidden_state; torch.cuda.synchronize(); print("encode (first)", time.perf_counter() - t0)
t0 = time.perf_counter(); enc = m.avhubert(input_values=a, pixel_values=v, padding_mask=mask).last_hidden_state; torch.cuda.synchronize(); print("encode", time.perf_counter() - t0)
dev = m.dev
for rows_k in (1, 5):
    dec = dev.decoding(enc, mask, rows_k, 64)
    rows = B * rows_k
    tok = np.zeros((rows,), np.int64); src = np.arange(rows)
    torch.cuda.synchronize()
    for label, use_src in (("no reparent", None), ("reparent", src)):
        ts = []
        for step in range(24):
            t0 = time.perf_counter()
            lg = dec.step(tok, step, use_src)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"beams {rows_k} rows {rows} {label}: step wall ms (steps 0..23):", " ".join(f"{t*1e3:.2f}" for t in ts[::3]))
    # device time of one step by events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for step in range(24, 40):
        dec.step(tok, step, src)
    e1.record(); torch.cuda.synchronize()
    print(f"   16 steps back to back (no host sync between): {e0.elapsed_time(e1)/16:.3f} ms per step")
    t0 = time.perf_counter()
    lg = dec.step(tok, 40, src)
    logp = torch.log_softmax(lg.float(), dim=-1).view(B, rows_k, -1)
    top = torch.topk(logp.view(B, -1), k=2 * rows_k)
    x = top[0].cpu().numpy(), top[1].cpu().numpy()
    print("   step + log_softmax + topk + D2H:", (time.perf_counter() - t0) * 1e3, "ms")
for nb, nt in ((1, 32), (5, 32), (1, 32), (5, 32)):
    t0 = time.perf_counter(); out = m.generate(input_values=a, pixel_values=v, padding_mask=mask, num_beams=nb, max_new_tokens=nt); torch.cuda.synchronize()
    print(f"generate beams {nb} tokens {nt}: {(time.perf_counter()-t0)*1e3:.1f} ms total (incl. encode), out len {out.shape[1]}")
