"""Time the ESPnet Conformer-Transducer 120M path on the GPU box: B x 10 s utterances (+ the reference's (16000, 8000)
padding), inputs resident, front-end + encoder (+ CTC blank column) + transducer greedy search; per-class HIP-event times.

    python scripts/espnet_bench.py [--batch=256] [--steps=5]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime import capi                                    # noqa: E402
from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M            # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                   # noqa: E402
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet  # noqa: E402
from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list   # noqa: E402


def main():
    B = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--batch=")] or [256])[0]
    steps = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--steps=")] or [5])[0]
    cfg = ESPNET_CONFORMER_120M
    model = EspnetModel(cfg, synthetic_state_dict_espnet(cfg, 0), synthetic_token_list(cfg.vocab_size, 0), device="cuda:0")
    am = model.am
    audio, lens = synthetic_batch(B, 10.0, seed=1234)
    waves = [np.pad(audio[b, :lens[b]], (16000, 8000)) for b in range(B)]
    buf = am.stage(waves, buf=am.new_buffers(B, len(waves[0])))
    blank = torch.zeros((buf.B * buf.tp_max,), dtype=torch.float32, device=am.device)
    am.ctx.set_ctc_out(None, blank)
    am.run_device(buf)
    torch.cuda.synchronize()
    classes = {"gemm": capi.PROF_GEMM, "attention": capi.PROF_ATTN, "frontend": capi.PROF_FRONTEND, "decode": capi.PROF_DECODE,
               "elementwise": capi.PROF_ELEMENTWISE, "subsample": capi.PROF_SUBSAMPLE}
    am.ctx.profile_reset()
    am.ctx.profile_enable(sum(classes.values()))
    am.run_device(buf)
    torch.cuda.synchronize()
    prof = {k: am.ctx.profile_read(v) for k, v in classes.items()}
    shapes = {}
    for M, N, K, flags, flops, ms in am.ctx.profile_launches(capi.PROF_GEMM):
        a = shapes.setdefault((M, N, K), [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += flops
    am.ctx.profile_enable(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        am.run_device(buf)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    got = am.collect(buf)
    secs = float(lens.sum()) / 16000.0
    print(f"espnet 120M: B={B} x 10 s, T'={buf.tp_max}: {dt * 1e3:.1f} ms per batch (sequential schedule) = {secs / dt:.0f} x real-time; "
          f"{np.mean([len(x) for x in got.ids]):.1f} tokens / utterance")
    for k, v in prof.items():
        tf = v["flops"] / max(v["ms"], 1e-9) / 1e9
        print(f"  {k:12s} {v['ms']:8.2f} ms  {v['launches']:5d} launches  {tf:8.1f} TFLOP/s  {v['bytes'] / max(v['ms'], 1e-9) / 1e9:8.2f} TB/s(alg)")
    for (M, N, K), (n, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
        print(f"  gemm M{M} N{N} K{K}: {n} launches, {ms / n * 1e3:8.1f} us avg, {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
