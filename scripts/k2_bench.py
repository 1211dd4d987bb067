"""One configuration of bench.py on its own: reazonspeech.k2.asr (Zipformer2 159M), 256 x 10 s per step.
    python scripts/k2_bench.py [steps] [--seq]      (--seq: sequential schedule, for rocprofv3 --kernel-trace --stats)"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
    args = types.SimpleNamespace(batch=256, seconds=10.0, dec_streams=2)
    if "--seq" in sys.argv:
        from reazonspeech_amd.runtime.k2_config import ZIPFORMER_159M
        from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
        from reazonspeech_amd.runtime.synth import synthetic_batch
        from reazonspeech_amd.k2.asr.model import K2Model, synthetic_tokens
        import time
        cfg = ZIPFORMER_159M
        km = K2Model(cfg, synthetic_state_dict_k2(cfg, 0), synthetic_tokens(cfg.vocab_size, 0), device="cuda:0")
        audio, lens = synthetic_batch(256, 10.0, seed=4242)
        waves = [np.pad(audio[i, :lens[i]], 14400) for i in range(256)]
        buf = km.am.stage(waves, buf=km.am.new_buffers(256, len(waves[0])))
        km.am.run_device(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            km.am.run_device(buf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print(json.dumps({"sequential_ms_per_batch": round(dt * 1e3, 2), "rtfx": round(2560.0 / dt, 1), "tokens": float(buf.n_ids.float().mean())}))
    else:
        print(json.dumps(bench.k2_config(torch.device("cuda:0"), args, steps=steps)))
