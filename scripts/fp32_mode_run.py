"""Run the float32 parity mode over the benchmark batch (and the ragged set) on the GPU, time it, and write the ids /
frames / joint-projection fingerprint to gpurun_out/fp32_mode_<set>.npz (compared with tests/golden/bench_fp32.npz by
tests/test_gpu_fullsize.py; this script is the stand-alone form for the evidence files under profiles/)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M          # noqa: E402
from reazonspeech_amd.runtime.model import AsrModel                     # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch              # noqa: E402
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer       # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict       # noqa: E402


def main():
    cfg = FASTCONFORMER_619M
    sd = synthetic_state_dict(cfg, seed=0)
    t0 = time.time()
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0", precision="fp32")
    print(f"load {time.time() - t0:.1f} s", flush=True)
    g = torch.Generator().manual_seed(20240926)
    R = (torch.randn((cfg.joint_hidden, 8), generator=g, dtype=torch.float32) / cfg.joint_hidden ** 0.5).to(model.device)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name, kw in (("equal", dict(seed=1234)), ("ragged", dict(seed=1235, ragged=True, min_seconds=2.0))):
        audio, lens = synthetic_batch(256, 10.0, **kw)
        buf = model.stage([audio[b, :lens[b]] for b in range(256)], buf=model.new_buffers(256, 160000))
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            model.run_device(buf)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        got = model.collect(buf)
        proj = (buf.joint_enc @ R).cpu().numpy()
        off = np.zeros(257, np.int64)
        off[1:] = np.cumsum([len(x) for x in got.ids])
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"fp32_mode_{name}.npz"), offsets=off,
                            ids=np.asarray([k for x in got.ids for k in x], np.int32),
                            frames=np.asarray([k for x in got.frames for k in x], np.int32),
                            enc_lens=np.asarray(got.enc_lens, np.int32), proj=proj,
                            f_rows=buf.joint_enc[:2].cpu().numpy())
        print(f"{name}: {dt * 1e3:.0f} ms per batch of 256 in float32 mode ({float(lens.sum()) / 16000 / dt:.0f} x real-time), "
              f"{off[-1] / 256:.1f} tokens / utterance", flush=True)


if __name__ == "__main__":
    main()
