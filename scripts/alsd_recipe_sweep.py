"""Label density of the device ALSD (beam 4) and greedy searches under synthetic checkpoints whose prediction network weighs
more in the joint (dec_gain) — the search for a synthetic recipe on which ALSD behaves like on a trained model (~50 labels per
10 s utterance instead of spending its whole label budget).  256 x 10 s, decode only, joint-encoder projection resident.

    python scripts/alsd_recipe_sweep.py [--batch=256] [--gains=1,4,8] [--bias=6.2,8,10] [--out-gain=8]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M                 # noqa: E402
from reazonspeech_amd.runtime.model import AsrModel                           # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                    # noqa: E402
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer              # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict             # noqa: E402


def arg(name, default):
    v = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith(f"--{name}=")]
    return v[0] if v else default


def main():
    B = int(arg("batch", 256))
    gains = [float(x) for x in arg("gains", "1,8").split(",")]
    biases = [float(x) for x in arg("bias", "6.2,9").split(",")]
    out_gain = float(arg("out-gain", "1"))
    audio, lens = synthetic_batch(B, 10.0, seed=1234)
    waves = [audio[b, :lens[b]] for b in range(B)]
    secs = float(lens.sum()) / 16000.0
    for g in gains:
        for bias in biases:
            cfg = FASTCONFORMER_619M.with_(decoding="alsd", beam_size=4)
            sd = synthetic_state_dict(cfg, 0, blank_bias=bias, dec_gain=g, out_gain=out_gain)
            m = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
            buf = m.stage(waves, buf=m.new_buffers(B, int(10.0 * 16000)))
            m.run_device(buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.run_device(buf)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            alsd = m.collect(buf)
            stream = torch.cuda.current_stream().cuda_stream
            ids = torch.zeros((B, buf.tp_max * cfg.max_symbols), dtype=torch.int32, device=m.device)
            frames = torch.zeros_like(ids)
            n_ids = torch.zeros((B,), dtype=torch.int32, device=m.device)
            m.ctx.rnnt_greedy(buf.joint_enc, buf.enc_lens, B, buf.tp_max, ids.shape[1], ids, frames, n_ids, buf.ws, stream)
            torch.cuda.synchronize()
            ng = n_ids.cpu().numpy()
            na = np.asarray([len(x) for x in alsd.ids])
            print(f"dec_gain {g:g} out_gain {out_gain:g} blank_bias {bias:g}: ALSD-4 labels / utterance mean {na.mean():.1f} max {na.max()}, greedy mean {ng.mean():.1f} max {ng.max()}; "
                  f"front-end + encoder + ALSD sequential {dt * 1e3:.1f} ms per batch = {secs / dt:.0f} x real-time", flush=True)
            del m, buf
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
