"""Time the default transducer beam search (k_rnnt_beam.hip) on the ESPnet Conformer-Transducer 120M shape: B x 10 s utterances
(+ the reference's (16000, 8000) padding), joint-encoder projection resident, decode only — next to the greedy search on the same
projection.  The synthetic checkpoint uses dec_gain = 8 (the search needs a prediction network that matters: see
tests/test_oracle_espnet_beam.py); --bias sweeps its blank offset, i.e. the label density.

    python scripts/espnet_beam_bench.py [--batch=256] [--beam=20] [--bias=15,16,17] [--max-pops=640]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M            # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                   # noqa: E402
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet  # noqa: E402
from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list   # noqa: E402


def arg(name, default):
    v = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith(f"--{name}=")]
    return v[0] if v else default


def main():
    B = int(arg("batch", 256))
    beams = [int(x) for x in arg("beam", "20").split(",")]
    biases = [float(x) for x in arg("bias", "16").split(",")]
    max_pops = int(arg("max-pops", 640))
    cfg = ESPNET_CONFORMER_120M
    audio, lens = synthetic_batch(B, 10.0, seed=1234)
    waves = [np.pad(audio[b, :lens[b]], (16000, 8000)) for b in range(B)]
    secs = float(sum(len(w) for w in waves)) / 16000.0
    for bias in biases:
        model = EspnetModel(cfg, synthetic_state_dict_espnet(cfg, 0, blank_bias=bias, dec_gain=8.0), synthetic_token_list(cfg.vocab_size, 0),
                            device="cuda:0")
        am = model.am
        buf = am.stage(waves, buf=am.new_buffers(B, len(waves[0])))
        t0 = time.perf_counter()
        am.run_device(buf)
        torch.cuda.synchronize()
        greedy = am.collect(buf)
        stream = torch.cuda.current_stream().cuda_stream
        t0 = time.perf_counter()
        for _ in range(3):
            am.ctx.rnnt_greedy(buf.joint_enc, buf.enc_lens, B, buf.tp_max, buf.u_max, buf.ids, buf.frames, buf.n_ids, buf.ws, stream)
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 3
        frames = float(buf.enc_lens.sum())
        print(f"bias {bias}: B={B}, T'={buf.tp_max}, greedy {np.mean([len(x) for x in greedy.ids]):.1f} labels / utterance, greedy search {tg * 1e3:.1f} ms")
        for beam in beams:
            dev = am.device
            cap = 2 * buf.tp_max + 16
            ids = torch.zeros((B, cap), dtype=torch.int32, device=dev)
            n_ids = torch.zeros((B,), dtype=torch.int32, device=dev)
            scores = torch.zeros((B,), dtype=torch.float32, device=dev)
            pops = torch.zeros((B,), dtype=torch.int32, device=dev)
            nbytes = am.ctx.beam_workspace_bytes(B, beam, buf.tp_max, max_pops)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            try:
                am.ctx.rnnt_beam(buf.joint_enc, buf.enc_lens, B, buf.tp_max, beam, True, max_pops, ids, n_ids, scores, pops, ws, stream)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                am.ctx.rnnt_beam(buf.joint_enc, buf.enc_lens, B, buf.tp_max, beam, True, max_pops, ids, n_ids, scores, pops, ws, stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            except RuntimeError as e:
                print(f"  beam {beam}: {e}")
                continue
            p = pops.cpu().numpy().astype(np.float64)
            n = n_ids.cpu().numpy()
            same = sum(ids[b, :n[b]].cpu().tolist() == greedy.ids[b] for b in range(B))
            print(f"  beam {beam}: {dt * 1e3:.1f} ms per batch = {secs / dt:.0f} x real-time (decode only); workspace {nbytes / 2**20:.0f} MiB; "
                  f"pops / frame mean {p.sum() / frames:.1f}, per-utterance total min {p.min():.0f} mean {p.mean():.0f} max {p.max():.0f} "
                  f"-> {dt * 1e6 / p.max():.1f} us per iteration; labels / utterance {n.mean():.1f}; same as greedy {same}/{B}")
        del model, am, buf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
