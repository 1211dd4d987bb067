"""Attention with two workgroups per CU (small key chunks, fewer waves per workgroup: $RS_ATTN64=1) against the default (one
workgroup per CU): same bits, time of the sequential schedule, both families.   python scripts/attn_2wg_ab.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M  # noqa: E402
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet  # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch  # noqa: E402
from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list, PADDING  # noqa: E402


def ab(am, waves, label, modes=("0", "4,4")):
    buf = am.stage(waves, buf=am.new_buffers(len(waves), max(len(w) for w in waves)))
    res = {}
    for rep in range(2):
        for mode in modes:
            os.environ["RS_ATTN64"] = mode
            am.run_device(buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                am.run_device(buf)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            res.setdefault(mode, []).append((ms, buf.joint_enc.clone(), am.collect(buf).ids))
    same = all(torch.equal(res[modes[0]][0][1], res[m][0][1]) and res[modes[0]][0][2] == res[m][0][2] for m in modes)
    times = "; ".join(f"[{m}] {res[m][0][0]:.2f} / {res[m][1][0]:.2f}" for m in modes)
    print(f"{label}: ms per batch (sequential, 2 repetitions) by $RS_ATTN64 = key blocks per chunk, waves per workgroup (0 = one workgroup "
          f"per CU: 6 blocks, 6 waves): {times}; joint projection and ids bit-identical across all: {same}", flush=True)
    os.environ.pop("RS_ATTN64", None)
    return same


ok = True
cfg = ESPNET_CONFORMER_120M
em = EspnetModel(cfg, synthetic_state_dict_espnet(cfg, 0), synthetic_token_list(cfg.vocab_size, 0), device="cuda:0")
audio, lens = synthetic_batch(256, 10.0, seed=5)
ok &= ab(em.am, [np.pad(audio[i, :lens[i]], PADDING) for i in range(256)], "espnet 120M, head_dim 64, T' = 358, B = 256",
         modes=("0", "4,4", "3,4", "4,3", "2,3", "2,2", "4,6"))
audio, lens = synthetic_batch(19, 6.0, seed=6, ragged=True, min_seconds=0.6)
ok &= ab(em.am, [np.pad(audio[i, :lens[i]], PADDING) for i in range(19)], "espnet 120M, ragged 19 x <= 6 s")
assert ok
