"""ESPnet Conv2dSubsampling: the second convolution's patches read in place by the GEMM (default) against the gathered patch matrix
($RS_SUB_IM2COL=1): same bits, time of the sequential schedule.   python scripts/espnet_conv_ab.py"""
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

cfg = ESPNET_CONFORMER_120M
em = EspnetModel(cfg, synthetic_state_dict_espnet(cfg, 0), synthetic_token_list(cfg.vocab_size, 0), device="cuda:0")
am = em.am
out = {}
for B, ragged in ((256, False), (37, True), (1, False)):
    audio, lens = synthetic_batch(B, 10.0, seed=77 + B, ragged=ragged, min_seconds=1.0)
    waves = [np.pad(audio[i, :lens[i]], PADDING) for i in range(B)]
    res = {}
    for mode in ("in_place", "im2col"):
        if mode == "im2col":
            os.environ["RS_SUB_IM2COL"] = "1"
        else:
            os.environ.pop("RS_SUB_IM2COL", None)
        buf = am.stage(waves, buf=am.new_buffers(B, max(len(w) for w in waves)))     # (the workspace plan depends on the mode)
        am.run_device(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            am.run_device(buf)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / n * 1e3, buf.joint_enc.clone(), am.collect(buf).ids)
    same = torch.equal(res["in_place"][1], res["im2col"][1]) and res["in_place"][2] == res["im2col"][2]
    print(f"B={B} ragged={ragged}: in place {res['in_place'][0]:.2f} ms, gathered {res['im2col'][0]:.2f} ms per batch (sequential); "
          f"joint projection and ids bit-identical: {same}")
    assert same
os.environ.pop("RS_SUB_IM2COL", None)
