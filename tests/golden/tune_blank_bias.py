"""Tune the synthetic joint's blank bias with the CPU oracle: prints the
number of emitted tokens per utterance for a few candidate biases.

    python scripts/tune_blank_bias.py [tiny|full] [n_utt]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reazonspeech_amd.runtime.config import TINY, FASTCONFORMER_619M   # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict      # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch             # noqa: E402
from oracle import model as om                                           # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cfg = TINY if which == "tiny" else FASTCONFORMER_619M
    t0 = time.time()
    sd = synthetic_state_dict(cfg, 0, blank_bias=0.0)
    print("weights", time.time() - t0)
    audio, lens = synthetic_batch(n_utt, 10.0, seed=1234)
    audio = torch.from_numpy(np.pad(audio, ((0, 0), (8000, 8000))))
    lens = torch.from_numpy(lens + 16000)
    t0 = time.time()
    f, el = om.forward_to_joint(cfg, sd, audio, lens, "fp32")
    print("encoder", time.time() - t0, "T'", el.tolist())
    base = sd["joint.joint_net.2.bias"][cfg.blank_id].item()
    for bb in [float(x) for x in (sys.argv[3:] or [0, 1, 2, 3, 4, 5, 6])]:
        sd["joint.joint_net.2.bias"][cfg.blank_id] = base + bb
        out = om.greedy_torch(cfg, sd, f, el)
        n = [len(i) for i, _ in out]
        distinct = [len(set(i)) for i, _ in out]
        print(f"blank_bias {bb:5.2f}: tokens {n} distinct {distinct}")


if __name__ == "__main__":
    main()
