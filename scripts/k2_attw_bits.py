"""Alternative kernel forms of the Zipformer path give the same BITS (the one-sweep / three-sweep attention-weights kernels:
$RS_K2_ATTW_SWEEPS=3; the fused / two-launch ConvNeXt pointwise pair: $RS_K2_CNX_FUSED=0): run once per form (the knobs are read once
per process) and compare the dumps.
    python scripts/k2_attw_bits.py dump gpurun_out/k2_a.pt ; RS_K2_ATTW_SWEEPS=3 python scripts/k2_attw_bits.py dump gpurun_out/k2_b.pt
    python scripts/k2_attw_bits.py cmp gpurun_out/k2_a.pt gpurun_out/k2_b.pt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "dump":
    from reazonspeech_amd.runtime.k2_config import ZIPFORMER_159M as cfg
    from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
    from reazonspeech_amd.runtime.synth import synthetic_batch
    from reazonspeech_amd.k2.asr.model import K2Model, synthetic_tokens
    am = K2Model(cfg, synthetic_state_dict_k2(cfg, 0), synthetic_tokens(cfg.vocab_size, 0), device="cuda:0").am
    audio, lens = synthetic_batch(24, 10.0, seed=77, ragged=True, min_seconds=1.0)
    waves = [np.pad(audio[i, :lens[i]], 14400) for i in range(24)]
    buf = am.stage(waves, buf=am.new_buffers(24, max(len(w) for w in waves)))
    am.run_device(buf)
    torch.cuda.synchronize()
    got = am.collect(buf)
    torch.save({"joint_enc": buf.joint_enc.cpu(), "ids": got.ids, "frames": got.frames, "form": os.environ.get("RS_K2_ATTW_SWEEPS", "1") + "/cnx" + os.environ.get("RS_K2_CNX_FUSED", "1")}, sys.argv[2])
    print("dumped", sys.argv[2], "form", os.environ.get("RS_K2_ATTW_SWEEPS", "1"), "tokens", sum(len(x) for x in got.ids))
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    same = torch.equal(a["joint_enc"], b["joint_enc"]) and a["ids"] == b["ids"] and a["frames"] == b["frames"]
    print(f"forms {a['form']} vs {b['form']}: joint projection, ids and frames bit-identical: {same}; max |diff| {float((a['joint_enc'] - b['joint_enc']).abs().max()):.3g}")
    sys.exit(0 if same else 1)
