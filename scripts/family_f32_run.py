"""One float32-mode pass of a model family over its benchmark batch (for rocprofv3 --kernel-trace): python scripts/family_f32_run.py nemo|espnet|k2|avsr
($RS_F32_PRECISION=fp32x3 / $REAZONSPEECH_AVSR_PRODUCTS=x3: the three-term-product forms)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from reazonspeech_amd.runtime.synth import synthetic_batch          # noqa: E402


PREC = os.environ.get("RS_F32_PRECISION", "fp32")          # "fp32x3": the three-term-product form of the same mode


def main(which):
    if which == "avsr":
        from reazonspeech_amd.runtime.avsr_config import AVSR_BASE
        from reazonspeech_amd.runtime.avsr_synth import synthetic_clips
        from reazonspeech_amd.avsr import synthetic_model
        model = synthetic_model(AVSR_BASE, 0, "cuda:0")
        a, v, m, _ = synthetic_clips(16, 250, seed=4242)
        ad, vd, md = (torch.from_numpy(x).cuda() for x in (a, v[:, :, 0], m))
        for rep in range(2):
            t0 = time.perf_counter()
            model.dev.encode(ad, vd, md)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model.generate(input_values=ad, pixel_values=vd, padding_mask=md, num_beams=5, max_new_tokens=16)
            torch.cuda.synchronize()
            print(f"avsr: encoder {1e3 * (t1 - t0):.1f} ms, beam-5 x 16 tokens {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
        return
    if which == "k2":
        from reazonspeech_amd.runtime.k2_config import ZIPFORMER_159M as cfg
        from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
        from reazonspeech_amd.k2.asr.model import K2Model, synthetic_tokens
        am = K2Model(cfg, synthetic_state_dict_k2(cfg, 0), synthetic_tokens(cfg.vocab_size, 0), device="cuda:0", precision=PREC).am
        pad = (int(0.9 * 16000),) * 2
    elif which == "espnet":
        from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M as cfg
        from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet
        from reazonspeech_amd.espnet.asr.model import EspnetModel, synthetic_token_list, PADDING
        am = EspnetModel(cfg, synthetic_state_dict_espnet(cfg, 0), synthetic_token_list(cfg.vocab_size, 0), device="cuda:0", precision=PREC).am
        pad = PADDING
    else:
        from reazonspeech_amd.runtime.config import FASTCONFORMER_619M as cfg
        from reazonspeech_amd.runtime.model import AsrModel
        from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
        from reazonspeech_amd.runtime.weights import synthetic_state_dict
        am = AsrModel(cfg, synthetic_state_dict(cfg, 0), SyntheticTokenizer(cfg.vocab_size), device="cuda:0", precision=PREC)
        pad = (0, 0)
    audio, lens = synthetic_batch(256, 10.0, seed=4242 if which != "nemo" else 1234)
    waves = [np.pad(audio[b, :lens[b]], pad) for b in range(256)]
    buf = am.stage(waves, buf=am.new_buffers(256, len(waves[0])))
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        am.run_device(buf)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{which}: {dt * 1e3:.0f} ms per batch of 256 in float32 mode ({float(lens.sum()) / 16000 / dt:.0f} x real-time)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "nemo")
