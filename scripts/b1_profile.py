"""One utterance per call (the reference's pattern): wall time of transcribe_waveforms([wave]) next to the kernel time inside it.

    rocprofv3 --kernel-trace -d OUT -o trace -- python scripts/b1_profile.py [--decoding=alsd]
"""
import os
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.nemo.asr import load_model   # noqa: E402


def main():
    warnings.simplefilter("ignore")
    dec = ([a.split("=")[1] for a in sys.argv[1:] if a.startswith("--decoding=")] or [None])[0]
    model = load_model("cuda:0", decoding=dec, synthetic=True)
    rng = np.random.default_rng(0)
    secs = ([float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--seconds=")] or [10.0])[0]
    wave = (0.1 * rng.standard_normal(int(secs * 16000))).astype(np.float32)
    for _ in range(3):
        model.transcribe_waveforms([wave])
    lat = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = model.transcribe_waveforms([wave])
        lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"{secs:g} s decoding={dec or 'greedy'} latency median {lat[5] * 1e3:.2f} ms min {lat[0] * 1e3:.2f} ms tokens {len(r.ids[0])}")
    # stage / encoder / decode split (host clocks around synchronising calls)
    buf = model.stage([wave])
    torch.cuda.synchronize()
    t0 = time.perf_counter(); model.run_encoder(buf, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); t1 = time.perf_counter()
    model.decode(model.ctx, buf, buf.ws_dec, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"  encoder {1e3 * (t1 - t0):.2f} ms, decode {1e3 * (t2 - t1):.2f} ms")


if __name__ == "__main__":
    main()
