"""One utterance per call with a different length every call (the reference's pattern on a folder of files): per-call latency
when the buffer geometry is new against when it is cached (run on the GPU box)."""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.nemo.asr import load_model   # noqa: E402

warnings.simplefilter("ignore")
model = load_model("cuda:0", synthetic=True)
rng = np.random.default_rng(0)
secs = [5.0, 12.3, 3.1, 25.0, 8.8, 17.5, 2.2, 29.0, 6.4, 21.0, 4.0, 14.2]
waves = [(0.1 * rng.standard_normal(int(s * 16000))).astype(np.float32) for s in secs]
model.transcribe_waveforms([waves[0]])
for rnd in range(3):
    line = []
    for s, w in zip(secs, waves):
        t0 = time.perf_counter()
        model.transcribe_waveforms([w])
        line.append(f"{s:g}s:{(time.perf_counter() - t0) * 1e3:.1f}")
    print(f"round {rnd}: " + "  ".join(line), flush=True)
