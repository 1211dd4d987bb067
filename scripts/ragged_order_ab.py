"""SURVEY §8(d) ragged set (1024 utterances, U(2 s, 10 s), seed 1235) through transcribe_waveforms: batches in ascending
against descending length order (run on the GPU box)."""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.nemo.asr import load_model            # noqa: E402
from reazonspeech_amd.runtime.model import AsrModel         # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch  # noqa: E402

warnings.simplefilter("ignore")
model = load_model("cuda:0", synthetic=True)
audio, lens = synthetic_batch(1024, 10.0, seed=1235, ragged=True, min_seconds=2.0)
waves = [audio[i, :lens[i]] for i in range(1024)]
secs = float(lens.sum()) / 16000.0
model.transcribe_waveforms(waves)
ref = None
for rep in range(3):
    for first in (False, True):
        AsrModel.LONGEST_FIRST = first
        t0 = time.perf_counter()
        res = model.transcribe_waveforms(waves)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = res.ids
        assert res.ids == ref
        print(f"longest batch first = {first!s:5}: {dt * 1e3:7.1f} ms  {secs / dt:8.1f} RTFx", flush=True)
