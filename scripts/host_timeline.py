"""Timeline of the host-to-host boundary (run on the GPU box): where the wall time of `transcribe_batch` on 2048 utterances
goes — stager fills, encoder enqueues, decode lanes, harvest (D2H + unpack), post-processing (ids -> text).

    python scripts/host_timeline.py [--batches=8] [--reps=3]
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.nemo.asr import load_model, transcribe_batch, audio_from_numpy, TranscribeConfig   # noqa: E402


def main():
    nb = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--batches=")] or [8])[0]
    reps = ([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--reps=")] or [3])[0]
    import warnings
    warnings.simplefilter("ignore")
    model = load_model("cuda:0", synthetic=True)
    rng = np.random.default_rng(0)
    if "--ragged" in sys.argv:          # SURVEY 8(d)'s ragged set: lengths U(2 s, 10 s), seed 1235
        from reazonspeech_amd.runtime.synth import synthetic_batch
        audio, lens = synthetic_batch(nb * 256, 10.0, seed=1235, ragged=True, min_seconds=2.0)
        waves = [audio[i, :lens[i]] for i in range(nb * 256)]
    else:
        base = [(0.1 * rng.standard_normal(160000)).astype(np.float32) for _ in range(256)]
        waves = [base[i % 256] for i in range(nb * 256)]
    audios = [audio_from_numpy(w, 16000) for w in waves]
    model.transcribe_waveforms(waves[:1024])
    events = []
    t0 = [0.0]

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def inner(*a, **k):
            s = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                events.append((label, threading.current_thread().name, s - t0[0], time.perf_counter() - t0[0]))
        setattr(obj, name, inner)
    wrap(model, "fill_host", "fill")
    wrap(model, "run_encoder", "enc_enqueue")
    wrap(model, "decode", "decode")
    wrap(model, "collect", "collect")
    post = []
    orig_tw = model.transcribe_waveforms

    def tw(waveforms, max_batch=256, on_batch=None):
        def ob(indices, decoded):
            s = time.perf_counter()
            on_batch(indices, decoded)
            post.append((s - t0[0], time.perf_counter() - t0[0]))
        return orig_tw(waveforms, max_batch=max_batch, on_batch=ob if on_batch else None)
    model.transcribe_waveforms = tw
    for rep in range(reps):
        del events[:], post[:]
        t0[0] = time.perf_counter()
        res = transcribe_batch(model, audios, TranscribeConfig(verbose=False))
        wall = time.perf_counter() - t0[0]
        print(f"== rep {rep}: transcribe_batch {wall * 1e3:.1f} ms, {len(res)} results")
        for label in ("fill", "enc_enqueue", "decode", "collect"):
            ev = [e for e in events if e[0] == label]
            print(f"  {label:12s}" + " ".join(f"[{s * 1e3:6.1f}-{e * 1e3:6.1f}]" for _, _, s, e in ev))
        print(f"  {'post':12s}" + " ".join(f"[{s * 1e3:6.1f}-{e * 1e3:6.1f}]" for s, e in post))
    for rep in range(reps):
        t1 = time.perf_counter()
        orig_tw(waves)
        print(f"== transcribe_waveforms (ids only) {(time.perf_counter() - t1) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
