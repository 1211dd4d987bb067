"""Small batches (BASELINE configs[1]: 32 x 10 s) through the pipeline with ONE and with TWO encoder lanes, same box, interleaved.

A B = 32 encoder launch is 92 - 184 workgroups on 256 CUs (N = 1024 GEMMs: 35 x 4 tiles of 128 rows = 55 % of the CUs in one
round; ffn_up: 23 pair workgroups per XCD on 32 CUs), so two batches' encoders on two HIP streams can fill the holes of each other —
what lost at B = 256 (profiles/r04f_ab_RS_ENC_STREAMS.txt: full launches evict each other's operands) may win here.

    python scripts/small_batch_enc_lanes_ab.py [batch ...]          (default 32 64 128)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M          # noqa: E402
from reazonspeech_amd.runtime.model import AsrModel                     # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch              # noqa: E402
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer       # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict       # noqa: E402

cfg = FASTCONFORMER_619M
model = AsrModel(cfg, synthetic_state_dict(cfg, seed=0), SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
DEC = int(os.environ.get("RS_DEC_STREAMS", "2"))
for B in [int(a) for a in sys.argv[1:]] or [32, 64, 128]:
    n_sets = DEC + 3
    bufs, secs = [], []
    for k in range(n_sets):
        audio, lens = synthetic_batch(B, 10.0, seed=4321 + 1000 * k)
        bufs.append(model.stage([audio[i, :lens[i]] for i in range(B)], buf=model.new_buffers(B, 160000)))
        secs.append(float(lens.sum()) / 16000.0)
    torch.cuda.synchronize()
    steps = max(20, 2048 // B)
    ids = {}
    line = []
    for rep in range(2):
        for lanes in (1, 2, 3):
            model.run_pipelined(bufs, 5, dec_streams=DEC, enc_streams=lanes)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.run_pipelined(bufs, steps, dec_streams=DEC, enc_streams=lanes)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ids.setdefault(lanes, [model.collect(b).ids for b in bufs])
            line.append(f"[{lanes} lane(s)] {dt / steps * 1e3:7.3f} ms/step {sum(secs[i % n_sets] for i in range(steps)) / dt:8.0f} RTFx")
    same = all(ids[k] == ids[1] for k in ids)
    print(f"B = {B:3d} x 10 s, {DEC} decode lanes, {n_sets} resident sets, {steps} steps: " + " | ".join(line) + f" | ids identical across lane counts: {same}",
          flush=True)
    assert same
    del bufs
