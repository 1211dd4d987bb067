"""Generate tests/golden/bench_fp32.npz — the float32 oracle's END-TO-END outputs for EVERY row of the benchmark batch
(256 x 10 s, seed 1234) and of the ragged set (256 utterances, lengths U(2 s, 10 s), seed 1235), at the
619M geometry with the seeded synthetic weights of bench.py.  Run in the BUILD container (CPU, ~10 minutes on 8 cores):

    python tests/golden/make_bench_golden.py [--rows N]

The oracle processes ONE utterance per call with the reference's 0.5 s padding, exactly like the reference drives NeMo
(pkg/nemo-asr/src/transcribe.py:44-53: pad_audio, batch_size=1).  Stored per set:

  ids / frames / offsets   greedy token ids and emission frames of every row (oracle/rnnt_greedy.c on the oracle's own
                           joint projection), ragged -> flat + offsets
  enc_lens                 T'_b
  min_margin               the smallest top-1 minus top-2 joint logit margin along the row's own decision path (float64 walk,
                           oracle/audit.py) and `n_decisions`: a row whose min_margin is far above float32 reassociation
                           noise MUST come out identical from any float32 implementation; rows below NEAR_TIE are listed as
                           such by the generator, i.e. independently of any result they are compared with
  proj                     f[T'][J] @ R[J][8] per row, R = seeded N(0, 1) / sqrt(J): a 1.1-MB fingerprint of all 256 joint-
                           projection tensors (the tensors themselves are 90 MB)
  f_rows                   the joint projection itself for rows 0 and 1
  audio_sha256             checksum of the regenerated inputs

`-m gpu` consumers: tests/test_gpu_fullsize.py (float32 parity mode over all rows; flip audit of the throughput mode over all
rows) and bench.py's `parity` object.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime.config import FASTCONFORMER_619M          # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch              # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict       # noqa: E402
from oracle import model as om, greedy as og, audit                     # noqa: E402

PAD = 8000                       # pad_audio: int(0.5 * 16000) each side (pkg/nemo-asr/src/audio.py:80-82)
PROJ_SEED, PROJ_DIM = 20240926, 8
NEAR_TIE = 1e-3                  # margins below this are "near-ties" for a float32 implementation (noise ~1e-5)
SETS = {"equal": dict(seed=1234, ragged=False), "ragged": dict(seed=1235, ragged=True, min_seconds=2.0)}


def projection(J):
    g = torch.Generator().manual_seed(PROJ_SEED)
    return (torch.randn((J, PROJ_DIM), generator=g, dtype=torch.float32) / J ** 0.5).numpy()


def main():
    rows = 256
    if "--rows" in sys.argv:
        rows = int(sys.argv[sys.argv.index("--rows") + 1])
    cfg = FASTCONFORMER_619M
    sd = synthetic_state_dict(cfg, seed=0)
    R = projection(cfg.joint_hidden)
    store = {"rows": np.int64(rows), "proj_seed": np.int64(PROJ_SEED), "near_tie": np.float64(NEAR_TIE)}
    for name, kw in SETS.items():
        audio, lens = synthetic_batch(256, 10.0, **kw)
        store[name + "_audio_sha256"] = np.frombuffer(hashlib.sha256(audio.tobytes()).digest(), np.uint8)
        tp_max = cfg.enc_frames(cfg.mel_frames(audio.shape[1] + 2 * PAD))
        ids, frames, enc_lens, margins, ndec = [], [], [], [], []
        proj = np.zeros((rows, tp_max, PROJ_DIM), np.float32)
        f_rows = np.zeros((2, tp_max, cfg.joint_hidden), np.float32)
        t0 = time.time()
        for b in range(rows):
            wav = np.pad(audio[b, :int(lens[b])], PAD)
            f, el = om.forward_to_joint(cfg, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32")
            n = int(el[0])
            fb = f[0, :n].numpy()
            hyp = og.rnnt_greedy(cfg, sd, f.numpy(), el.numpy())[0]
            a = audit.flip_audit(cfg, sd, fb, fb, n, hyp[0], hyp[1])
            assert a["path_ok"] and not a["flips"], (name, b)
            ids.append(hyp[0]); frames.append(hyp[1]); enc_lens.append(n)
            margins.append(float(a["margins"].min())); ndec.append(int(a["decisions"]))
            proj[b, :n] = fb @ R
            if b < 2:
                f_rows[b, :n] = fb
            if b % 16 == 15:
                print(f"{name}: {b + 1}/{rows} rows, {time.time() - t0:.0f} s, tokens/row {np.mean([len(x) for x in ids]):.1f}, "
                      f"min margin so far {min(margins):.2e}", flush=True)
        off = np.zeros(rows + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in ids])
        store[name + "_offsets"] = off
        store[name + "_ids"] = np.asarray([k for x in ids for k in x], np.int32)
        store[name + "_frames"] = np.asarray([k for x in frames for k in x], np.int32)
        store[name + "_enc_lens"] = np.asarray(enc_lens, np.int32)
        store[name + "_min_margin"] = np.asarray(margins, np.float64)
        store[name + "_n_decisions"] = np.asarray(ndec, np.int32)
        store[name + "_proj"] = proj
        store[name + "_f_rows"] = f_rows
        near = [b for b in range(rows) if margins[b] < NEAR_TIE]
        print(f"{name}: done in {time.time() - t0:.0f} s; {sum(ndec)} decisions, rows with a margin below {NEAR_TIE:g}: {near}", flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_fp32.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
