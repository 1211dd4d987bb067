"""Generate tests/golden/bench_k2_fp32.npz — the float32 oracle's END-TO-END outputs for EVERY row of the Zipformer benchmark
batch (bench.py `configs.k2_zipformer_159m`: 256 x 10 s, seed 4242, each padded with the reference's 0.9 s on both sides —
pkg/k2-asr/src/transcribe.py:7,31-33), at the 159M Zipformer2 geometry with the seeded synthetic weights of bench.py.  Run in the
BUILD container (CPU, about 20 minutes):

    python tests/golden/make_k2_golden.py [--rows N]

The oracle (oracle/zipformer.py, float32, its OWN window / mel banks / relative-position rows) processes ONE utterance per call,
like the reference drives sherpa-onnx (create_stream / accept_waveform / decode_stream: transcribe.py:36-39).  Stored:

  ids / frames / ids_offsets   sherpa-onnx's offline greedy search (one symbol per frame, blank and <unk> skipped) of every row —
                               oracle/k2_greedy.c on the oracle's own joint projection, which equals the torch restatement
                               (`greedy_search`: asserted here on the first rows)
  enc_lens                     T'_b
  min_margin / n_decisions     the smallest top-1 minus top-2 joiner-logit margin along the row's own decision path (float64 walk):
                               a row whose margin is far above float32 reassociation noise MUST come out identical from any float32
                               implementation
  proj                         f[T'][J] @ R[J][8] per row (R seeded N(0, 1) / sqrt(J)): fingerprint of every joint projection
  f_rows                       the joint projection itself for rows 0 and 1
  feat_proj                    feats[T][80] @ Rf[80][4] per row: fingerprint of the fbank features (the oracle's own tables)
  audio_sha256                 checksum of the regenerated inputs

Consumers: tests/test_gpu_k2_fp32.py (`-m gpu`: the float32 parity mode over all 256 rows, the bf16 mode's flip audit),
tests/test_k2_host.py (CPU: the golden's own consistency on a few rows), bench.py `configs.k2_zipformer_159m.parity`.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime.k2_config import ZIPFORMER_159M                         # noqa: E402
from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2               # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                           # noqa: E402
from oracle import zipformer as oz, greedy as og                                     # noqa: E402

PAD = int(0.9 * 16000)
PROJ_SEED, PROJ_DIM, FEAT_DIM = 20240930, 8, 4
NEAR_TIE = 1e-3
SEED, SECONDS = 4242, 10.0


def projection(J, dim=PROJ_DIM, seed=PROJ_SEED):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((J, dim), generator=g, dtype=torch.float32) / J ** 0.5).numpy()


def greedy_margins(cfg, sd, f, ids, frames):
    """float64 walk along the greedy path (one decision per frame; blank and <unk> keep the context): the smallest margin
    between the two best joiner logits"""
    emb = sd["decoder.embedding.weight"].double()
    cw = sd["decoder.conv.weight"].double()                     # [D][4][ctx], groups = D / 4
    wp, bp = sd["joiner.decoder_proj.weight"].double(), sd["joiner.decoder_proj.bias"].double()
    wo, bo = sd["joiner.output_linear.weight"].double(), sd["joiner.output_linear.bias"].double()
    D = emb.shape[1]

    def dec(ctx):
        e = torch.stack([emb[t] if t >= 0 else torch.zeros(D, dtype=torch.float64) for t in ctx], dim=1)      # [D][ctx]
        h = (cw * e.reshape(D // 4, 1, 4, len(ctx)).expand(D // 4, 4, 4, len(ctx)).reshape(D, 4, len(ctx))).sum(dim=(1, 2))
        return wp @ torch.relu(h) + bp
    hist = [-1] * (cfg.context_size - 1) + [cfg.blank_id]
    g = dec(hist[-cfg.context_size:])
    fd = torch.from_numpy(f).double()
    emitted = dict(zip(frames, ids))
    worst = float("inf")
    for t in range(fd.shape[0]):
        z = wo @ torch.tanh(fd[t] + g) + bo
        top = torch.topk(z, 2)
        worst = min(worst, float(top.values[0] - top.values[1]))
        k = int(top.indices[0])
        want = emitted.get(t, None)
        took = k if k not in (cfg.blank_id, cfg.unk_id) else None
        if took != want:                   # the float64 walk itself disagrees: a tie at float32 resolution
            worst = 0.0
        if want is not None:
            hist.append(want)
            g = dec(hist[-cfg.context_size:])
    return worst


def main():
    rows = 256
    if "--rows" in sys.argv:
        rows = int(sys.argv[sys.argv.index("--rows") + 1])
    cfg = ZIPFORMER_159M
    sd = synthetic_state_dict_k2(cfg, 0)
    R, Rf = projection(cfg.joiner_dim), projection(cfg.n_mels, FEAT_DIM, PROJ_SEED + 1)
    audio, lens = synthetic_batch(256, SECONDS, seed=SEED)
    store = {"rows": np.int64(rows), "proj_seed": np.int64(PROJ_SEED), "near_tie": np.float64(NEAR_TIE), "seed": np.int64(SEED),
             "audio_sha256": np.frombuffer(hashlib.sha256(audio.tobytes()).digest(), np.uint8)}
    l_pad = audio.shape[1] + 2 * PAD
    t_max = cfg.fbank_frames(l_pad)
    tp_max = cfg.enc_frames(t_max)
    ids, frames, enc_lens, margins = [], [], [], []
    proj = np.zeros((rows, tp_max, PROJ_DIM), np.float32)
    feat_proj = np.zeros((rows, t_max, FEAT_DIM), np.float32)
    f_rows = np.zeros((2, tp_max, cfg.joiner_dim), np.float32)
    t0 = time.time()
    for b in range(rows):
        wav = np.pad(audio[b, :int(lens[b])], PAD)
        out = oz.forward(cfg, sd, wav, "fp32")
        f = out["joint_enc"].numpy()
        n = f.shape[0]
        hyp = og.k2_greedy(cfg, sd, f[None], np.asarray([n], np.int32))[0]
        if b < 4:
            ref = oz.greedy_search(cfg, sd, out["joint_enc"])
            assert (hyp[0], hyp[1]) == (ref[0], ref[1]), f"row {b}: k2_greedy.c != greedy_search"
        ids.append(hyp[0]); frames.append(hyp[1]); enc_lens.append(n)
        margins.append(greedy_margins(cfg, sd, f, hyp[0], hyp[1]))
        proj[b, :n] = f @ R
        feat_proj[b, :out["feats"].shape[0]] = out["feats"].numpy() @ Rf
        if b < 2:
            f_rows[b, :n] = f
        if b % 8 == 7 or b == rows - 1:
            print(f"{b + 1}/{rows} rows, {time.time() - t0:.0f} s, tokens/row {np.mean([len(x) for x in ids]):.1f}, min margin {min(margins):.2e}", flush=True)
    off = np.zeros(rows + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in ids])
    store["ids_offsets"] = off
    store["ids"] = np.asarray([k for x in ids for k in x], np.int32)
    store["frames"] = np.asarray([k for x in frames for k in x], np.int32)
    store["enc_lens"] = np.asarray(enc_lens, np.int32)
    store["min_margin"] = np.asarray(margins, np.float64)
    store["n_decisions"] = np.asarray(enc_lens, np.int32)
    store["proj"], store["f_rows"], store["feat_proj"] = proj, f_rows, feat_proj
    near = [b for b in range(rows) if margins[b] < NEAR_TIE]
    print(f"done in {time.time() - t0:.0f} s; rows with a greedy margin below {NEAR_TIE:g}: {near}", flush=True)
    name = "bench_k2_fp32.npz" if rows == 256 else f"bench_k2_fp32_{rows}.npz"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
