"""Generate tests/golden/bench_espnet_fp32.npz — the float32 oracle's END-TO-END outputs for EVERY row of the ESPnet
benchmark batch (bench.py `configs.espnet_120m`: 256 x 10 s, seed 4242, each padded with the reference's (16000, 8000)
samples — pkg/espnet-asr/src/transcribe.py:10,69), at the 120M Conformer-Transducer geometry with the seeded synthetic
weights of bench.py.  Run in the BUILD container (CPU):

    python tests/golden/make_espnet_golden.py [--rows N] [--beam-rows N]

The oracle (oracle/espnet.py, float32) processes ONE utterance per call, like the reference drives Speech2Text (:69).  Stored:

  greedy (the `configs.espnet_120m` checkpoint: synthetic_state_dict_espnet(cfg, 0))
    ids / frames / offsets    ESPnet's greedy_search (one symbol per frame) of every row — oracle/rnnt_greedy.c with the tanh
                              joint on the oracle's own joint projection, which equals the torch restatement
                              (`greedy_torch`: asserted here on the first rows)
    enc_lens                  T'_b
    min_margin / n_decisions  the smallest top-1 minus top-2 joint-logit margin along the row's own decision path (float64
                              walk): a row whose margin is far above float32 reassociation noise MUST come out identical
                              from any float32 implementation
    proj                      f[T'][J] @ R[J][8] per row (R seeded N(0, 1) / sqrt(J)): fingerprint of every joint projection
    f_rows                    the joint projection itself for rows 0 and 1
    ctc_blank                 the CTC head's blank posterior per frame, every row (what `find_blank` thresholds: ctc.py:29-58)
    ctc_argmax                argmax of the CTC posteriors per frame, every row (int16)
  beam (the `configs.espnet_120m_beam20` checkpoint: blank_bias 16, dec_gain 8; same encoder weights, so the same joint
  projection) for the first `--beam-rows` rows (default 32), WHOLE utterances, beam 20, score_norm:
    f64_*                     oracle/espnet.py: default_beam_search_torch — the statement-for-statement restatement of ESPnet's
                              default_beam_search with Python-float (float64) score sums like upstream: labels, frames, score, pops
    c_*                       oracle/espnet_beam.c on the same projection (float32 sums in a fixed order; the bit-exact checker
                              of the device search): labels, frames, score, pops
  audio_sha256               checksum of the regenerated inputs

Consumers: tests/test_oracle_espnet_beam.py (CPU: the C checker against the float64 restatement over whole utterances at the
120M shape), tests/test_gpu_espnet.py (`-m gpu`: the float32 parity mode over all 256 rows; the beam search over 32 whole
rows), bench.py `configs.espnet_120m.parity`.
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime.config import ESPNET_CONFORMER_120M                     # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                           # noqa: E402
from reazonspeech_amd.runtime.weights_espnet import synthetic_state_dict_espnet      # noqa: E402
from oracle import espnet as oe, greedy as og                                        # noqa: E402

PADDING = (16000, 8000)
PROJ_SEED, PROJ_DIM = 20240927, 8
NEAR_TIE = 1e-3
BEAM, MAX_POPS = 20, 640
SEED, SECONDS = 4242, 10.0
BEAM_SD = dict(blank_bias=16.0, dec_gain=8.0)        # bench.py: espnet_beam_config


def projection(J):
    g = torch.Generator().manual_seed(PROJ_SEED)
    return (torch.randn((J, PROJ_DIM), generator=g, dtype=torch.float32) / J ** 0.5).numpy()


def greedy_margins(cfg, sd, f, n, ids, frames):
    """float64 walk along the greedy path (ESPnet greedy_search: one decision per frame): smallest top-1 minus top-2 margin"""
    H = cfg.pred_hidden
    emb = sd["decoder.embed.weight"].double()
    wd = sd["joint_network.lin_dec.weight"].double()
    wo, bo = sd["joint_network.lin_out.weight"].double(), sd["joint_network.lin_out.bias"].double()
    lstm = []
    for l in range(cfg.pred_layers):
        P = f"decoder.decoder.{l}."
        lstm.append((sd[P + "weight_ih_l0"].double(), sd[P + "weight_hh_l0"].double(), (sd[P + "bias_ih_l0"].double() + sd[P + "bias_hh_l0"].double())))
    st = [(torch.zeros(H, dtype=torch.float64), torch.zeros(H, dtype=torch.float64)) for _ in lstm]

    def step(tok):
        x = emb[tok]
        for l, (wi, wh, b) in enumerate(lstm):
            h, c = st[l]
            g = wi @ x + wh @ h + b
            i, fg, gg, o = g[:H].sigmoid(), g[H:2 * H].sigmoid(), g[2 * H:3 * H].tanh(), g[3 * H:].sigmoid()
            c = fg * c + i * gg
            h = o * c.tanh()
            st[l] = (h, c)
            x = h
        return wd @ x
    g = step(cfg.blank_id)
    fd = torch.from_numpy(f).double()
    emitted = dict(zip(frames, ids))
    worst = float("inf")
    for t in range(n):
        z = wo @ torch.tanh(fd[t] + g) + bo
        top = torch.topk(z, 2)
        worst = min(worst, float(top.values[0] - top.values[1]))
        k = int(top.indices[0])
        want = emitted.get(t, cfg.blank_id)
        if k != want:                      # the float64 walk itself disagrees: a tie at float32 resolution
            worst = 0.0
            k = want
        if k != cfg.blank_id:
            g = step(k)
    return worst


def main():
    rows, beam_rows = 256, 32
    if "--rows" in sys.argv:
        rows = int(sys.argv[sys.argv.index("--rows") + 1])
    if "--beam-rows" in sys.argv:
        beam_rows = int(sys.argv[sys.argv.index("--beam-rows") + 1])
    beam_rows = min(beam_rows, rows)
    cfg = ESPNET_CONFORMER_120M
    sd = synthetic_state_dict_espnet(cfg, 0)
    sd_beam = synthetic_state_dict_espnet(cfg, 0, **BEAM_SD)
    for k in sd:                     # the two checkpoints share every encoder / projection weight
        if k.startswith(("encoder.", "joint_network.lin_enc", "ctc.", "frontend.", "normalize.")):
            assert torch.equal(sd[k], sd_beam[k]), k
    R = projection(cfg.joint_hidden)
    audio, lens = synthetic_batch(256, SECONDS, seed=SEED)
    store = {"rows": np.int64(rows), "beam_rows": np.int64(beam_rows), "proj_seed": np.int64(PROJ_SEED), "near_tie": np.float64(NEAR_TIE),
             "beam": np.int64(BEAM), "max_pops": np.int64(MAX_POPS), "seed": np.int64(SEED),
             "audio_sha256": np.frombuffer(hashlib.sha256(audio.tobytes()).digest(), np.uint8)}
    l_pad = audio.shape[1] + sum(PADDING)
    tp_max = cfg.enc_frames(cfg.mel_frames(l_pad))
    ids, frames, enc_lens, margins = [], [], [], []
    proj = np.zeros((rows, tp_max, PROJ_DIM), np.float32)
    f_rows = np.zeros((2, tp_max, cfg.joint_hidden), np.float32)
    ctc_blank = np.zeros((rows, tp_max), np.float32)
    ctc_argmax = np.zeros((rows, tp_max), np.int16)
    bm = {k: [] for k in ("f64_ids", "f64_frames", "f64_score", "f64_pops", "c_ids", "c_frames", "c_score", "c_pops")}
    t0 = time.time()
    for b in range(rows):
        wav = np.pad(audio[b, :int(lens[b])], PADDING)
        out = oe.forward(cfg, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32")
        n = int(out["enc_lens"][0])
        f = out["joint_enc"].numpy()
        el = np.asarray([n], np.int32)
        hyp = og.rnnt_greedy(cfg, sd, f, el)[0]
        if b < 4:
            ref = oe.greedy_torch(cfg, sd, out["joint_enc"], out["enc_lens"])[0]
            assert (hyp[0], hyp[1]) == (ref[0], ref[1]), f"row {b}: rnnt_greedy.c != greedy_torch"
        ids.append(hyp[0]); frames.append(hyp[1]); enc_lens.append(n)
        margins.append(greedy_margins(cfg, sd, f[0], n, hyp[0], hyp[1]))
        proj[b, :n] = f[0, :n] @ R
        if b < 2:
            f_rows[b, :n] = f[0, :n]
        ctc_blank[b, :n] = out["ctc"][0, :n, cfg.blank_id].numpy()
        ctc_argmax[b, :n] = out["ctc"][0, :n].argmax(-1).numpy().astype(np.int16)
        if b < beam_rows:
            py = oe.default_beam_search_torch(cfg, sd_beam, out["joint_enc"], out["enc_lens"], beam_size=BEAM, score_norm=True, with_frames=True)[0]
            cc = og.espnet_beam(cfg, sd_beam, f, el, beam=BEAM, max_pops=MAX_POPS, out_cap=2 * tp_max + 16, with_frames=True)[0]
            bm["f64_ids"].append(list(py[0])); bm["f64_frames"].append(list(py[1])); bm["f64_score"].append(float(py[2])); bm["f64_pops"].append(int(py[3]))
            bm["c_ids"].append(list(cc[0])); bm["c_frames"].append(list(cc[1])); bm["c_score"].append(float(cc[2])); bm["c_pops"].append(int(cc[3]))
            print(f"  beam row {b}: f64 {len(py[0])} labels score {py[2]:.4f} pops {py[3]} | C {len(cc[0])} labels score {cc[2]:.4f} pops {cc[3]} | "
                  f"labels equal {list(py[0]) == list(cc[0])}", flush=True)
        if b % 8 == 7:
            print(f"{b + 1}/{rows} rows, {time.time() - t0:.0f} s, tokens/row {np.mean([len(x) for x in ids]):.1f}, min margin {min(margins):.2e}", flush=True)

    def ragged(name, lists, dtype):
        off = np.zeros(len(lists) + 1, np.int64)
        off[1:] = np.cumsum([len(x) for x in lists])
        store[name + "_offsets"] = off
        store[name] = np.asarray([k for x in lists for k in x], dtype)
    ragged("ids", ids, np.int32)
    store["frames"] = np.asarray([k for x in frames for k in x], np.int32)
    store["enc_lens"] = np.asarray(enc_lens, np.int32)
    store["min_margin"] = np.asarray(margins, np.float64)
    store["n_decisions"] = np.asarray(enc_lens, np.int32)
    store["proj"], store["f_rows"], store["ctc_blank"], store["ctc_argmax"] = proj, f_rows, ctc_blank, ctc_argmax
    for side in ("f64", "c"):
        ragged(f"beam_{side}_ids", bm[f"{side}_ids"], np.int32)
        store[f"beam_{side}_frames"] = np.asarray([k for x in bm[f"{side}_frames"] for k in x], np.int32)
        store[f"beam_{side}_score"] = np.asarray(bm[f"{side}_score"], np.float64)
        store[f"beam_{side}_pops"] = np.asarray(bm[f"{side}_pops"], np.int32)
    near = [b for b in range(rows) if margins[b] < NEAR_TIE]
    print(f"done in {time.time() - t0:.0f} s; rows with a greedy margin below {NEAR_TIE:g}: {near}", flush=True)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_espnet_fp32.npz")
    np.savez_compressed(out, **store)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
