#!/usr/bin/env python
"""bench.py — RTFx of the MI355X FastConformer-RNNT path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (log-mel front-end -> 24-layer FastConformer encoder
-> joint projection -> batched greedy RNN-T decode) over one batch of 256 synthetic 10 s
utterances per GPU, inputs already resident in HBM when the timed region starts.  Weak scaling:
every rank processes its own 256 utterances (BASELINE.json configs[2]: 2048 = 8 x 256); with
N > 1 each step ends with the one collective of the path, an RCCL all_gather of the hypotheses.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime import capi                                  # noqa: E402
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M, TINY        # noqa: E402
from reazonspeech_amd.runtime.model import AsrModel                         # noqa: E402
from reazonspeech_amd.runtime.synth import synthetic_batch                  # noqa: E402
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer           # noqa: E402
from reazonspeech_amd.runtime.weights import synthetic_state_dict           # noqa: E402
from reazonspeech_amd.runtime import dist as rdist                          # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (guides/MI355X_MICROARCH.md)


def algorithmic_gflop_per_utt(cfg, tp, mean_tokens):
    """SURVEY.md §8(d) work model, evaluated for this config / T' / measured U"""
    d, f, c = cfg.d_model, cfg.ff_dim, cfg.sub_channels
    lin = 2 * (2 * d * f * 2) + 4 * 2 * d * d + 2 * d * 2 * d + 2 * d * d          # per frame per layer
    att = 3 * 2 * tp * d                                                              # ac + bd + pv per frame
    dw = 2 * cfg.conv_kernel * d
    enc = cfg.n_layers * tp * (lin + att + dw)
    t1, t2 = 4 * tp, 2 * tp
    sub = 2 * (t1 * 40 * c * 9 + t2 * 20 * c * 9 + t2 * 20 * c * c + tp * 10 * c * 9 + tp * 10 * c * c
               + tp * c * 10 * d)
    jenc = 2 * tp * d * cfg.joint_hidden
    H, J, V = cfg.pred_hidden, cfg.joint_hidden, cfg.n_logits
    dec = (tp + mean_tokens) * 2 * J * V + mean_tokens * (cfg.pred_layers * 2 * 4 * H * 2 * H + 2 * H * J)
    fe = 8 * tp * (5 * 512 * 9 + 3 * 257 + 2 * 600)
    return (enc + sub + jenc + dec + fe) / 1e9


def cpu_baseline(cfg, sd, audio, lens, seconds_budget=20.0, max_utt=8):
    """The reference's own CPU transcribe() cannot run here (NeMo absent, no checkpoint —
    BASELINE.md §4); timed instead: the repo's CPU oracle (fp32 torch + C greedy), driven like the
    reference — one utterance per call, 0.5 s padding — on the host cores of this node."""
    try:
        from oracle import model as om, greedy as og
    except Exception as e:                      # oracle is optional infrastructure for this leg
        return {"value": None, "unit": "x real-time", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    threads = torch.get_num_threads()
    done, audio_s = 0, 0.0
    t0 = time.perf_counter()
    for b in range(min(max_utt, audio.shape[0])):
        n = int(lens[b])
        wav = np.pad(audio[b, :n], 8000)
        f, el = om.forward_to_joint(cfg, sd, torch.from_numpy(wav)[None], torch.tensor([len(wav)]), "fp32")
        og.rnnt_greedy(cfg, sd, f.numpy(), el.numpy())
        done += 1
        audio_s += n / 16000.0
        if time.perf_counter() - t0 > seconds_budget:
            break
    dt = time.perf_counter() - t0
    return {"value": round(audio_s / dt, 3), "unit": "x real-time", "cores": threads, "kind": "port",
            "sample": f"{done} utterance(s) x 10 s, one per call (batch_size=1), fp32 torch CPU oracle + C greedy, "
                      f"{dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--tiny", action="store_true", help="debug: 2-layer toy config (NOT a valid bench line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--enc-streams", type=int, default=int(os.environ.get("RS_ENC_STREAMS", "1")),
                    help="experimental: encoders of consecutive batches on two streams (four resident batches)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run encoder and decode of each batch back to back on one stream")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        rdist.init("nccl")

    cfg = TINY if args.tiny else FASTCONFORMER_619M
    t0 = time.time()
    sd = synthetic_state_dict(cfg, seed=0)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device=f"cuda:{local_rank}")
    # two resident batches (different utterances): the pipelined path alternates between them
    bufs, lens_all = [], []
    n_sets = 4 if args.enc_streams == 2 else 2
    for k in range(n_sets):
        audio, lens = synthetic_batch(args.batch, args.seconds, seed=1234 + 17 * rank + 1000 * k)
        b = model.stage([audio[i, :lens[i]] for i in range(args.batch)],
                        buf=model.new_buffers(args.batch, int(args.seconds * 16000)))
        torch.cuda.synchronize()
        bufs.append(b)
        lens_all.append(lens)
    buf = bufs[0]
    audio0, lens0 = synthetic_batch(args.batch, args.seconds, seed=1234 + 17 * rank)
    setup_s = time.time() - t0
    pipelined = not args.no_pipeline

    def gather(bf):
        if world > 1:
            rdist.gather_hypotheses(bf.ids, bf.frames, bf.n_ids)

    def run_steps(n):
        if pipelined:
            model.run_pipelined(bufs, n, after_decode=None, enc_streams=args.enc_streams)
            for i in range(n):                      # one collective per step, as the path defines it
                gather(bufs[i % n_sets])
        else:
            for i in range(n):
                model.run_device(bufs[i % n_sets])
                gather(bufs[i % n_sets])

    run_steps(args.warmup)
    prof = not args.no_profile
    if prof:
        model.ctx.profile_reset()
        model.ctx.profile_enable(capi.PROF_GEMM)
    rdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    rdist.barrier()
    dt = time.perf_counter() - t0
    gemm = model.ctx.profile_read(capi.PROF_GEMM) if prof else None
    model.ctx.profile_enable(0)
    dt = rdist.max_over_ranks(dt)

    # PCIe-inclusive rate (never `value`): same steps, but every batch is copied from pinned host
    # memory first and its hypotheses are copied back after decode
    dt_host = None
    if pipelined and world == 1:
        model.run_pipelined(bufs, 2, from_host=True, enc_streams=args.enc_streams)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.run_pipelined(bufs, args.steps, from_host=True, enc_streams=args.enc_streams)
        torch.cuda.synchronize()
        dt_host = time.perf_counter() - t1

    n_ids = np.concatenate([b.n_ids.cpu().numpy() for b in bufs])
    mean_tokens = float(n_ids.mean())
    audio_seconds = sum(float(lens_all[i % n_sets].sum()) for i in range(args.steps)) / 16000.0 * world
    value = audio_seconds / dt

    if rank == 0:
        out = {
            "metric": "RTFx (audio-sec/wall-sec), FastConformer-RNNT 619M batch=256",
            "value": round(value, 1), "unit": "x real-time", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"FastConformer-RNNT {cfg.n_params() / 1e6:.0f}M, {args.batch} x "
                                   f"{args.seconds:g} s utterances per GPU (+0.5 s pad each side), greedy decode, "
                                   "random-init weights", "global_batch": args.batch * world,
                       "utterance_seconds": args.seconds, "parallelism": f"dp{world}",
                       "enc_frames": buf.tp_max, "mean_tokens_per_utt": round(mean_tokens, 1),
                       "max_tokens_per_utt": int(n_ids.max()),
                       "schedule": ("2-stage pipeline: encoder(i+1) || greedy decode(i) on two HIP streams"
                                    + (" (encoders of consecutive batches on two streams)" if args.enc_streams == 2 else ""))
                                   if pipelined else "sequential"},
            "setup_s": round(setup_s, 1),
        }
        if dt_host:
            out["value_pcie_inclusive"] = round(audio_seconds / dt_host, 1)
        gf = algorithmic_gflop_per_utt(cfg, buf.tp_max, mean_tokens)
        out["algorithmic_tflops_whole_path"] = round(gf * args.batch * world * args.steps / dt / 1e3, 1)
        if gemm and gemm["launches"]:
            per_launch_ms = gemm["ms"] / gemm["launches"]
            achieved = gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12
            traffic = None
            try:   # HBM bytes per launch from the committed PMC passes (cannot be collected in-process)
                with open(os.path.join(ROOT, "profiles", "gemm_traffic.json")) as fp:
                    traffic = json.load(fp)["hbm_bytes_per_launch"]
            except Exception:
                pass
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_mf16_kernel / gemm_bf16_kernel (all encoder linears)",
                               "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                               "traffic_unit": "HBM bytes per launch (PMC, profiles/gemm_traffic.json)",
                               "algorithmic_bytes_per_launch": round(gemm["bytes"] / gemm["launches"]),
                               "launches": gemm["launches"], "avg_launch_us": round(per_launch_ms * 1e3, 2),
                               "share_of_step": round(gemm["ms"] / (dt * 1e3), 3)}
        if world == 1 and not args.no_cpu_baseline and not args.tiny:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, audio0, lens0)
        elif world == 1 and args.tiny and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, audio0, lens0, seconds_budget=5.0, max_utt=2)
        print(json.dumps(out), flush=True)
    rdist.shutdown()


if __name__ == "__main__":
    main()
