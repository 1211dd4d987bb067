"""The model object `load_model()` returns: FastConformer-RNNT on one MI355X.

It owns (through torch) every device buffer — weights, activations, workspace — and drives the
three stage entry points of librs_asr.so on torch's current HIP stream.  It stands where NeMo's
`EncDecRNNTBPEModel` stands in the reference (pkg/nemo-asr/src/transcribe.py:26-28, :48-53)
and exposes the one attribute the reference's post-processing touches: `.tokenizer`
(pkg/nemo-asr/src/decode.py:41,47).
"""
import gc
import os
import queue
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import capi
from .config import ModelConfig
from .weights import prepare_weights, DEFAULT_POS_CAP


@dataclass
class DecodedBatch:
    """host-side result of one batch: token ids and emission frames per utterance (+ the log-probability of the
    hypothesis when it came from the beam search)"""
    ids: List[List[int]]
    frames: List[List[int]]
    enc_lens: List[int]
    scores: Optional[List[float]] = None
    degraded: Optional[List[bool]] = None     # per utterance: the requested search overflowed its bound and a weaker one was used (espnet `_search`)


def alsd_label_budget(t_frames: int, max_target_len) -> int:
    """labels a hypothesis may emit beyond the frames (oracle/alsd.py: float = multiple of T', int = absolute)"""
    return int(max_target_len * t_frames) if isinstance(max_target_len, float) else int(max_target_len)


class _Buffers:
    """activation / workspace buffers for one (B, Lmax) geometry"""

    def __init__(self, model, B, l_max):
        cfg, dev, ctx = model.cfg, model.device, model.ctx
        self.B, self.l_max = B, l_max
        self.l_pad = l_max + model.pad_left + model.pad_right
        self.t_max = max(ctx.mel_frames(self.l_pad), 1)
        self.tp_max = max(ctx.enc_frames(self.t_max), 1)
        model.ensure_pos_cap(self.tp_max)
        self.u_max = cfg.label_cap(self.tp_max)
        i32, f32 = torch.int32, torch.float32
        self.audio = torch.zeros((B, l_max), dtype=f32, device=dev)
        self.lens = torch.zeros((B,), dtype=i32, device=dev)
        self.feats = torch.empty((B, self.t_max, cfg.n_mels), dtype=f32, device=dev)
        self.n_frames = torch.zeros((B,), dtype=i32, device=dev)
        self.joint_enc = torch.empty((B, self.tp_max, cfg.joint_hidden), dtype=f32, device=dev)
        self.enc_lens = torch.zeros((B,), dtype=i32, device=dev)
        self.ids = torch.zeros((B, self.u_max), dtype=i32, device=dev)
        self.frames = torch.zeros((B, self.u_max), dtype=i32, device=dev)
        self.n_ids = torch.zeros((B,), dtype=i32, device=dev)
        self.scores = torch.zeros((B,), dtype=f32, device=dev)
        self.pops = torch.zeros((B,), dtype=i32, device=dev)     # "beam": prediction-network evaluations per utterance
        self.ws_alsd = None              # beam-search scratch (grows with beam and alignment length): on first use
        self.ws = torch.empty((ctx.workspace_bytes(B, self.l_pad),), dtype=torch.uint8, device=dev)
        # the decoder of batch i overlaps the encoder of batch i+1 in the pipelined path: own scratch
        self.ws_dec = torch.empty((ctx.workspace_bytes(B, 16),), dtype=torch.uint8, device=dev)
        # pinned staging for the host boundary (inputs, and the hypotheses on the way back)
        self.h_audio = torch.zeros((B, l_max), dtype=f32).pin_memory()
        self.h_lens = torch.zeros((B,), dtype=i32).pin_memory()
        self.h_out = None
        self.step = -1                   # index of the pipeline step this buffer set currently carries (run_pipelined)
        self._model = model

    def narrow(self, l_max):
        """The same memory with the geometry of a batch whose longest utterance has `l_max` samples: tight padding for a
        group of short utterances without another allocation (the kernels take extents and strides as arguments and mask
        by per-utterance length, so results do not depend on the padded extent)."""
        l_max = (max(int(l_max), 1) + 63) // 64 * 64
        if l_max >= self.l_max:
            return self
        return _BufView(self, l_max)


class _BufView:
    """a `_Buffers` re-viewed for a smaller padded extent (see `_Buffers.narrow`); shares every allocation"""

    def __init__(self, base, l_max):
        model = base._model
        cfg, ctx = model.cfg, model.ctx
        self.base, self.B, self.l_max = base, base.B, l_max
        self.l_pad = l_max + model.pad_left + model.pad_right
        self.t_max = max(ctx.mel_frames(self.l_pad), 1)
        self.tp_max = max(ctx.enc_frames(self.t_max), 1)
        self.u_max = cfg.label_cap(self.tp_max)
        B = self.B

        def cut(t, *shape):
            n = 1
            for d in shape:
                n *= d
            return t.view(-1)[:n].view(*shape)

        # device and pinned staging re-viewed as contiguous [B][l_max] (the kernels take the row pitch as an argument):
        # the H2D copy of a narrowed batch is then ONE contiguous asynchronous copy — a column slice of the full-pitch
        # matrices made torch stage 164 MB through a pageable temporary, synchronously (profiles/r03x_host_timeline_ragged.txt)
        self.audio, self.lens = cut(base.audio, B, l_max), base.lens
        self.feats = cut(base.feats, B, self.t_max, cfg.n_mels)
        self.n_frames = base.n_frames
        self.joint_enc = cut(base.joint_enc, B, self.tp_max, cfg.joint_hidden)
        self.enc_lens = base.enc_lens
        self.ids = cut(base.ids, B, self.u_max)
        self.frames = cut(base.frames, B, self.u_max)
        self.n_ids, self.scores, self.pops = base.n_ids, base.scores, base.pops
        self.ws, self.ws_dec = base.ws, base.ws_dec
        self.h_audio, self.h_lens = cut(base.h_audio, B, l_max), base.h_lens
        self.h_out = None
        self.step = -1

    def narrow(self, l_max):
        return self.base.narrow(l_max)

    @property
    def ws_alsd(self):
        return self.base.ws_alsd

    @ws_alsd.setter
    def ws_alsd(self, v):
        self.base.ws_alsd = v


class AsrModel:
    def __init__(self, cfg: ModelConfig, state_dict, tokenizer, device="cuda", pos_cap: int = DEFAULT_POS_CAP,
                 pad_seconds: float = 0.5, precision: str = "bf16", pad_samples=None):
        """precision: "bf16" = the throughput mode (bf16 GEMM operands, float32 accumulation and residual stream);
        "fp32" = the parity mode: float32 weights, activations and arithmetic end to end, what the reference computes
        (pkg/nemo-asr/src/transcribe.py:26-28, :48-53) — about 20x slower, 2.4 GB more weights."""
        cfg.validate()
        if precision not in ("bf16", "fp32", "fp32x3"):
            raise ValueError(f"precision must be 'bf16', 'fp32' or 'fp32x3', not {precision!r}")
        # "fp32x3": the float32 mode (float32 weights, activations, accumulation, IEEE exp / divide) with every float32 PRODUCT of its
        # GEMMs formed from three bf16 matrix-core terms (csrc/k_f32.hip X3: hi / lo split, 16 mantissa bits per operand) — 2x the
        # float32 mode's speed; not an IEEE chain, but held to the same 256-row goldens (ids identical on every row of all three)
        self.x3 = precision == "fp32x3"
        precision = "fp32" if self.x3 else precision
        self.precision = precision
        if not torch.cuda.is_available():
            raise RuntimeError("reazonspeech_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is "
                               "False and there is no CPU fallback for this path")
        self.cfg = cfg
        self.tokenizer = tokenizer
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"device {device!r}: only ROCm ('cuda[:N]') devices are supported")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        self.pad_left = self.pad_right = int(pad_seconds * cfg.sample_rate)   # audio.py:80-82 via decode.py:4
        if pad_samples is not None:          # asymmetric padding in samples (espnet: PADDING = (16000, 8000), transcribe.py:10,69)
            self.pad_left, self.pad_right = int(pad_samples[0]), int(pad_samples[1])
        self._bufs = {}
        self._dec_lanes = []            # [(context, stream)] of the decode lanes beyond what was needed so far
        self._streams = None
        self._streams_prio = None
        self.pos_cap = 0
        with torch.cuda.device(self.device):
            self.ctx = capi.Context(cfg, index)
            if getattr(cfg, "family", "") == "k2":
                from .k2_weights import prepare_weights_k2
                self._k2_sd = state_dict          # the position tables are re-projected when a longer utterance arrives
                self._upload_k2(prepare_weights_k2(cfg, state_dict, pos_cap, f32=precision == "fp32"))
                self.pos_cap = pos_cap
            elif cfg.espnet:
                from .weights_espnet import prepare_weights_espnet
                self._upload(prepare_weights_espnet(cfg, state_dict, pos_cap, f32=precision == "fp32"))
            else:
                self._upload(prepare_weights(cfg, state_dict, pos_cap, f32=precision == "fp32"))
            if precision == "fp32":
                self.ctx.set_option("precision_f32", 1)
                self.ctx.set_option("gemm_f32_x3", 1 if self.x3 else 0)

    # ------------------------------------------------------------------------------------------
    def _upload(self, tensors):
        dev = {}
        for name, t in tensors.items():
            dev[name] = t.to(self.device, non_blocking=False).contiguous()
            self.ctx.set_tensor(name, dev[name])
        self._pos_w = [dev[f"L{i}.att.pos.w"] for i in range(self.cfg.n_layers)]
        self._set_pos_tables(dev["pos.table"], dev.get("pos.table.f32"))

    def _upload_k2(self, tensors):
        for name, t in tensors.items():
            dev = t.to(self.device, non_blocking=False).contiguous()
            for c in self._contexts():
                c.set_tensor(name, dev)
        for c in self._contexts():
            c.finalize()
            if self.precision == "fp32":
                c.set_option("precision_f32", 1)
                c.set_option("gemm_f32_x3", 1 if self.x3 else 0)

    def _contexts(self):
        return [self.ctx] + [c for c, _ in self._dec_lanes]

    def _set_pos_tables(self, table, table_f32=None):
        """register the relative-position table (bf16 [2*cap-1][d]) and the derived per-layer tensors: the table
        projected by every layer's linear_pos, computed once with the library's own GEMM (same kernel and row
        arithmetic as the per-call projection it replaces, so results are bit-identical) — 24 x [2*cap-1][d] bf16
        = 100 MB at cap 1024"""
        projs = []
        for i in range(self.cfg.n_layers):
            proj = torch.empty_like(table)
            self.ctx.gemm(table, self._pos_w[i], proj, flags=0)
            projs.append(proj)
        torch.cuda.synchronize(self.device)
        for c in self._contexts():
            c.set_tensor("pos.table", table)
            if table_f32 is not None:            # float32 parity mode: its layers project the rows they need per call
                c.set_tensor("pos.table.f32", table_f32)
            for i, proj in enumerate(projs):
                c.set_tensor(f"L{i}.att.pos_proj", proj)
            c.finalize()
            if self.precision == "fp32":         # (rs_finalize ran again: the option survives, set it anyway for new contexts)
                c.set_option("precision_f32", 1)
                c.set_option("gemm_f32_x3", 1 if self.x3 else 0)
        self.pos_cap = (table.shape[0] + 1) // 2

    def ensure_pos_cap(self, tp: int):
        """Long-form audio: the reference hands a whole file to the model as ONE utterance
        (pkg/nemo-asr/src/transcribe.py:44-53), so T' is unbounded.  The resident position tables cover T' up to
        `pos_cap`; a longer utterance grows them (next power of two) instead of failing."""
        if getattr(self.cfg, "family", "") == "k2":
            # the 50 Hz stack of a Zipformer has twice the output frames (+ 1): its position tables must cover that
            need = 2 * int(tp) + 2
            if need <= self.pos_cap:
                return
            from .k2_weights import prepare_weights_k2
            cap = 1 << (need - 1).bit_length()
            torch.cuda.synchronize(self.device)
            with torch.cuda.device(self.device):
                tensors = prepare_weights_k2(self.cfg, self._k2_sd, cap)
                self._upload_k2({k: v for k, v in tensors.items() if k.endswith("attw.pos_proj")})
            self.pos_cap = cap
            return
        if tp <= self.pos_cap:
            return
        from .weights import rel_pos_table
        cap = 1 << (int(tp) - 1).bit_length()
        torch.cuda.synchronize(self.device)          # nothing may still read the tables being replaced
        with torch.cuda.device(self.device):
            host = torch.from_numpy(rel_pos_table(self.cfg, cap))
            table = host.to(torch.bfloat16).to(self.device).contiguous()
            self._set_pos_tables(table, host.to(self.device).contiguous() if self.precision == "fp32" else None)

    BUCKET = 16000      # cached buffer sets are sized in whole seconds of audio

    def _decode_policy(self, ctx, B, pipelined, lanes=1):
        """Pick the decode kernel family for a call (all are bit-identical; tests/test_gpu_fullsize.py).  Measured on
        MI355X (profiles/r02o_pipeline_decode_variants_ab.txt, r02q_small_batch_decode_ab.txt, r02zz_decode_family_ab.txt):
          * ONE decode stream next to the encoder is on the critical path: what a step costs there is the number of
            launches and of workgroups that must find free CUs, not the length of each kernel on an idle chip: at
            B = 256 the wide-tile kernels with the exact joint (5 launches, 40-workgroup LSTM) give 65.5 ms per step,
            screened / narrow 66.5-68.5;
          * TWO decode lanes have an encoder period of slack each: then the family with the least exact-f32 work wins,
            because its CU time is what the GEMMs lose (screened joint + narrow tiles 59.0 ms, wide / exact 60.0);
          * small batches are decode-bound: narrow tiles + exact joint win (B = 32: 19.6 ms vs 25.8 wide, 22.6 screened);
          * big batches on an otherwise idle chip (sequential schedule): screened joint + narrow tiles (79.9 vs 84.7 ms).
        $RS_DECODE_SCREEN / $RS_DECODE_NARROW override (A/B runs)."""
        if "RS_DECODE_SCREEN" in os.environ or "RS_DECODE_NARROW" in os.environ:
            return
        if self is not None and self.cfg.decoding == "alsd" and B * self.cfg.beam_size >= 256:
            # ALSD runs the LSTM over the hundreds of hypothesis rows that took a label (about half of B x beam per alignment
            # step): that is throughput work, and the wide tiles re-read the weights a quarter as often as the narrow ones —
            # 133 vs 160 ms per step at B = 256, beam 4 (profiles/r05f_alsd_narrow_ab.txt); results are bit-identical
            ctx.set_option("decode_narrow", 0)
            ctx.set_option("decode_screen", 0)
            return
        if self is not None and self.cfg.decoding == "beam":          # (the screened joint serves the greedy searches only)
            ctx.set_option("decode_screen", 0)
            ctx.set_option("decode_narrow", 0 if (pipelined and lanes == 1 and B >= 128) else 1)
            return
        big = B >= 128
        critical = pipelined and lanes == 1
        ctx.set_option("decode_narrow", 0 if (critical and big) else 1)
        ctx.set_option("decode_screen", 1 if (big and not critical) else 0)

    def buffers(self, B, l_max) -> _Buffers:
        """a cached buffer set for B utterances of up to l_max samples.  Lengths are bucketed to whole seconds and a
        larger cached set of the same B is reused: the kernels mask by per-utterance length, so the result does not
        depend on the padded extent (tests: batch invariance), and consecutive transcribe() calls on clips of
        different lengths stop re-allocating pinned staging + workspace every time."""
        l_max = (max(int(l_max), 1) + self.BUCKET - 1) // self.BUCKET * self.BUCKET
        # any cached set that is long enough will do: `stage` narrows it to the batch at hand (`_Buffers.narrow`), so a
        # longer set costs nothing per call and a folder of files of mixed lengths settles on ONE set instead of cycling
        # through the four cached geometries (per-call latency by length: profiles/r03x_varied_lengths_latency_reuse.txt —
        # it follows frames + emitted tokens, e.g. 7.8 ms for 2.2 s, 10 ms for 5 s, 29 ms for 29 s of noise)
        fits = [k for k in self._bufs if k[0] == B and l_max <= k[1]]
        if fits:
            key = min(fits, key=lambda k: k[1])
            self._bufs[key] = self._bufs.pop(key)          # most recently used last
            return self._bufs[key]
        key = (B, l_max)
        if len(self._bufs) >= 4:           # keep HBM bounded: drop the least recently used geometry
            self._bufs.pop(next(iter(self._bufs)))
        with torch.cuda.device(self.device):
            self._bufs[key] = _Buffers(self, B, l_max)
        return self._bufs[key]

    @property
    def n_params(self):
        return self.cfg.n_params()

    # ------------------------------------------------------------------------------------------
    def run_device(self, buf: _Buffers, want_enc: Optional[torch.Tensor] = None):
        """front-end -> encoder -> greedy decode on data already in `buf.audio`/`buf.lens` (HBM).
        Outputs land in buf.ids / buf.frames / buf.n_ids.  rs_rnnt_greedy synchronises the stream."""
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            self._decode_policy(self.ctx, buf.B, pipelined=False)
            self.ctx.frontend(buf.audio, buf.lens, self.pad_left, self.pad_right, buf.t_max, buf.feats,
                              buf.n_frames, buf.ws, stream)
            self.ctx.encoder(buf.feats, buf.n_frames, buf.B, buf.t_max, want_enc, buf.joint_enc, buf.enc_lens,
                             buf.ws, stream)
            self.decode(self.ctx, buf, buf.ws, stream)

    def decode(self, ctx, buf: _Buffers, ws, stream, max_pops=None, decoding=None):
        """stage 3 on `stream`: the checkpoint's decoding strategy (cfg.decoding).  Greedy fills buf.ids / buf.frames
        (emission frames); ALSD fills buf.ids / buf.frames (alignment steps i = frame + labels before) / buf.scores.
        Both synchronise the stream."""
        cfg = self.cfg
        if max_pops is not None or decoding is not None:     # per-call overrides (the caller's retry policy): never written into self.cfg
            cfg = cfg.with_(**({"beam_max_pops": int(max_pops)} if max_pops is not None else {}), **({"decoding": decoding} if decoding is not None else {}))
        if cfg.decoding == "beam":
            n = ctx.beam_workspace_bytes(buf.B, cfg.beam_size, buf.tp_max, cfg.beam_max_pops)
            if buf.ws_alsd is None or buf.ws_alsd.numel() < n:
                buf.ws_alsd = torch.empty((n,), dtype=torch.uint8, device=self.device)
            ctx.rnnt_beam(buf.joint_enc, buf.enc_lens, buf.B, buf.tp_max, cfg.beam_size, cfg.beam_score_norm, cfg.beam_max_pops,
                          buf.ids, buf.n_ids, buf.scores, buf.pops, buf.ws_alsd, stream, frames=buf.frames)
            return
        if cfg.decoding != "alsd":
            ctx.rnnt_greedy(buf.joint_enc, buf.enc_lens, buf.B, buf.tp_max, buf.u_max, buf.ids, buf.frames, buf.n_ids,
                            ws, stream)
            return
        n = ctx.alsd_workspace_bytes(buf.B, cfg.beam_size, buf.tp_max, cfg.alsd_max_target_len)
        if buf.ws_alsd is None or buf.ws_alsd.numel() < n:       # (a narrowed view may have sized it for a shorter batch)
            buf.ws_alsd = torch.empty((n,), dtype=torch.uint8, device=self.device)
        ctx.rnnt_alsd(buf.joint_enc, buf.enc_lens, buf.B, buf.tp_max, cfg.beam_size, cfg.alsd_max_target_len,
                      cfg.beam_score_norm, False, buf.ids, buf.frames, buf.n_ids, buf.scores, buf.ws_alsd, stream)

    # ------------------------------------------------------------------------------------------
    def run_encoder(self, buf: _Buffers, stream, ctx=None):
        """front-end + encoder of one batch on `stream` (asynchronous)"""
        ctx = ctx or self.ctx
        ctx.frontend(buf.audio, buf.lens, self.pad_left, self.pad_right, buf.t_max, buf.feats, buf.n_frames,
                     buf.ws, stream)
        ctx.encoder(buf.feats, buf.n_frames, buf.B, buf.t_max, None, buf.joint_enc, buf.enc_lens, buf.ws, stream)

    def run_pipelined(self, bufs: Sequence[_Buffers], steps: int, after_decode=None, from_host: bool = False,
                      dec_streams: int = 1, before_encoder=None, fill=None, enc_streams: Optional[int] = None):
        """Process `steps` batches as a pipeline with up to len(bufs) batches in flight: the throughput-bound front-end +
        encoder of batch i+1 (i+2) runs on one HIP stream while the latency-bound decode of batch i (and i+1) — a
        dependency chain of small launches that leaves most CUs idle — runs on one (two) more streams, each driven by a
        worker thread (ctypes releases the GIL; the decode entry points synchronise only their own stream).  Every batch
        is fully decoded when this returns.

        bufs            resident buffer sets; step i uses bufs[i % len(bufs)] (or what `fill` returned for it) and waits
                        for that set's previous decode.  n decode lanes (`dec_streams`; batch i goes to lane i mod n)
                        need at least n + 1 sets.
        after_decode    `after_decode(buf)` on the decode worker right after batch `buf.step` is decoded (and, with
                        `from_host`, its hypotheses are in `buf.h_out` on the host).  Hooks run in batch order whichever
                        lane finishes first (a hook may issue a collective: every rank has to issue them in the same
                        order, from one thread at a time).
        from_host       every batch is first copied from its pinned host buffer (H2D on the encoder stream) and its
                        hypotheses are copied back after decode: the host-to-host boundary of SURVEY.md §8(d).
        fill            `fill(i, buf) -> buf or a narrowed view of it`: a stager thread calls it for step i as soon as the
                        buffer set's previous use is over, to put batch i into the pinned staging buffers (implies
                        `from_host`) while the GPU works on the batches before it; the encoder of step i waits for it.
        before_encoder  `before_encoder(i)` on the caller's thread right before batch i's encoder is enqueued.
        enc_streams     encoder lanes: consecutive batches' front-end + encoder alternate between that many HIP streams (each
                        batch has its own workspace in its buffer set); needs dec_streams + enc_streams buffer sets.  None:
                        `encoder_lanes()` = ONE lane ($RS_ENC_STREAMS overrides) — two lanes were measured slower at B = 256
                        (profiles/r04f_ab_RS_ENC_STREAMS.txt) AND at B = 32 / 64 / 128, whose launches leave a third to half of
                        the CUs idle (profiles/r06_11_small_batch_enc_lanes_ab.txt).
                        (Also measured and dropped in round 6: the LAST batch's greedy decode split by rows over the decode
                        lanes to shorten the drain — 55.15 vs 55.45 ms per step over 20 steps, no gain.)"""
        nb = len(bufs)
        if enc_streams is None:
            enc_streams = self.encoder_lanes(bufs[0].B, nb, dec_streams)
        assert enc_streams >= 1 and (enc_streams == 1 or nb >= dec_streams + enc_streams), "n encoder lanes + m decode lanes need n + m buffer sets"
        assert nb >= 2 or steps <= 1, "the pipeline needs two buffer sets"
        assert dec_streams >= 1 and (dec_streams == 1 or nb >= dec_streams + 1), "n decode streams need n + 1 buffer sets"
        from_host = from_host or fill is not None
        with torch.cuda.device(self.device):
            # the decode chain of ONE stream is latency-critical (its workgroups should take the first free slots: high
            # priority); with several decode lanes each has whole extra encoder periods of slack and normal priority
            # leaves the GEMM rounds alone (profiles/r02w_bench_ab.txt)
            dec_prio = int(os.environ.get("RS_DECODE_PRIORITY", "0" if dec_streams >= 2 else "-1"))
            if self._streams is None or self._streams_prio != dec_prio:
                enc = self._streams[0] if self._streams is not None else torch.cuda.Stream(device=self.device)
                self._streams = (enc,)
                self._streams_prio = dec_prio
                self._dec_lanes = [(c, None) for c, _ in self._dec_lanes]       # keep the contexts, remake the streams
            while len(self._streams) < enc_streams:
                self._streams = self._streams + (torch.cuda.Stream(device=self.device),)
            enc_lanes = self._streams[:enc_streams]
            dec_cus = int(os.environ.get("RS_DECODE_CUS", "0"))
            while len(self._dec_lanes) < dec_streams:
                self._dec_lanes.append((self.ctx.clone(), None))
            dec_lanes = []
            for k in range(dec_streams):
                ctx_d, st = self._dec_lanes[k]
                if st is None:
                    if dec_cus > 0:
                        # decode confined to a slice of the chip (A/B knob): raw HIP stream with a CU mask
                        raw = capi.create_stream(self.device.index, dec_cus, self.ctx.n_cus(), dec_prio)
                        st = torch.cuda.ExternalStream(raw, device=self.device)
                    else:
                        st = torch.cuda.Stream(device=self.device, priority=dec_prio)
                    self._dec_lanes[k] = (ctx_d, st)
                self._decode_policy(ctx_d, bufs[0].B, pipelined=True, lanes=dec_streams)
                dec_lanes.append((ctx_d, st))
            for es in enc_lanes:
                es.wait_stream(torch.cuda.current_stream())
            queues = [queue.Queue() for _ in dec_lanes]
            done = [threading.Event() for _ in range(steps)]
            hooked = [threading.Event() for _ in range(steps)]
            staged = [threading.Event() for _ in range(steps)]
            views = [None] * steps
            errors = []
            stop = threading.Event()

            def worker(lane):
                ctx_d, stream = dec_lanes[lane]
                jobs = queues[lane]
                with torch.cuda.device(self.device):
                    while True:
                        item = jobs.get()
                        if item is None:
                            return
                        i, buf, ev = item
                        try:
                            stream.wait_event(ev)
                            # decode scratch lives past the encoder's scratch in buf.ws_dec
                            self.decode(ctx_d, buf, buf.ws_dec, stream.cuda_stream)
                            if from_host:
                                with torch.cuda.stream(stream):
                                    buf.h_out = (buf.n_ids.cpu(), buf.ids.cpu(), buf.frames.cpu(), buf.enc_lens.cpu(),
                                                 buf.scores.cpu() if self.cfg.has_scores else None)
                            if after_decode is not None:
                                if i > 0:
                                    hooked[i - 1].wait()
                                after_decode(buf)
                        except Exception as e:          # surfaced on the caller's thread below
                            errors.append(e)
                        finally:
                            hooked[i].set()
                            done[i].set()

            def stager():
                # host side of the pipeline: batch i goes into its pinned staging buffers as soon as the buffer set is
                # free again (its previous batch decoded, hence its H2D long finished)
                for i in range(steps):
                    try:
                        while i >= nb and not done[i - nb].wait(0.05):
                            if stop.is_set():
                                return
                        if errors or stop.is_set():
                            return
                        views[i] = fill(i, bufs[i % nb])
                    except Exception as e:
                        errors.append(e)
                    finally:
                        staged[i].set()

            threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(len(dec_lanes))]
            if fill is not None:
                threads.append(threading.Thread(target=stager, daemon=True))
            for th in threads:
                th.start()
            # whatever happens while batches are being enqueued (an RsError from the encoder, an allocation failure, an exception
            # of the caller's before_encoder hook), the workers and the stager are always released and joined before it propagates
            try:
                for i in range(steps):
                    if fill is not None:
                        staged[i].wait()
                        if errors:
                            break
                        buf = views[i]
                    else:
                        buf = bufs[i % nb]
                        if i >= nb:
                            done[i - nb].wait()           # this buffer set's previous decode must be finished
                    buf.step = i
                    if before_encoder is not None:
                        before_encoder(i)
                    enc_stream = enc_lanes[i % len(enc_lanes)]
                    if from_host:
                        with torch.cuda.stream(enc_stream):          # only the columns this batch's geometry reads
                            w = buf.l_max
                            buf.audio[:, :w].copy_(buf.h_audio[:, :w], non_blocking=True)
                            buf.lens.copy_(buf.h_lens, non_blocking=True)
                    self.run_encoder(buf, enc_stream.cuda_stream)
                    ev = torch.cuda.Event()
                    ev.record(enc_stream)
                    queues[i % len(queues)].put((i, buf, ev))          # batch i goes to decode lane i mod lanes
            finally:
                stop.set()              # releases the stager
                for q in queues:
                    q.put(None)
                for th in threads:
                    th.join()
            for es in enc_lanes:
                torch.cuda.current_stream().wait_stream(es)
            for _, ds in dec_lanes:
                torch.cuda.current_stream().wait_stream(ds)
            if errors:
                raise errors[0]

    def encoder_lanes(self, B: int, n_sets: int, dec_streams: int) -> int:
        """encoder lanes `run_pipelined` uses for batches of B utterances when the caller does not say ($RS_ENC_STREAMS overrides)"""
        e = os.environ.get("RS_ENC_STREAMS")
        want = int(e) if e else 1
        return max(1, min(want, n_sets - dec_streams))

    def new_buffers(self, B, l_max) -> _Buffers:
        """an un-cached buffer set (the pipelined path keeps two batches in flight)"""
        with torch.cuda.device(self.device):
            return _Buffers(self, B, (max(int(l_max), 1) + 63) // 64 * 64)

    def fill_host(self, waveforms: Sequence[np.ndarray], buf):
        """copy host waveforms (16 kHz mono float32, un-padded) into the buffer set's pinned staging buffers; the tail of
        every row is zeroed (the kernels mask by length, this only keeps the padding deterministic).  Returns the view of
        `buf` narrowed to this batch's longest utterance."""
        assert len(waveforms) <= buf.B
        longest = max((len(w) for w in waveforms), default=0)
        view = buf.narrow(longest)
        assert view.l_max >= longest
        # one native call for the whole batch (it runs without the interpreter lock: a per-utterance numpy copy would
        # queue behind whatever Python thread holds it — the ids -> text post-processing of an earlier batch)
        capi.host_stage_rows(view.h_audio, view.l_max, waveforms, buf.h_lens)
        return view

    def stage(self, waveforms: Sequence[np.ndarray], l_max: Optional[int] = None, buf: Optional[_Buffers] = None) -> _Buffers:
        """copy host waveforms (16 kHz mono float32, un-padded) into a pinned buffer and on to HBM"""
        B = len(waveforms)
        longest = max((len(w) for w in waveforms), default=0)
        l_max = max(int(l_max or 0), longest, 1)
        l_max = (l_max + 63) // 64 * 64          # rows stay 256-B aligned
        if buf is None:
            buf = self.buffers(B, l_max)
        assert buf.B == B and buf.l_max >= longest
        buf = buf.narrow(l_max)                  # the set itself when it has exactly this extent, else a tight view of it
        ha = buf.h_audio.numpy()
        hl = buf.h_lens.numpy()
        for b, w in enumerate(waveforms):
            n = len(w)
            ha[b, :n] = np.asarray(w, dtype=np.float32)
            ha[b, n:] = 0.0
            hl[b] = n
        with torch.cuda.device(self.device):
            buf.audio.copy_(buf.h_audio, non_blocking=True)
            buf.lens.copy_(buf.h_lens, non_blocking=True)
        return buf

    def collect(self, buf, host=None, decoding=None) -> DecodedBatch:
        """hypotheses of a decoded batch as host lists; `host` = the (n_ids, ids, frames, enc_lens, scores) tensors a
        pipeline worker already copied back (buf.h_out), otherwise they are fetched here; `decoding`: the search that
        produced them when it was overridden for the call (see `decode`)"""
        decoding = decoding or self.cfg.decoding
        if host is None:
            host = (buf.n_ids.cpu(), buf.ids.cpu(), buf.frames.cpu(), buf.enc_lens.cpu(),
                    buf.scores.cpu() if decoding in ("alsd", "beam") else None)
        n, ids, frames, el = (t.numpy() for t in host[:4])
        if decoding == "alsd":      # alignment step i = frame + labels emitted before
            frames = frames - np.arange(frames.shape[1], dtype=frames.dtype)[None, :]
            scores = host[4].numpy().tolist()
        elif decoding == "beam":    # frames = the frame each label was appended at ([UPSTREAM] NeMo Hypothesis.timestep)
            scores = host[4].numpy().tolist()
        else:
            scores = None
        return DecodedBatch([ids[b, :n[b]].tolist() for b in range(buf.B)],
                            [frames[b, :n[b]].tolist() for b in range(buf.B)], el.tolist(), scores)

    def transcribe_waveforms_sharded(self, waveforms: Sequence[np.ndarray], max_batch: int = 256) -> DecodedBatch:
        """SPMD form of `transcribe_waveforms` for one process per GPU (`torch.distributed` initialised, RCCL):
        every rank passes the same list, decodes its length-balanced shard on its own GPU and receives all
        hypotheses (ids, frames, encoder lengths and — beam search — scores) in the caller's order after the path's one
        collective (runtime/dist.py: sharded_decode).  With a single process it is `transcribe_waveforms`."""
        from . import dist as rdist

        def run_local(indices, batch):
            # `batch`: the chunk size of the dealing plan (ragged input is dealt as length-sorted chunks, balanced over the
            # ranks: runtime/dist.py shard_balanced) — this rank's batches are exactly its chunks
            res = self.transcribe_waveforms([waveforms[i] for i in indices], max_batch=min(max_batch, batch))
            return res.ids, res.frames, res.enc_lens, res.scores

        ids, frames, enc_lens, scores = rdist.sharded_decode([len(w) for w in waveforms], run_local, max_batch=max_batch)
        return DecodedBatch(ids, frames, enc_lens, scores if self.cfg.has_scores else None)

    LONGEST_FIRST = True    # batch order of a long list (A/B hook of scripts/ragged_order_ab.py)
    POOL_SETS = 4       # resident batches of the host-to-host pipeline (encoder(i+2) || decode(i+1), decode(i) + one being staged)

    def _pool(self, B, l_max, n_sets):
        """buffer sets of the host-to-host pipeline, kept across calls (pinned staging is expensive to allocate);
        bucketed to whole seconds like `buffers`, one geometry at a time"""
        l_max = (max(int(l_max), 1) + self.BUCKET - 1) // self.BUCKET * self.BUCKET
        key = getattr(self, "_pool_key", None)
        if key is None or key[0] != B or not (l_max <= key[1] <= 2 * l_max) or len(self._pool_sets) < n_sets:
            self._pool_sets = []              # drop the old geometry before allocating the new one
            self._pool_key = (B, l_max)
            with torch.cuda.device(self.device):
                self._pool_sets = [_Buffers(self, B, l_max) for _ in range(n_sets)]
        return self._pool_sets[:n_sets]

    def transcribe_waveforms(self, waveforms: Sequence[np.ndarray], max_batch: int = 256, on_batch=None) -> DecodedBatch:
        """host float32 waveforms -> token ids / frames (the batched boundary).

        `on_batch(indices, decoded)`: called once per batch, in batch order, as soon as that batch's hypotheses are on
        the host (`indices` = positions in `waveforms`, `decoded` = their DecodedBatch) — on a decode worker thread
        while the GPU is already busy with the following batches, so host post-processing (ids -> text) overlaps.

        Up to `max_batch` utterances run as one batch.  Longer lists are sorted by length, cut into batches of
        `max_batch` (each padded only to ITS longest utterance) and pushed through the persistent pipeline of
        `run_pipelined` — four resident batches, two decode lanes, a stager thread that fills the pinned buffers of
        batch i+2 / i+3 while the GPU works on the batches before them, H2D on the encoder stream, hypotheses copied back
        by the decode workers: no drain between batches.  Results come back in the caller's order."""
        n = len(waveforms)
        if n == 0:
            return DecodedBatch([], [], [])
        waveforms = [np.asarray(w, dtype=np.float32) for w in waveforms]
        if n <= max_batch:
            buf = self.stage(waveforms)
            self.run_device(buf)
            res = self.collect(buf)
            if on_batch is not None:
                on_batch(list(range(n)), res)
            return res
        order = sorted(range(n), key=lambda i: (len(waveforms[i]), i))
        ids, frames, enc_lens, scores = [None] * n, [None] * n, [None] * n, [None] * n
        groups = [order[i:i + max_batch] for i in range(0, n, max_batch)]
        # longest batch first: what is left after the last encoder is one decode, and the shortest batch's is the shortest
        # (a ragged list's drain shrinks from the longest batch's decode to the shortest's)
        if self.LONGEST_FIRST:
            groups.reverse()
        l_max = max(len(w) for w in waveforms)
        n_sets = min(self.POOL_SETS, len(groups))
        pool = self._pool(max_batch, l_max, n_sets)

        def fill(i, buf):
            return self.fill_host([waveforms[k] for k in groups[i]], buf)

        # `on_batch` runs on its own thread, in batch order: a decode lane that ran the caller's post-processing itself
        # would start its next batch that much later (ids -> text is ~25 ms of Python per 256 utterances)
        post_q = queue.Queue() if on_batch is not None else None
        post_err = []

        def post_worker():
            while True:
                item = post_q.get()
                if item is None:
                    return
                try:
                    if not post_err:
                        on_batch(*item)
                except Exception as e:          # surfaced on the caller's thread below
                    post_err.append(e)

        def harvest(buf):
            res = self.collect(buf, host=buf.h_out)
            group = groups[buf.step]
            for k, i in enumerate(group):
                ids[i], frames[i], enc_lens[i] = res.ids[k], res.frames[k], res.enc_lens[k]
                scores[i] = res.scores[k] if res.scores is not None else None
            if post_q is not None:
                m = len(group)
                post_q.put((group, DecodedBatch(res.ids[:m], res.frames[:m], res.enc_lens[:m],
                                                res.scores[:m] if res.scores is not None else None)))

        post = None
        if post_q is not None:
            post = threading.Thread(target=post_worker, daemon=True)
            post.start()
        # The post-processing thread allocates ~10^5 small objects per batch; a generation-2 collection triggered in the
        # middle of the pipeline holds the interpreter lock for 50-60 ms (profiles/r03n_host_timeline.txt) and every
        # decode lane behind it: the cyclic collector is paused for the duration of the call (nothing here makes cycles).
        # (process-global, so only for the duration of the call, and only the automatic trigger: gc.collect() still works for
        # a host application's other threads; `pause_gc=False` on the model opts out)
        gc_was_on = gc.isenabled() and getattr(self, "pause_gc", True)
        if gc_was_on:
            gc.disable()
        try:
            self.run_pipelined(pool, len(groups), after_decode=harvest, fill=fill, dec_streams=2 if n_sets >= 3 else 1)
        finally:
            if post is not None:
                post_q.put(None)
                post.join()
            if gc_was_on:
                gc.enable()
        if post_err:
            raise post_err[0]
        return DecodedBatch(ids, frames, enc_lens, scores if self.cfg.has_scores else None)
