"""The model object `load_model()` returns: FastConformer-RNNT on one MI355X.

It owns (through torch) every device buffer — weights, activations, workspace — and drives the
three stage entry points of librs_asr.so on torch's current HIP stream.  It stands where NeMo's
`EncDecRNNTBPEModel` stands in the reference (pkg/nemo-asr/src/transcribe.py:26-28, :48-53)
and exposes the one attribute the reference's post-processing touches: `.tokenizer`
(pkg/nemo-asr/src/decode.py:41,47).
"""
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
        self.u_max = self.tp_max * cfg.max_symbols
        if cfg.decoding == "alsd":       # a hypothesis has at most one label per alignment step
            self.u_max = max(1, self.tp_max + alsd_label_budget(self.tp_max, cfg.alsd_max_target_len))
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
        self.ws_alsd = None              # beam-search scratch (grows with beam and alignment length): on first use
        self.ws = torch.empty((ctx.workspace_bytes(B, self.l_pad),), dtype=torch.uint8, device=dev)
        # the decoder of batch i overlaps the encoder of batch i+1 in the pipelined path: own scratch
        self.ws_dec = torch.empty((ctx.workspace_bytes(B, 16),), dtype=torch.uint8, device=dev)
        # second encoder scratch: the pipelined path runs the two halves of a batch on two streams
        self.ws2 = None
        # pinned staging for the host boundary
        self.h_audio = torch.zeros((B, l_max), dtype=f32).pin_memory()
        self.h_lens = torch.zeros((B,), dtype=i32).pin_memory()


class AsrModel:
    def __init__(self, cfg: ModelConfig, state_dict, tokenizer, device="cuda", pos_cap: int = DEFAULT_POS_CAP,
                 pad_seconds: float = 0.5):
        cfg.validate()
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
        self._bufs = {}
        self._ctx_dec = None
        self._ctx_dec2 = None
        self._dec2_stream = None
        self._ctx_enc2 = None
        self._enc2_stream = None
        self._streams = None
        self._streams_prio = None
        self.pos_cap = 0
        with torch.cuda.device(self.device):
            self.ctx = capi.Context(cfg, index)
            self._upload(prepare_weights(cfg, state_dict, pos_cap))

    # ------------------------------------------------------------------------------------------
    def _upload(self, tensors):
        dev = {}
        for name, t in tensors.items():
            dev[name] = t.to(self.device, non_blocking=False).contiguous()
            self.ctx.set_tensor(name, dev[name])
        self._pos_w = [dev[f"L{i}.att.pos.w"] for i in range(self.cfg.n_layers)]
        self._set_pos_tables(dev["pos.table"])

    def _contexts(self):
        return [c for c in (self.ctx, self._ctx_dec, self._ctx_dec2, self._ctx_enc2) if c is not None]

    def _set_pos_tables(self, table):
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
            for i, proj in enumerate(projs):
                c.set_tensor(f"L{i}.att.pos_proj", proj)
            c.finalize()
        self.pos_cap = (table.shape[0] + 1) // 2

    def ensure_pos_cap(self, tp: int):
        """Long-form audio: the reference hands a whole file to the model as ONE utterance
        (pkg/nemo-asr/src/transcribe.py:44-53), so T' is unbounded.  The resident position tables cover T' up to
        `pos_cap`; a longer utterance grows them (next power of two) instead of failing."""
        if tp <= self.pos_cap:
            return
        from .weights import rel_pos_table
        cap = 1 << (int(tp) - 1).bit_length()
        torch.cuda.synchronize(self.device)          # nothing may still read the tables being replaced
        with torch.cuda.device(self.device):
            table = torch.from_numpy(rel_pos_table(self.cfg, cap)).to(torch.bfloat16).to(self.device).contiguous()
            self._set_pos_tables(table)

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
        fits = [k for k in self._bufs if k[0] == B and l_max <= k[1] <= 2 * l_max]
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

    def decode(self, ctx, buf: _Buffers, ws, stream):
        """stage 3 on `stream`: the checkpoint's decoding strategy (cfg.decoding).  Greedy fills buf.ids / buf.frames
        (emission frames); ALSD fills buf.ids / buf.frames (alignment steps i = frame + labels before) / buf.scores.
        Both synchronise the stream."""
        cfg = self.cfg
        if cfg.decoding != "alsd":
            ctx.rnnt_greedy(buf.joint_enc, buf.enc_lens, buf.B, buf.tp_max, buf.u_max, buf.ids, buf.frames, buf.n_ids,
                            ws, stream)
            return
        if buf.ws_alsd is None:
            n = ctx.alsd_workspace_bytes(buf.B, cfg.beam_size, buf.tp_max, cfg.alsd_max_target_len)
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

    def run_encoder_split(self, buf: _Buffers, streams):
        """front-end + encoder of one batch as two independent half-batches on two streams: the
        memory-bound kernels of one half (LayerNorm, attention, conv, GEMM epilogues) overlap the
        MFMA-bound main loops of the other, and each GEMM's partial last round of tiles is filled
        by the other stream's work."""
        B = buf.B
        h0 = (B + 1) // 2
        if buf.ws2 is None:
            buf.ws2 = torch.empty((self.ctx.workspace_bytes(B - h0 if B > h0 else 1, buf.l_pad),), dtype=torch.uint8,
                                  device=self.device)
        parts = ((0, h0, self.ctx, buf.ws, streams[0]), (h0, B, self._ctx_enc2, buf.ws2, streams[1]))
        for lo, hi, ctx, ws, st in parts:
            if hi <= lo:
                continue
            s = st.cuda_stream
            ctx.frontend(buf.audio[lo:hi], buf.lens[lo:hi], self.pad_left, self.pad_right, buf.t_max, buf.feats[lo:hi],
                         buf.n_frames[lo:hi], ws, s)
            ctx.encoder(buf.feats[lo:hi], buf.n_frames[lo:hi], hi - lo, buf.t_max, None, buf.joint_enc[lo:hi],
                        buf.enc_lens[lo:hi], ws, s)

    def run_pipelined(self, bufs: Sequence[_Buffers], steps: int, after_decode=None, split_encoder: bool = None,
                      from_host: bool = False, enc_streams: int = 1, dec_streams: int = 1, before_encoder=None):
        """Process `steps` batches (bufs[i % len(bufs)], inputs already in HBM) as a two-stage
        pipeline: the throughput-bound front-end + encoder of batch i+1 runs on one HIP stream while
        the latency-bound greedy decode of batch i (a dependency chain of small launches that leaves
        most CUs idle) runs on a second stream, driven by a worker thread (ctypes releases the GIL;
        rs_rnnt_greedy synchronises only its own stream).  Every batch is fully decoded when this
        returns.  `after_decode(buf)` is called on the worker thread after each batch.  With
        `from_host` every batch is first copied from its pinned host buffer (H2D on the encoder
        stream) and its hypotheses are copied back to the host after decode (the PCIe-inclusive
        boundary).  `enc_streams=2` (experimental) runs the encoders of consecutive batches on two
        streams (each with its full-size launches) so that one batch's HBM-bound kernels can overlap the
        other's GEMMs; it needs four buffer sets.  `dec_streams=2` decodes consecutive batches on two streams
        (two worker threads): next to the encoder a decode launch spends most of its time waiting for compute
        units to free up, so two interleaved chains nearly double the decode rate; needs three buffer sets.
        `before_encoder(i)` is called on the caller's thread right before batch i's encoder is enqueued."""
        assert len(bufs) >= 2, "the pipeline needs two buffer sets"
        assert dec_streams in (1, 2) and (dec_streams == 1 or len(bufs) >= 3), "two decode streams need three buffer sets"
        assert enc_streams in (1, 2) and (enc_streams == 1 or len(bufs) >= 4), "two encoder streams need four buffer sets"
        with torch.cuda.device(self.device):
            if split_encoder is None:
                # measured on MI355X: splitting the encoder batch over two streams LOSES ~5 % (83.8 vs
                # 79.8 ms/step, profiles/r01h_bench_matrix_split.txt) — the GEMM grids halve and the
                # tile rounds quantise worse than the overlap wins; kept as an opt-in experiment
                split_encoder = os.environ.get("RS_SPLIT_ENCODER", "0") != "0"
            # the decode chain of ONE stream is latency-critical (its workgroups should take the first free slots: high
            # priority); with two decode lanes it has a whole extra encoder period of slack and normal priority leaves
            # the GEMM rounds alone (profiles/r02w_bench_ab.txt)
            dec_prio = int(os.environ.get("RS_DECODE_PRIORITY", "0" if dec_streams == 2 else "-1"))
            if self._ctx_dec is None:
                self._ctx_dec = self.ctx.clone()
                self._ctx_enc2 = self.ctx.clone()
                self._enc2_stream = torch.cuda.Stream(device=self.device)
            if self._streams is None or self._streams_prio != dec_prio:
                dec_cus = int(os.environ.get("RS_DECODE_CUS", "0"))
                if dec_cus > 0:
                    # decode confined to a slice of the chip (A/B knob): raw HIP stream with a CU mask
                    self._dec_raw = capi.create_stream(self.device.index, dec_cus, self.ctx.n_cus(), dec_prio)
                    dec = torch.cuda.ExternalStream(self._dec_raw, device=self.device)
                else:
                    dec = torch.cuda.Stream(device=self.device, priority=dec_prio)
                enc = self._streams[0] if self._streams is not None else torch.cuda.Stream(device=self.device)
                self._streams = (enc, dec)
                self._streams_prio = dec_prio
                self._dec2_stream = None
            enc_stream, dec_stream = self._streams
            self._decode_policy(self._ctx_dec, bufs[0].B, pipelined=True, lanes=dec_streams)
            dec_lanes = [(self._ctx_dec, dec_stream)]
            if dec_streams == 2:
                if self._ctx_dec2 is None:
                    self._ctx_dec2 = self.ctx.clone()
                if self._dec2_stream is None:
                    self._dec2_stream = torch.cuda.Stream(device=self.device, priority=dec_prio)
                self._decode_policy(self._ctx_dec2, bufs[0].B, pipelined=True, lanes=dec_streams)
                dec_lanes.append((self._ctx_dec2, self._dec2_stream))
            enc_stream.wait_stream(torch.cuda.current_stream())
            queues = [queue.Queue() for _ in dec_lanes]
            done = [threading.Event() for _ in range(steps)]
            hooked = [threading.Event() for _ in range(steps)]
            errors = []

            def worker(lane):
                ctx_d, dec_stream = dec_lanes[lane]
                jobs = queues[lane]
                with torch.cuda.device(self.device):
                    while True:
                        item = jobs.get()
                        if item is None:
                            return
                        i, buf, ev = item
                        try:
                            dec_stream.wait_event(ev)
                            # decode scratch lives past the encoder's scratch in buf.ws_dec
                            self.decode(ctx_d, buf, buf.ws_dec, dec_stream.cuda_stream)
                            if from_host:
                                with torch.cuda.stream(dec_stream):
                                    buf.h_out = (buf.n_ids.cpu(), buf.ids.cpu(), buf.frames.cpu())
                            if after_decode is not None:
                                # hooks run in batch order whatever lane finishes first (a hook may issue a collective:
                                # every rank has to issue them in the same order, from one thread at a time)
                                if i > 0:
                                    hooked[i - 1].wait()
                                after_decode(buf)
                        except Exception as e:          # surfaced on the caller's thread below
                            errors.append(e)
                        finally:
                            hooked[i].set()
                            done[i].set()

            threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(len(dec_lanes))]
            for th in threads:
                th.start()

            class _Jobs:          # batch i goes to decode lane i mod lanes
                @staticmethod
                def put(item):
                    if item is None:
                        for q in queues:
                            q.put(None)
                    else:
                        queues[item[0] % len(queues)].put(item)
            jobs = _Jobs
            nb = len(bufs)
            for i in range(steps):
                buf = bufs[i % nb]
                if i >= nb:
                    done[i - nb].wait()           # this buffer set's previous decode must be finished
                if before_encoder is not None:
                    before_encoder(i)
                if from_host:
                    with torch.cuda.stream(self._enc2_stream if (enc_streams == 2 and (i & 1)) else enc_stream):
                        buf.audio.copy_(buf.h_audio, non_blocking=True)
                        buf.lens.copy_(buf.h_lens, non_blocking=True)
                if enc_streams == 2 and (i & 1):
                    # odd batches: second context (a context is bound to one stream) on the second stream
                    es = self._enc2_stream
                    if i < 2:
                        es.wait_stream(torch.cuda.current_stream())
                    self.run_encoder(buf, es.cuda_stream, ctx=self._ctx_enc2)
                    ev = torch.cuda.Event()
                    ev.record(es)
                    jobs.put((i, buf, ev))
                    continue
                if split_encoder and buf.B >= 2:
                    self._enc2_stream.wait_stream(enc_stream)       # keep batch order across both halves
                    self.run_encoder_split(buf, (enc_stream, self._enc2_stream))
                    enc_stream.wait_stream(self._enc2_stream)
                else:
                    self.run_encoder(buf, enc_stream.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(enc_stream)
                jobs.put((i, buf, ev))
            jobs.put(None)
            for th in threads:
                th.join()
            torch.cuda.current_stream().wait_stream(enc_stream)
            torch.cuda.current_stream().wait_stream(self._enc2_stream)
            for _, ds in dec_lanes:
                torch.cuda.current_stream().wait_stream(ds)
            if errors:
                raise errors[0]

    def new_buffers(self, B, l_max) -> _Buffers:
        """an un-cached buffer set (the pipelined path keeps two batches in flight)"""
        with torch.cuda.device(self.device):
            return _Buffers(self, B, (max(int(l_max), 1) + 63) // 64 * 64)

    def stage(self, waveforms: Sequence[np.ndarray], l_max: Optional[int] = None, buf: Optional[_Buffers] = None) -> _Buffers:
        """copy host waveforms (16 kHz mono float32, un-padded) into a pinned buffer and on to HBM"""
        B = len(waveforms)
        longest = max((len(w) for w in waveforms), default=0)
        l_max = max(int(l_max or 0), longest, 1)
        l_max = (l_max + 63) // 64 * 64          # rows stay 256-B aligned
        if buf is None:
            buf = self.buffers(B, l_max)
        assert buf.B == B and buf.l_max >= longest
        # rows past each utterance's length are masked by the kernels, but keep the tail deterministic
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

    def collect(self, buf: _Buffers) -> DecodedBatch:
        n = buf.n_ids.cpu().numpy()
        ids = buf.ids.cpu().numpy()
        frames = buf.frames.cpu().numpy()
        el = buf.enc_lens.cpu().numpy()
        if self.cfg.decoding == "alsd":      # alignment step i = frame + labels emitted before
            frames = frames - np.arange(frames.shape[1], dtype=frames.dtype)[None, :]
            scores = buf.scores.cpu().numpy().tolist()
        else:
            scores = None
        return DecodedBatch([ids[b, :n[b]].tolist() for b in range(buf.B)],
                            [frames[b, :n[b]].tolist() for b in range(buf.B)], el.tolist(), scores)

    def transcribe_waveforms_sharded(self, waveforms: Sequence[np.ndarray], max_batch: int = 256) -> DecodedBatch:
        """SPMD form of `transcribe_waveforms` for one process per GPU (`torch.distributed` initialised, RCCL):
        every rank passes the same list, decodes its length-balanced shard on its own GPU and receives all
        hypotheses in the caller's order after the path's one collective (runtime/dist.py: sharded_decode).
        With a single process it is `transcribe_waveforms`."""
        from . import dist as rdist

        def run_local(indices):
            res = self.transcribe_waveforms([waveforms[i] for i in indices], max_batch=max_batch)
            return res.ids, res.frames, res.enc_lens

        ids, frames, enc_lens = rdist.sharded_decode([len(w) for w in waveforms], run_local)
        return DecodedBatch(ids, frames, enc_lens)

    def transcribe_waveforms(self, waveforms: Sequence[np.ndarray], max_batch: int = 256) -> DecodedBatch:
        """host float32 waveforms -> token ids / frames (the batched boundary).

        Up to `max_batch` utterances run as one batch.  Longer lists are sorted by length, cut into
        batches of `max_batch` (tight padding per batch) and pushed through the two-stage pipeline
        (encoder of batch i+1 || decode of batch i); results come back in the caller's order."""
        n = len(waveforms)
        if n == 0:
            return DecodedBatch([], [], [])
        if n <= max_batch:
            buf = self.stage(waveforms)
            self.run_device(buf)
            return self.collect(buf)
        order = sorted(range(n), key=lambda i: (len(waveforms[i]), i))
        ids, frames, enc_lens, scores = [None] * n, [None] * n, [None] * n, [None] * n
        groups = [order[i:i + max_batch] for i in range(0, n, max_batch)]
        l_max = max(len(w) for w in waveforms)
        pool = [self.new_buffers(max_batch, l_max), self.new_buffers(max_batch, l_max)]

        def fill(buf, group):
            # short last group: pad with empty utterances (length 0 decodes to nothing)
            waves = [waveforms[i] for i in group] + [np.zeros(0, np.float32)] * (max_batch - len(group))
            self.stage(waves, buf=buf)

        def harvest(buf, group):
            torch.cuda.current_stream().synchronize()
            res = self.collect(buf)
            for k, i in enumerate(group):
                ids[i], frames[i], enc_lens[i] = res.ids[k], res.frames[k], res.enc_lens[k]
                scores[i] = res.scores[k] if res.scores is not None else None

        # the pipeline needs inputs resident before a step starts: stage two groups ahead of use
        pending = {}

        def after(buf):
            harvest(buf, pending.pop(id(buf)))

        # process pairs of groups through run_pipelined so staging of the next pair never races the
        # encoder of the current one
        for g0 in range(0, len(groups), 2):
            pair = groups[g0:g0 + 2]
            for k, group in enumerate(pair):
                fill(pool[k], group)
                pending[id(pool[k])] = group
            torch.cuda.current_stream().synchronize()
            if len(pair) == 2:
                self.run_pipelined(pool, 2, after_decode=after)
            else:
                self.run_device(pool[0])
                after(pool[0])
        return DecodedBatch(ids, frames, enc_lens, scores if self.cfg.decoding == "alsd" else None)
