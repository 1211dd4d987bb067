"""The object `reazonspeech.espnet.asr.load_model()` returns here: the ESPnet2 Conformer-Transducer on one MI355X.

It stands where `espnet2.bin.asr_inference.Speech2Text` stands in the reference (pkg/espnet-asr/src/transcribe.py:26-32)
and answers the calls the reference makes on it:
    model(padded_samples)[0][0]            -> text of the best hypothesis            (transcribe.py:69)
    model.asr_model.blank_id / .token_list                                           (ctc.py:36,67)
    model.asr_model.encode(speech, length)[0], model.asr_model.ctc.softmax(enc)      (ctc.py:24-26)
    model.dtype, model.device                                                        (ctc.py:19-21)
so that the reference's own ctc.py runs against it unmodified (tests/test_espnet_host.py does exactly that); the package's
own code uses the direct forms `recognize` / `ctc_posteriors`.  Decoding on the device follows `beam_size` the way
[UPSTREAM] BeamSearchTransducer does: beam_size <= 1 is greedy_search (one symbol per frame), anything larger the "default"
beam search with score normalisation (k_rnnt_beam.hip) — the reference's setting is Speech2Text's default, beam_size 20
(transcribe.py:27-31 overrides only lm_weight)."""
import numpy as np
import torch

from ...runtime.model import AsrModel

PADDING = (16000, 8000)          # transcribe.py:10


class _Ctc:
    def __init__(self, owner):
        self._owner = owner

    def softmax(self, enc):
        """CTC.softmax(enc) of an `encode` result: the posteriors the same encoder pass produced"""
        if enc is not self._owner._last_enc:
            raise RuntimeError("ctc.softmax expects the tensor the last asr_model.encode call returned")
        return self._owner._last_ctc


class _AsrModelView:
    """the attributes of `Speech2Text.asr_model` the reference touches"""

    def __init__(self, owner):
        self._owner = owner
        self.blank_id = owner.cfg.blank_id
        self.token_list = owner.token_list
        self.ctc = _Ctc(owner)

    def encode(self, speech, length):
        o = self._owner
        wav = speech.detach().float().cpu().numpy().reshape(-1)[:int(length.reshape(-1)[0])]
        enc, probs = o._encode_with_ctc(wav)
        o._last_enc, o._last_ctc = enc, probs
        return enc, torch.tensor([enc.shape[1]], dtype=torch.long)


class EspnetModel:
    def __init__(self, cfg, state_dict, token_list, device="cuda", beam_size=1, max_pops=0, precision="bf16"):
        assert cfg.espnet and len(token_list) == cfg.vocab_size
        if beam_size is not None and int(beam_size) > 1:
            cfg = cfg.with_(decoding="beam", beam_size=int(beam_size), beam_score_norm=True, beam_max_pops=int(max_pops))
        self.beam_size = cfg.beam_size if cfg.decoding == "beam" else 1
        self.cfg = cfg
        self.token_list = list(token_list)
        self.am = AsrModel(cfg, state_dict, None, device=device, pad_seconds=0.0, precision=precision)
        self.device = self.am.device
        self.dtype = "float32"
        self._last_enc = self._last_ctc = None
        self.asr_model = _AsrModelView(self)

    # ---- the reference's call forms -------------------------------------------------------------------------------
    def __call__(self, speech):
        """Speech2Text.__call__: n-best list of (text, tokens, token ids, hypothesis); here nbest = 1 (upstream's default)"""
        wav = np.asarray(speech.detach().cpu().numpy() if isinstance(speech, torch.Tensor) else speech, dtype=np.float32).reshape(-1)
        ids = self._search([wav]).ids[0]
        tokens = [self.token_list[i] for i in ids]
        return [(self.tokens2text(tokens), tokens, ids, None)]

    # ---- direct forms -----------------------------------------------------------------------------------------------
    @staticmethod
    def tokens2text(tokens):
        """[UPSTREAM] espnet2 CharTokenizer.tokens2text (token_type: char): '<space>' stands for ' ', the rest joins as is"""
        return "".join(" " if t == "<space>" else t for t in tokens)

    def ids_to_text(self, ids):
        return self.tokens2text([self.token_list[i] for i in ids])

    def recognize_batch(self, waves):
        """padded like the reference pads each window (np.pad(samples, PADDING), transcribe.py:69) -> [text]"""
        res = self._search([np.pad(np.asarray(w, np.float32), PADDING, mode="constant") for w in waves])
        return [self.ids_to_text(ids) for ids in res.ids]

    def _search(self, waves, max_batch=256):
        """the transducer search over a batch of (padded) windows.  Upstream's default beam search has no bound on the
        prediction-network evaluations a frame may take; the device search has one (`max_pops`, which sizes its workspace)
        and reports RS_EOVERFLOW instead of truncating.  The front end and the encoder run ONCE per batch; a decode that hits
        the bound is retried on the same joint projection with 4x and 16x the bound, then done greedily with a warning — one
        pathological window must not abort a whole file — and the result says so (`DecodedBatch.degraded`).  The overrides are
        arguments of the decode call: nothing is written into the shared model configuration."""
        from ...runtime.capi import RsError, RS_EOVERFLOW
        from ...runtime.model import DecodedBatch
        am = self.am
        if am.cfg.decoding != "beam":
            return am.transcribe_waveforms(waves)
        bound = am.cfg.beam_max_pops or 16 * am.cfg.beam_size
        out = DecodedBatch([], [], [], [], [])
        for lo in range(0, len(waves), max_batch):
            buf = am.stage([np.asarray(w, np.float32) for w in waves[lo:lo + max_batch]])
            with torch.cuda.device(am.device):
                stream = torch.cuda.current_stream().cuda_stream
                am.run_encoder(buf, stream)
                used = None
                for factor in (1, 4, 16):
                    try:
                        am.decode(am.ctx, buf, buf.ws, stream, max_pops=bound * factor)
                        used = "beam"
                        break
                    except RsError as e:
                        if e.code != RS_EOVERFLOW:
                            raise
                if used is None:
                    import warnings
                    warnings.warn(f"beam search: a frame needed more than {16 * bound} prediction-network evaluations; this batch of windows "
                                  "is decoded with the greedy search instead", RuntimeWarning, stacklevel=3)
                    am.decode(am.ctx, buf, buf.ws, stream, decoding="greedy_batch")
                    used = "greedy_batch"
                res = am.collect(buf, decoding=used)
            out.ids += res.ids; out.frames += res.frames; out.enc_lens += res.enc_lens
            out.scores += res.scores if res.scores is not None else [float("nan")] * buf.B
            out.degraded += [used != "beam"] * buf.B
        return out

    def recognize(self, samples):
        return self.recognize_batch([samples])[0]

    def _encode_with_ctc(self, wav):
        am = self.am
        buf = am.stage([np.asarray(wav, np.float32)])
        M = buf.B * buf.tp_max
        vp = (self.cfg.n_logits + 3) // 4 * 4      # row pitch of the posteriors (include/rs_asr.h: rs_encoder_set_ctc_out)
        probs = torch.empty((M, vp), dtype=torch.float32, device=am.device)
        enc = torch.empty((buf.B, buf.tp_max, self.cfg.d_model), dtype=torch.float32, device=am.device)
        with torch.cuda.device(am.device):
            stream = torch.cuda.current_stream().cuda_stream
            am.ctx.set_ctc_out(probs, None)
            try:
                am.ctx.frontend(buf.audio, buf.lens, 0, 0, buf.t_max, buf.feats, buf.n_frames, buf.ws, stream)
                am.ctx.encoder(buf.feats, buf.n_frames, buf.B, buf.t_max, enc, buf.joint_enc, buf.enc_lens, buf.ws, stream)
            finally:
                am.ctx.set_ctc_out(None, None)
            torch.cuda.synchronize(am.device)
        n = int(buf.enc_lens.cpu()[0])
        return enc[:, :n].cpu(), probs.view(buf.B, buf.tp_max, vp)[:, :n, :self.cfg.n_logits].contiguous().cpu()

    def ctc_posteriors(self, samples):
        """softmax(ctc_lo(encoder(samples))) as float32 numpy [T'][vocab] (ctc.py:12-27 — no padding)"""
        return self._encode_with_ctc(samples)[1][0].numpy()


def synthetic_token_list(vocab_size: int, seed: int = 0):
    """An ESPnet-style character token list for synthetic-weight runs: '<blank>', '<unk>', punctuation the segmenter looks
    for (ctc.py:6-8), kana / kanji, '<sos/eos>' last ([UPSTREAM] ESPnet2 token_list layout, token_type: char)."""
    fixed = ["<blank>", "<unk>", "。", "、", "?", "!", ","]
    pool = [chr(c) for c in range(0x3041, 0x3097)] + [chr(c) for c in range(0x30A1, 0x30FB)] + [chr(c) for c in range(0x4E00, 0x4E00 + 8192)]
    rng = np.random.default_rng(seed)
    rng.shuffle(pool)
    body = pool[:max(0, vocab_size - len(fixed) - 1)]
    toks = (fixed + body)[:vocab_size - 1] + ["<sos/eos>"]
    assert len(toks) == vocab_size and len(set(toks)) == vocab_size
    return toks
