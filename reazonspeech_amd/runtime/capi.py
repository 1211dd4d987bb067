"""ctypes binding of librs_asr.so (include/rs_asr.h).

There is exactly one backend: if the shared library is missing or fails to load this module
raises — there is no CPU or PyTorch fallback (the CPU oracle under /oracle is test
infrastructure and is never imported from here).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_void_p)

_LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib", "librs_asr.so")

RS_OK = 0
RS_EINVAL = -1
RS_EOVERFLOW = -5
ERRORS = {-1: "RS_EINVAL", -2: "RS_EMISSING", -3: "RS_EWORKSPACE", -4: "RS_EHIP", -5: "RS_EOVERFLOW",
          -6: "RS_ESTATE"}

GEMM_BIAS, GEMM_RELU, GEMM_SILU, GEMM_RESIDUAL, GEMM_OUT_F32, GEMM_ROWMASK, GEMM_GLU = 1, 2, 4, 8, 16, 32, 64
GEMM_SWOOSHL, GEMM_SWOOSHR, GEMM_GELU = 128, 256, 512
GLU_HALVES, GLU_BLOCK32, GLU_APPLIED = 0, 1, 2
ALSD_SCORE_NORM, ALSD_MERGE = 1, 2
PROF_GEMM, PROF_ATTN, PROF_FRONTEND, PROF_DECODE, PROF_ELEMENTWISE, PROF_SUBSAMPLE = 1, 2, 4, 8, 16, 32

# every symbol include/rs_asr.h declares (tests/test_capi_exports.py checks the .so exports them)
EXPORTS = [
    "rs_abi_version", "rs_create", "rs_destroy", "rs_last_error", "rs_set_tensor", "rs_finalize",
    "rs_workspace_bytes", "rs_mel_frames", "rs_enc_frames", "rs_frontend_logmel", "rs_encoder_forward",
    "rs_rnnt_greedy", "rs_profile_enable", "rs_profile_read", "rs_profile_reset", "rs_gemm_bf16",
    "rs_layernorm", "rs_relpos_attention", "rs_glu_dwconv_silu", "rs_glu_dwconv_silu_layout", "rs_encoder_set_taps", "rs_set_option", "rs_stream_create", "rs_stream_destroy",
    "rs_rnnt_alsd", "rs_rnnt_alsd_workspace_bytes", "rs_rnnt_beam", "rs_rnnt_beam_workspace_bytes", "rs_host_stage_rows",
    "rs_gemm_f32", "rs_relpos_attention_f32", "rs_glu_dwconv_silu_f32", "rs_profile_read_launches", "rs_encoder_set_ctc_out",
    "rs_k2_create", "rs_k2_encoder_set_taps",
    "rs_avsr_create", "rs_avsr_workspace_bytes", "rs_avsr_encoder_forward", "rs_avsr_encoder_set_taps", "rs_avsr_decoder_state_bytes",
    "rs_avsr_decoder_begin", "rs_avsr_decoder_step",
]


class RsDims(Structure):
    """mirror of `struct rs_dims`"""
    _fields_ = [
        ("n_mels", c_int32), ("n_fft", c_int32), ("win_length", c_int32), ("hop_length", c_int32),
        ("preemph", c_float), ("log_guard", c_float), ("norm_eps", c_float),
        ("d_model", c_int32), ("n_heads", c_int32), ("ff_dim", c_int32), ("n_layers", c_int32),
        ("conv_kernel", c_int32), ("sub_channels", c_int32), ("sub_stages", c_int32), ("xscaling", c_int32),
        ("ln_eps", c_float), ("att_left", c_int32), ("att_right", c_int32), ("n_global", c_int32),
        ("n_logits", c_int32), ("blank_id", c_int32), ("pred_hidden", c_int32), ("pred_layers", c_int32),
        ("joint_hidden", c_int32), ("max_symbols", c_int32),
        ("frontend_kind", c_int32), ("sub_kind", c_int32), ("final_norm", c_int32), ("joint_act", c_int32), ("ctc_vocab", c_int32),
    ]

    @classmethod
    def from_config(cls, cfg):
        return cls(cfg.n_mels, cfg.n_fft, cfg.win_length, cfg.hop_length, cfg.preemph, cfg.log_guard,
                   cfg.norm_eps, cfg.d_model, cfg.n_heads, cfg.ff_dim, cfg.n_layers, cfg.conv_kernel,
                   cfg.sub_channels, cfg.n_sub_stages, int(cfg.xscaling), cfg.ln_eps, cfg.att_left,
                   cfg.att_right, cfg.n_global, cfg.n_logits, cfg.blank_id, cfg.pred_hidden, cfg.pred_layers,
                   cfg.joint_hidden, cfg.max_symbols,
                   *((1, 1, 1, 1, cfg.n_logits) if getattr(cfg, "espnet", False) else (0, 0, 0, 0, 0)))


class RsK2Dims(Structure):
    """mirror of `struct rs_k2_dims` (the Zipformer2 transducer of reazonspeech.k2.asr)"""
    _fields_ = [
        ("n_mels", c_int32), ("frame_length", c_int32), ("frame_shift", c_int32), ("preemph", c_float),
        ("embed_c1", c_int32), ("embed_c2", c_int32), ("embed_c3", c_int32), ("n_stacks", c_int32),
        ("encoder_dim", c_int32 * 8), ("num_layers", c_int32 * 8), ("ff_dim", c_int32 * 8), ("num_heads", c_int32 * 8),
        ("cnn_kernel", c_int32 * 8), ("downsampling", c_int32 * 8),
        ("query_head_dim", c_int32), ("value_head_dim", c_int32), ("pos_head_dim", c_int32), ("pos_dim", c_int32),
        ("vocab_size", c_int32), ("decoder_dim", c_int32), ("joiner_dim", c_int32), ("context_size", c_int32),
        ("blank_id", c_int32), ("unk_id", c_int32),
    ]

    @classmethod
    def from_config(cls, cfg):
        arr = lambda t: (c_int32 * 8)(*(list(t) + [0] * (8 - len(t))))      # noqa: E731
        return cls(cfg.n_mels, cfg.frame_length, cfg.frame_shift, cfg.preemph, *cfg.embed_channels, cfg.n_stacks,
                   arr(cfg.encoder_dim), arr(cfg.num_layers), arr(cfg.ff_dim), arr(cfg.num_heads), arr(cfg.cnn_kernel), arr(cfg.downsampling),
                   cfg.query_head_dim, cfg.value_head_dim, cfg.pos_head_dim, cfg.pos_dim, cfg.vocab_size, cfg.decoder_dim, cfg.joiner_dim,
                   cfg.context_size, cfg.blank_id, cfg.unk_id)


class RsAvsrDims(Structure):
    """mirror of `struct rs_avsr_dims` (the AV-HuBERT encoder-decoder of reazonspeech.avsr)"""
    _fields_ = [
        ("encoder_layers", c_int32), ("encoder_embed_dim", c_int32), ("encoder_ffn_dim", c_int32), ("encoder_heads", c_int32),
        ("conv_pos", c_int32), ("conv_pos_groups", c_int32), ("audio_feat_dim", c_int32), ("fuse_concat", c_int32), ("image_size", c_int32),
        ("decoder_layers", c_int32), ("decoder_embed_dim", c_int32), ("decoder_ffn_dim", c_int32), ("decoder_heads", c_int32),
        ("max_positions", c_int32), ("vocab_size", c_int32), ("layer_norm_eps", c_float),
    ]

    @classmethod
    def from_config(cls, cfg):
        return cls(cfg.encoder_layers, cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.encoder_attention_heads, cfg.conv_pos, cfg.conv_pos_groups,
                   cfg.audio_feat_dim, int(cfg.modality_fuse == "concat"), cfg.image_size, cfg.decoder_layers, cfg.decoder_embed_dim,
                   cfg.decoder_ffn_embed_dim, cfg.decoder_attention_heads, cfg.max_target_positions, cfg.vocab_size, cfg.layer_norm_eps)


class RsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{ERRORS.get(code, code)}: {message}")
        self.code = code


_lib = None


def library_path():
    return _LIB_PATH


def load():
    """Load librs_asr.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} not found — build it with `python -m reazonspeech_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    lib = ctypes.CDLL(_LIB_PATH)
    vp = c_void_p
    for name in EXPORTS:
        getattr(lib, name).restype = c_int
    lib.rs_create.argtypes = [POINTER(c_void_p), c_int, POINTER(RsDims)]
    lib.rs_destroy.argtypes = [vp]
    lib.rs_destroy.restype = None
    lib.rs_last_error.argtypes = [vp]
    lib.rs_last_error.restype = c_char_p
    lib.rs_set_tensor.argtypes = [vp, c_char_p, vp, c_size_t]
    lib.rs_finalize.argtypes = [vp]
    lib.rs_workspace_bytes.argtypes = [vp, c_int, c_int]
    lib.rs_workspace_bytes.restype = c_size_t
    lib.rs_mel_frames.argtypes = [vp, c_int]
    lib.rs_enc_frames.argtypes = [vp, c_int]
    lib.rs_frontend_logmel.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, c_size_t, vp]
    lib.rs_encoder_forward.argtypes = [vp, vp, vp, c_int, c_int, vp, vp, vp, vp, c_size_t, vp]
    lib.rs_host_stage_rows.argtypes = [vp, c_size_t, c_int, POINTER(c_void_p), POINTER(c_int32), c_int, c_int, vp]
    lib.rs_set_option.argtypes = [vp, c_char_p, c_int]
    lib.rs_stream_create.argtypes = [POINTER(c_void_p), c_int, POINTER(ctypes.c_uint32), c_int, c_int]
    lib.rs_stream_destroy.argtypes = [vp]
    lib.rs_encoder_set_taps.argtypes = [vp, vp, vp, POINTER(c_int32), c_int]
    lib.rs_rnnt_greedy.argtypes = [vp, vp, vp, c_int, c_int, c_int, vp, vp, vp, vp, c_size_t, vp]
    lib.rs_rnnt_beam_workspace_bytes.argtypes = [vp, c_int, c_int, c_int, c_int]
    lib.rs_rnnt_beam_workspace_bytes.restype = c_size_t
    lib.rs_rnnt_beam.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_int, c_int, c_int, vp, vp, vp, vp, vp, vp, c_size_t, vp]
    lib.rs_rnnt_beam.restype = c_int
    lib.rs_rnnt_alsd_workspace_bytes.argtypes = [vp, c_int, c_int, c_int, c_double, c_int]
    lib.rs_rnnt_alsd_workspace_bytes.restype = c_size_t
    lib.rs_rnnt_alsd.argtypes = [vp, vp, vp, c_int, c_int, c_int, c_double, c_int, c_int, c_int, vp, vp, vp, vp, vp,
                                 c_size_t, vp]
    lib.rs_profile_enable.argtypes = [vp, c_int]
    lib.rs_profile_reset.argtypes = [vp]
    lib.rs_profile_read.argtypes = [vp, c_int, POINTER(c_double), POINTER(c_int64), POINTER(c_double),
                                    POINTER(c_double)]
    lib.rs_gemm_bf16.argtypes = [vp, vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, vp,
                                 c_float, vp, vp, c_int, c_int, vp]
    lib.rs_layernorm.argtypes = [vp, vp, vp, vp, c_int, c_int, c_float, vp, vp, vp]
    lib.rs_relpos_attention.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp]
    lib.rs_glu_dwconv_silu.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp]
    lib.rs_glu_dwconv_silu_layout.argtypes = [vp, vp, c_int, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp]
    lib.rs_profile_read_launches.argtypes = [vp, c_int, vp, vp, vp, c_int, POINTER(c_int)]
    lib.rs_gemm_f32.argtypes = [vp, vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, vp, c_float, vp, vp, c_int,
                                c_int, vp]
    lib.rs_relpos_attention_f32.argtypes = [vp, vp, vp, vp, vp, vp, c_int, c_int, vp, vp]
    lib.rs_glu_dwconv_silu_f32.argtypes = [vp, vp, vp, vp, vp, c_int, c_int, c_int, c_int, vp, vp]
    lib.rs_encoder_set_ctc_out.argtypes = [vp, vp, vp]
    lib.rs_k2_create.argtypes = [POINTER(c_void_p), c_int, POINTER(RsK2Dims)]
    lib.rs_k2_encoder_set_taps.argtypes = [vp, vp, vp]
    lib.rs_avsr_create.argtypes = [POINTER(c_void_p), c_int, POINTER(RsAvsrDims)]
    lib.rs_avsr_workspace_bytes.argtypes = [vp, c_int, c_int]
    lib.rs_avsr_workspace_bytes.restype = c_size_t
    lib.rs_avsr_encoder_forward.argtypes = [vp, vp, vp, vp, c_int, c_int, vp, vp, c_size_t, vp]
    lib.rs_avsr_encoder_set_taps.argtypes = [vp, vp, vp, vp, vp, POINTER(c_int32), c_int]
    lib.rs_avsr_decoder_state_bytes.argtypes = [vp, c_int, c_int, c_int, c_int]
    lib.rs_avsr_decoder_state_bytes.restype = c_size_t
    lib.rs_avsr_decoder_begin.argtypes = [vp, vp, c_int, c_int, c_int, c_int, vp, c_size_t, vp]
    lib.rs_avsr_decoder_step.argtypes = [vp, vp, vp, c_int, vp, c_int, c_int, c_int, c_int, vp, vp, c_size_t, vp]
    if lib.rs_abi_version() != 6:
        raise ImportError("librs_asr.so ABI version mismatch")
    _lib = lib
    return lib


def _ptr(t):
    """pointer of a torch tensor (or NULL)"""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def host_stage_rows(dst, width, waveforms, dst_lens):
    """gather host float32 utterances into the pinned staging matrix `dst` (torch float32 [rows][pitch]) with ONE call that
    runs without the interpreter lock (rs_host_stage_rows); `dst_lens` = torch int32 [rows]"""
    import numpy as np
    lib = load()
    n = len(waveforms)
    rows = [np.ascontiguousarray(w, dtype=np.float32) for w in waveforms]      # (no copy for float32 C-contiguous input)
    ptrs = (c_void_p * max(n, 1))(*[r.ctypes.data for r in rows])
    lens = (c_int32 * max(n, 1))(*[len(r) for r in rows])
    rc = lib.rs_host_stage_rows(c_void_p(dst.data_ptr()), dst.stride(0), int(width), ptrs, lens, n, dst.shape[0],
                                c_void_p(dst_lens.data_ptr()))
    if rc != RS_OK:
        raise RsError(rc, "rs_host_stage_rows: bad arguments")


def create_stream(device_index, n_cus=0, total_cus=256, priority=0):
    """raw HIP stream handle (int) restricted to `n_cus` compute units spread evenly over the XCDs (0 = all CUs,
    plain stream of the given priority).  The bit pattern keeps whole groups of 8 consecutive bits together and
    spaces the groups, so both a round-robin and a blocked bit -> XCD numbering give every XCD the same share."""
    lib = load()
    out = c_void_p()
    if n_cus and n_cus < total_cus:
        words = (total_cus + 31) // 32
        groups, want = total_cus // 8, max(1, n_cus // 8)
        chosen = {int(round(k * groups / want)) for k in range(want)}
        mask = [0] * words
        for g in chosen:
            for b in range(8):
                i = g * 8 + b
                mask[i // 32] |= 1 << (i % 32)
        arr = (ctypes.c_uint32 * words)(*mask)
        rc = lib.rs_stream_create(byref(out), int(device_index), arr, words, int(priority))
    else:
        rc = lib.rs_stream_create(byref(out), int(device_index), None, 0, int(priority))
    if rc != RS_OK:
        raise RsError(rc, "rs_stream_create failed")
    return out.value


class Context:
    """Owns one rs_ctx.  Not thread safe; one context per stream."""

    def __init__(self, cfg, device_index=0):
        self.lib = load()
        self.cfg = cfg
        self.device_index = int(device_index)
        self._h = c_void_p()
        self._keep = {}
        if getattr(cfg, "family", "nemo") == "avsr":         # the AV-HuBERT encoder-decoder of reazonspeech.avsr
            dims = RsAvsrDims.from_config(cfg)
            rc = self.lib.rs_avsr_create(byref(self._h), int(device_index), byref(dims))
        elif getattr(cfg, "family", "nemo") == "k2":         # the Zipformer2 transducer of reazonspeech.k2.asr
            dims = RsK2Dims.from_config(cfg)
            rc = self.lib.rs_k2_create(byref(self._h), int(device_index), byref(dims))
        else:
            dims = RsDims.from_config(cfg)
            rc = self.lib.rs_create(byref(self._h), int(device_index), byref(dims))
        if rc != RS_OK:
            msg = self.lib.rs_last_error(self._h).decode() if self._h else "rs_create failed"
            if self._h:
                self.lib.rs_destroy(self._h)
                self._h = c_void_p()
            raise RsError(rc, msg)

    def clone(self):
        """a second context over the SAME device weights (contexts are per stream: the pipelined
        path runs the encoder and the decoder of consecutive batches on two streams)"""
        other = Context(self.cfg, self.device_index)
        for name, t in self._keep.items():
            other.set_tensor(name, t)
        other.finalize()
        return other

    def close(self):
        if getattr(self, "_h", None):
            self.lib.rs_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != RS_OK:
            raise RsError(rc, self.lib.rs_last_error(self._h).decode())

    def set_tensor(self, name, tensor):
        assert tensor.is_contiguous()
        self._keep[name] = tensor   # the library does not own memory: keep it alive here
        self.check(self.lib.rs_set_tensor(self._h, name.encode(), _ptr(tensor),
                                          tensor.numel() * tensor.element_size()))

    def finalize(self):
        self.check(self.lib.rs_finalize(self._h))

    def workspace_bytes(self, B, max_samples):
        return int(self.lib.rs_workspace_bytes(self._h, int(B), int(max_samples)))

    def mel_frames(self, n):
        return int(self.lib.rs_mel_frames(self._h, int(n)))

    def enc_frames(self, n):
        return int(self.lib.rs_enc_frames(self._h, int(n)))

    # ---- stages ----
    def frontend(self, audio, lens, pad_left, pad_right, t_max, feats, n_frames, ws, stream):
        self.check(self.lib.rs_frontend_logmel(self._h, _ptr(audio), _ptr(lens), audio.shape[0], audio.stride(0),
                                               pad_left, pad_right, t_max, _ptr(feats), _ptr(n_frames), _ptr(ws),
                                               ws.numel() * ws.element_size(), c_void_p(stream)))

    def encoder(self, feats, n_frames, B, t_max, enc_out, joint_enc, enc_lens, ws, stream):
        self.check(self.lib.rs_encoder_forward(self._h, _ptr(feats), _ptr(n_frames), B, t_max, _ptr(enc_out),
                                               _ptr(joint_enc), _ptr(enc_lens), _ptr(ws),
                                               ws.numel() * ws.element_size(), c_void_p(stream)))

    def n_cus(self):
        import torch
        return int(torch.cuda.get_device_properties(self.device_index).multi_processor_count)

    def set_option(self, key, value):
        self.check(self.lib.rs_set_option(self._h, key.encode(), int(value)))

    def set_taps(self, sub_out=None, layer_out=None, layer_ids=()):
        """parity taps of rs_encoder_forward (tests): f32 [B*tp_max][d] after subsampling, and
        [len(layer_ids)][B*tp_max][d] after the listed conformer layers; no arguments = off"""
        ids = (c_int32 * len(layer_ids))(*layer_ids)
        self._taps = (sub_out, layer_out)          # keep the buffers alive while registered
        self.check(self.lib.rs_encoder_set_taps(self._h, _ptr(sub_out), _ptr(layer_out), ids, len(layer_ids)))

    def set_k2_taps(self, embed_out=None, stack_out=None):
        """parity taps of a Zipformer context (tests): encoder_embed's output f32 [B*T3][encoder_dim[0]] and the stacks' outputs
        f32 [B*T3][encoder_dim[s]] one after the other; no arguments = off"""
        self._k2_taps = (embed_out, stack_out)
        self.check(self.lib.rs_k2_encoder_set_taps(self._h, _ptr(embed_out), _ptr(stack_out)))

    def set_ctc_out(self, probs=None, blank_prob=None):
        """CTC posteriors of the next encoder calls (ESPnet family): f32 [B*tp_max][vocab] and / or the blank column
        f32 [B*tp_max]; no arguments = off"""
        self._ctc = (probs, blank_prob)            # keep the buffers alive while registered
        self.check(self.lib.rs_encoder_set_ctc_out(self._h, _ptr(probs), _ptr(blank_prob)))

    def rnnt_greedy(self, joint_enc, enc_lens, B, tp_max, u_max, ids, frames, n_ids, ws, stream):
        self.check(self.lib.rs_rnnt_greedy(self._h, _ptr(joint_enc), _ptr(enc_lens), B, tp_max, u_max, _ptr(ids),
                                           _ptr(frames), _ptr(n_ids), _ptr(ws), ws.numel() * ws.element_size(),
                                           c_void_p(stream)))

    @staticmethod
    def _alsd_budget(max_target_len):
        """(ratio, abs) of the C ABI: a float is a multiple of the frame count, an int an absolute label budget"""
        if isinstance(max_target_len, float):
            return float(max_target_len), -1
        return 0.0, int(max_target_len)

    def alsd_workspace_bytes(self, B, beam, tp_max, max_target_len):
        ratio, abs_len = self._alsd_budget(max_target_len)
        n = self.lib.rs_rnnt_alsd_workspace_bytes(self._h, B, beam, tp_max, ratio, abs_len)
        if n == 0:
            raise RuntimeError("rs_rnnt_alsd_workspace_bytes: invalid arguments")
        return n

    def rnnt_alsd(self, joint_enc, enc_lens, B, tp_max, beam, max_target_len, score_norm, merge, ids, steps, n_ids,
                  scores, ws, stream):
        """ids / steps int32 [B][out_cap], n_ids int32 [B], scores float32 [B]"""
        ratio, abs_len = self._alsd_budget(max_target_len)
        flags = (ALSD_SCORE_NORM if score_norm else 0) | (ALSD_MERGE if merge else 0)
        self.check(self.lib.rs_rnnt_alsd(self._h, _ptr(joint_enc), _ptr(enc_lens), B, tp_max, beam, ratio, abs_len, flags,
                                         ids.shape[1], _ptr(ids), _ptr(steps), _ptr(n_ids), _ptr(scores), _ptr(ws),
                                         ws.numel() * ws.element_size(), c_void_p(stream)))

    def beam_workspace_bytes(self, B, beam, tp_max, max_pops=0):
        n = self.lib.rs_rnnt_beam_workspace_bytes(self._h, B, beam, tp_max, max_pops)
        if n == 0:
            raise RuntimeError("rs_rnnt_beam_workspace_bytes: invalid arguments")
        return n

    def rnnt_beam(self, joint_enc, enc_lens, B, tp_max, beam, score_norm, max_pops, ids, n_ids, scores, pops, ws, stream, frames=None):
        """ESPnet's default transducer beam search: ids (and frames, optional) int32 [B][out_cap], n_ids / pops int32 [B],
        scores float32 [B]"""
        assert frames is None or frames.shape == ids.shape
        self.check(self.lib.rs_rnnt_beam(self._h, _ptr(joint_enc), _ptr(enc_lens), B, tp_max, beam, 1 if score_norm else 0,
                                         int(max_pops), ids.shape[1], _ptr(ids), _ptr(frames) if frames is not None else None, _ptr(n_ids), _ptr(scores), _ptr(pops), _ptr(ws),
                                         ws.numel() * ws.element_size(), c_void_p(stream)))

    # ---- profiling ----
    def profile_enable(self, mask):
        self.check(self.lib.rs_profile_enable(self._h, int(mask)))

    def profile_reset(self):
        self.check(self.lib.rs_profile_reset(self._h))

    def profile_read(self, klass):
        ms, n, fl, by = c_double(), c_int64(), c_double(), c_double()
        self.check(self.lib.rs_profile_read(self._h, int(klass), byref(ms), byref(n), byref(fl), byref(by)))
        return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)

    def profile_launches(self, klass):
        """per-launch records of a profiled class since the last reset: list of (M, N, K, flags, flops, ms)"""
        import numpy as np
        n = c_int(0)
        self.check(self.lib.rs_profile_read_launches(self._h, int(klass), None, None, None, 0, byref(n)))
        cap = max(n.value, 1)
        shapes, fl, ms = np.zeros((cap, 4), np.int32), np.zeros((cap,), np.float64), np.zeros((cap,), np.float32)
        self.check(self.lib.rs_profile_read_launches(self._h, int(klass), c_void_p(shapes.ctypes.data), c_void_p(fl.ctypes.data),
                                                     c_void_p(ms.ctypes.data), cap, byref(n)))
        k = min(n.value, cap)
        return [(int(shapes[i, 0]), int(shapes[i, 1]), int(shapes[i, 2]), int(shapes[i, 3]), float(fl[i]), float(ms[i])) for i in range(k)]

    # ---- single operators (used by the parity tests) ----
    def gemm(self, A, W, out, flags=0, bias=None, alpha=1.0, residual=None, mask_lens=None, mask_rows=0,
             mask_steps=0, stream=0):
        M, K = A.shape
        N = W.shape[0]
        self.check(self.lib.rs_gemm_bf16(self._h, _ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(out),
                                         out.stride(0), M, N, K, flags, _ptr(bias), float(alpha), _ptr(residual),
                                         _ptr(mask_lens), mask_rows, mask_steps, c_void_p(stream)))

    def gemm_f32(self, A, W, out, flags=0, bias=None, alpha=1.0, residual=None, mask_lens=None, mask_rows=0, mask_steps=0,
                 stream=0):
        """the float32 parity mode's GEMM: A f32 [M][K], W f32 [N][K] -> out f32 [M][N]"""
        M, K = A.shape
        N = W.shape[0]
        self.check(self.lib.rs_gemm_f32(self._h, _ptr(A), A.stride(0), _ptr(W), W.stride(0), _ptr(out), out.stride(0), M, N, K,
                                        flags, _ptr(bias), float(alpha), _ptr(residual), _ptr(mask_lens), mask_rows,
                                        mask_steps, c_void_p(stream)))

    def attention_f32(self, qkv, pos, bias_u, bias_v, lens, B, T, out, stream=0):
        self.check(self.lib.rs_relpos_attention_f32(self._h, _ptr(qkv), _ptr(pos), _ptr(bias_u), _ptr(bias_v), _ptr(lens), B, T,
                                                    _ptr(out), c_void_p(stream)))

    def glu_dwconv_f32(self, x, w, b, lens, B, T, d, k, out, stream=0):
        self.check(self.lib.rs_glu_dwconv_silu_f32(self._h, _ptr(x), _ptr(w), _ptr(b), _ptr(lens), B, T, d, k, _ptr(out),
                                                   c_void_p(stream)))

    def layernorm(self, x, gamma, beta, eps, out_bf16=None, out_f32=None, stream=0):
        M, d = x.shape
        self.check(self.lib.rs_layernorm(self._h, _ptr(x), _ptr(gamma), _ptr(beta), M, d, float(eps),
                                         _ptr(out_bf16), _ptr(out_f32), c_void_p(stream)))

    def attention(self, qkv, pos, bias_u, bias_v, lens, B, T, out, stream=0):
        self.check(self.lib.rs_relpos_attention(self._h, _ptr(qkv), _ptr(pos), _ptr(bias_u), _ptr(bias_v),
                                                _ptr(lens), B, T, _ptr(out), c_void_p(stream)))

    def glu_dwconv(self, x, w, b, lens, B, T, d, k, out, stream=0, layout=None):
        if layout is None:
            self.check(self.lib.rs_glu_dwconv_silu(self._h, _ptr(x), _ptr(w), _ptr(b), _ptr(lens), B, T, d, k,
                                                   _ptr(out), c_void_p(stream)))
        else:
            self.check(self.lib.rs_glu_dwconv_silu_layout(self._h, _ptr(x), int(layout), _ptr(w), _ptr(b), _ptr(lens),
                                                          B, T, d, k, _ptr(out), c_void_p(stream)))

