"""Device side of `reazonspeech.avsr`: one AV-HuBERT encoder-decoder on one MI355X through the rs_avsr_* entry points of
librs_asr.so (include/rs_asr.h; csrc/k_avsr.hip).  PyTorch is used for device memory and streams only."""
import ctypes
import os

import numpy as np
import torch

from . import capi
from .avsr_config import AvsrConfig
from .avsr_weights import prepare_weights_avsr


class AvsrDevice:
    PRODUCTS = ("exact", "x3")

    def __init__(self, cfg: AvsrConfig, state_dict, device="cuda", products=None):
        cfg.validate()
        if not torch.cuda.is_available():
            raise RuntimeError("reazonspeech_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() is False and there is "
                               "no CPU fallback for this path")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"device {device!r}: only ROCm ('cuda[:N]') devices are supported")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        with torch.cuda.device(self.device):
            self.ctx = capi.Context(cfg, index)
            for name, t in prepare_weights_avsr(cfg, state_dict).items():
                self.ctx.set_tensor(name, t.to(self.device).contiguous())
            self.ctx.finalize()
            self.set_products(products or os.environ.get("REAZONSPEECH_AVSR_PRODUCTS", "exact"))
        self.vp = (cfg.vocab_size + 3) // 4 * 4
        self._ws = None
        self._state = None
        self._taps = None

    def set_products(self, products: str):
        """how the float32 products of the big GEMMs / convolutions are formed: "exact" = v_mfma_f32_16x16x4_f32 (an IEEE float32 chain:
        the default, what the parity statements are made with); "x3" = three bf16 matrix-core terms per product (hi / lo split, 16
        mantissa bits per operand, float32 accumulation: csrc/k_f32.hip X3) — 1.7x faster on the encoder, errors 5 - 7x the exact
        mode's and still 10x inside the stated tolerances, generate() ids equal to the reference's on every golden clip"""
        if products not in self.PRODUCTS:
            raise ValueError(f"products={products!r}: one of {self.PRODUCTS}")
        self.products = products
        self.ctx.set_option("gemm_f32_x3", 1 if products == "x3" else 0)

    # ---- encoder ---------------------------------------------------------------------------------------------------------------
    def _dev(self, x, dtype=torch.float32):
        t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        return t.to(device=self.device, dtype=dtype).contiguous()

    def encode(self, input_values, pixel_values, padding_mask, taps=None):
        """AVHubertModel.forward (modeling_avhubert.py:162-213): input_values [B][T][104], pixel_values [B][T][1][H][W] (or
        [B][T][H][W]), padding_mask [B][T] (nonzero / True = padding) -> last_hidden_state float32 [B][T][d] on the device.
        taps: list of encoder layer indices -> also returns {video, fused_ln, enc_ln, layers} (parity tests)."""
        cfg, lib, h = self.cfg, self.ctx.lib, self.ctx._h
        if input_values is None and pixel_values is None:
            raise ValueError("Either `input_values` or `pixel_values` must be passed")            # modeling_avhubert.py:181
        a = self._dev(input_values) if input_values is not None else None
        v = self._dev(pixel_values) if pixel_values is not None else None
        if v is not None and v.dim() == 5:
            v = v[:, :, 0].contiguous()
        m = self._dev(padding_mask)
        B, T = m.shape
        assert a is None or a.shape == (B, T, cfg.audio_feat_dim)
        assert v is None or v.shape == (B, T, cfg.image_size, cfg.image_size)
        d = cfg.encoder_embed_dim
        with torch.cuda.device(self.device):
            need = int(lib.rs_avsr_workspace_bytes(h, B, T))
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty((need,), dtype=torch.uint8, device=self.device)
            enc = torch.empty((B, T, d), dtype=torch.float32, device=self.device)
            out = None
            if taps is not None:
                ids = (ctypes.c_int32 * max(len(taps), 1))(*taps)
                out = {"video": torch.zeros((B, T, d), device=self.device), "fused_ln": torch.zeros((B, T, 2 * d), device=self.device),
                       "enc_ln": torch.zeros((B, T, d), device=self.device), "layers": torch.zeros((max(len(taps), 1), B, T, d), device=self.device)}
                self.ctx.check(lib.rs_avsr_encoder_set_taps(h, capi._ptr(out["video"]), capi._ptr(out["fused_ln"]), capi._ptr(out["enc_ln"]),
                                                            capi._ptr(out["layers"]), ids, len(taps)))
            stream = torch.cuda.current_stream().cuda_stream
            try:
                self.ctx.check(lib.rs_avsr_encoder_forward(h, capi._ptr(a), capi._ptr(v), capi._ptr(m), B, T, capi._ptr(enc), capi._ptr(self._ws),
                                                           self._ws.numel(), ctypes.c_void_p(stream)))
            finally:
                if taps is not None:
                    self.ctx.check(lib.rs_avsr_encoder_set_taps(h, None, None, None, None, None, 0))
        return (enc, out) if taps is not None else enc

    # ---- decoder ---------------------------------------------------------------------------------------------------------------
    class Decoding:
        """one batch of hypotheses being extended token by token (rs_avsr_decoder_begin / _step)"""

        def __init__(self, dev, enc, padding_mask, beams, max_len):
            self.dev, self.beams, self.max_len = dev, int(beams), int(max_len)
            self.B, self.T = enc.shape[:2]
            self.rows = self.B * self.beams
            self.enc = enc.contiguous()
            self.mask = dev._dev(padding_mask)
            lib, h = dev.ctx.lib, dev.ctx._h
            with torch.cuda.device(dev.device):
                need = int(lib.rs_avsr_decoder_state_bytes(h, self.B, self.T, self.beams, self.max_len))
                if dev._state is None or dev._state.numel() < need:
                    dev._state = None
                    dev._state = torch.empty((need,), dtype=torch.uint8, device=dev.device)
                self.state = dev._state
                self.logits = torch.empty((self.rows, dev.vp), dtype=torch.float32, device=dev.device)
                self.tok = torch.zeros((self.rows,), dtype=torch.int32, device=dev.device)
                self.src = torch.zeros((self.rows,), dtype=torch.int32, device=dev.device)
                dev.ctx.check(lib.rs_avsr_decoder_begin(h, capi._ptr(self.enc), self.B, self.T, self.beams, self.max_len, capi._ptr(self.state),
                                                        self.state.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

        def step(self, tokens, step, src_rows=None):
            """tokens: int array [rows] at position `step`; src_rows: int array [rows] (beam re-parenting) or None
            -> logits float32 [rows][vocab] on the device"""
            dev = self.dev
            lib, h = dev.ctx.lib, dev.ctx._h
            with torch.cuda.device(dev.device):
                self.tok.copy_(torch.as_tensor(np.asarray(tokens, dtype=np.int32)), non_blocking=False)
                src = None
                if src_rows is not None:
                    self.src.copy_(torch.as_tensor(np.asarray(src_rows, dtype=np.int32)), non_blocking=False)
                    src = self.src
                dev.ctx.check(lib.rs_avsr_decoder_step(h, capi._ptr(self.tok), capi._ptr(src), int(step), capi._ptr(self.mask), self.B, self.T, self.beams,
                                                       self.max_len, capi._ptr(self.logits), capi._ptr(self.state), self.state.numel(),
                                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
            return self.logits[:, :dev.cfg.vocab_size]

    def decoding(self, enc, padding_mask, beams, max_len):
        return AvsrDevice.Decoding(self, enc, padding_mask, beams, max_len)
