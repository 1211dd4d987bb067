"""Dimensions of the `reazonspeech.k2.asr` model: icefall's Zipformer2 transducer (encoder_embed + Zipformer2 stacks, stateless
decoder, joiner) as sherpa-onnx runs it (pkg/k2-asr/src/huggingface.py:73-83: `OfflineRecognizer.from_transducer(...,
sample_rate=16000, feature_dim=80, decoding_method="greedy_search")`).

The reference never states the architecture: it lives in three ONNX files of an unreachable Hugging Face repository
(huggingface.py:41-47).  README.rst:27-28 says "Zipformer ... 159M parameters": [UPSTREAM] that is icefall's "large" Zipformer2
recipe (egs/*/ASR/zipformer, `--num-encoder-layers 2,2,4,5,4,2 --feedforward-dim 512,768,1536,2048,1536,768 --encoder-dim
192,256,512,768,512,256 --encoder-unmasked-dim 192,192,256,320,256,192`, 148M with a 500-piece vocabulary) with the larger
Japanese vocabulary.  Every number below is a field a real checkpoint overrides (runtime/k2_weights.py derives them from the
tensor shapes of the ONNX files)."""
from dataclasses import dataclass, replace
from typing import Tuple


@dataclass(frozen=True)
class ZipformerConfig:
    # --- features: sherpa-onnx FeatureExtractorConfig over kaldi-native-fbank ([UPSTREAM] sherpa-onnx/csrc/features.h) ---
    sample_rate: int = 16000
    n_mels: int = 80            # feature_dim=80 (huggingface.py:80)
    frame_length: int = 400     # 25 ms
    frame_shift: int = 160      # 10 ms
    n_fft: int = 512            # frame_length rounded up to a power of two
    preemph: float = 0.97
    low_freq: float = 20.0
    high_freq: float = -400.0   # <= 0: offset from the Nyquist frequency (7600 Hz)
    # (dither 0, snip_edges false, remove_dc_offset, povey window, power spectrum, log with a floor of FLT_EPSILON, no CMVN)
    # --- encoder_embed: Conv2dSubsampling ([UPSTREAM] icefall zipformer/subsampling.py) ---
    embed_channels: Tuple[int, int, int] = (8, 32, 128)
    # --- Zipformer2 stacks ([UPSTREAM] icefall zipformer/zipformer.py) ---
    encoder_dim: Tuple[int, ...] = (192, 256, 512, 768, 512, 256)
    num_layers: Tuple[int, ...] = (2, 2, 4, 5, 4, 2)
    ff_dim: Tuple[int, ...] = (512, 768, 1536, 2048, 1536, 768)
    num_heads: Tuple[int, ...] = (4, 4, 4, 8, 4, 4)
    cnn_kernel: Tuple[int, ...] = (31, 31, 15, 15, 15, 31)
    downsampling: Tuple[int, ...] = (1, 2, 4, 8, 4, 2)
    query_head_dim: int = 32
    value_head_dim: int = 12
    pos_head_dim: int = 4
    pos_dim: int = 48
    output_downsampling: int = 2
    # --- stateless decoder + joiner ([UPSTREAM] icefall zipformer/decoder.py, joiner.py) ---
    vocab_size: int = 10720     # tokens.txt lines; chosen so the defaults add up to the published 159M (README.rst:27-28)
    decoder_dim: int = 512
    joiner_dim: int = 512
    context_size: int = 2
    blank_id: int = 0           # sherpa-onnx hard-codes 0
    unk_id: int = 2             # "<unk>" of tokens.txt, -1 = none: [UPSTREAM] sherpa-onnx's greedy search does not emit it
    # --- sherpa-onnx result conversion ---
    frame_shift_ms: int = 10
    subsampling_factor: int = 4

    family = "k2"
    # what the shared host runtime (runtime/model.py: AsrModel) reads off a model configuration
    espnet = False
    decoding = "greedy_batch"          # sherpa-onnx decoding_method="greedy_search" (huggingface.py:81)
    has_scores = False
    max_symbols = 1                    # one symbol per frame
    beam_size = 1

    @property
    def joint_hidden(self):
        return self.joiner_dim

    @property
    def n_layers(self):
        return sum(self.num_layers)

    def label_cap(self, tp_max: int) -> int:
        return max(tp_max, 1)

    # ---- derived -------------------------------------------------------------------------------------------------
    @property
    def n_stacks(self):
        return len(self.encoder_dim)

    @property
    def n_logits(self):
        return self.vocab_size

    @property
    def embed_freq(self):
        """frequency bins after encoder_embed's convolutions: (((80 - 1) // 2) - 1) // 2 = 19"""
        return (((self.n_mels - 1) // 2) - 1) // 2

    @property
    def out_dim(self):
        return max(self.encoder_dim)

    def fbank_frames(self, n_samples: int) -> int:
        """snip_edges = false: (n + shift / 2) // shift"""
        return (n_samples + self.frame_shift // 2) // self.frame_shift

    def embed_frames(self, n_feat: int) -> int:
        """Conv2dSubsampling: (T - 7) // 2"""
        return max((n_feat - 7) // 2, 0)

    def enc_frames(self, n_feat: int) -> int:
        """encoder output frames: ((T - 7) // 2 + 1) // 2 with the output down-sampling of 2"""
        t = self.embed_frames(n_feat)
        d = self.output_downsampling
        return (t + d - 1) // d

    def seconds_per_frame(self) -> float:
        return self.frame_shift_ms / 1000.0 * self.subsampling_factor

    def layer_ff(self, s):
        """(feed_forward1, feed_forward2, feed_forward3) hidden sizes of stack s"""
        f = self.ff_dim[s]
        return (f * 3) // 4, f, (f * 5) // 4

    def nonlin_hidden(self, s):
        return 3 * self.encoder_dim[s] // 4

    def n_params(self) -> int:
        c1, c2, c3 = self.embed_channels
        d0 = self.encoder_dim[0]
        n = (c1 * 9 + c1) + (c2 * c1 * 9 + c2) + (c3 * c2 * 9 + c3)
        n += (c3 * 49 + c3) + (3 * c3 * c3 + 3 * c3) + (c3 * 3 * c3 + c3)          # ConvNeXt
        n += self.embed_freq * c3 * d0 + d0 + (d0 + 1)                               # out, out_norm
        for s in range(self.n_stacks):
            d, h, k = self.encoder_dim[s], self.num_heads[s], self.cnn_kernel[s]
            f1, f2, f3 = self.layer_ff(s)
            hid = self.nonlin_hidden(s)
            layer = 0
            layer += d * (2 * self.query_head_dim + self.pos_head_dim) * h + (2 * self.query_head_dim + self.pos_head_dim) * h
            layer += self.pos_dim * h * self.pos_head_dim
            layer += 2 * (d * h * self.value_head_dim + h * self.value_head_dim + h * self.value_head_dim * d + d)
            for f in (f1, f2, f3):
                layer += d * f + f + f * d + d
            layer += d * 3 * hid + 3 * hid + hid * d + d
            layer += 2 * (d * 2 * d + 2 * d + d * k + d + d * d + d)
            layer += (d + 1) + d + d                                                 # norm, bypass, bypass_mid
            n += self.num_layers[s] * layer
            if self.downsampling[s] > 1:
                n += self.downsampling[s] + d                                        # downsample.bias, out_combiner.bypass_scale
        n += self.output_downsampling
        n += self.vocab_size * self.decoder_dim + self.decoder_dim * (self.decoder_dim // (self.decoder_dim // 4)) * self.context_size
        n += (self.out_dim * self.joiner_dim + self.joiner_dim) + (self.decoder_dim * self.joiner_dim + self.joiner_dim)
        n += self.joiner_dim * self.vocab_size + self.vocab_size
        return n

    def with_(self, **kw):
        return replace(self, **kw)

    def validate(self):
        n = self.n_stacks
        assert all(len(t) == n for t in (self.num_layers, self.ff_dim, self.num_heads, self.cnn_kernel, self.downsampling))
        assert self.n_fft == 512 and self.frame_length <= 512 and self.n_mels <= 128
        assert all(d % 64 == 0 for d in self.encoder_dim), "encoder_dim % 64 (GEMM K tiles)"
        assert all(f % 256 == 0 for f in self.ff_dim), "feedforward_dim % 256 (3/4 and 5/4 of it are GEMM K extents)"
        assert all(k in (7, 15, 31) for k in self.cnn_kernel), "cnn_module_kernel: the depthwise kernels are built for 7 / 15 / 31 taps"
        assert all(ds in (1, 2, 4, 8) for ds in self.downsampling) and self.downsampling[0] == 1
        assert self.query_head_dim == 32 and self.pos_head_dim == 4 and self.value_head_dim == 12, "the attention kernels are built for head dims 32 / 4 / 12"
        assert self.output_downsampling == 2 and self.context_size == 2
        assert self.decoder_dim % 128 == 0 and self.joiner_dim % 128 == 0 and self.decoder_dim % 4 == 0
        assert self.embed_channels[2] % 64 == 0 and (self.embed_freq * self.embed_channels[2]) % 64 == 0
        assert self.blank_id == 0
        return self


# the configuration README.rst:27-28 quotes ("159M parameters")
ZIPFORMER_159M = ZipformerConfig()

# a toy shape for CPU oracle runs: three stacks (one at full rate, two down-sampled), every module present
ZIPFORMER_TINY = ZipformerConfig(encoder_dim=(64, 128, 64), num_layers=(1, 2, 1), ff_dim=(256, 256, 256), num_heads=(2, 4, 2),
                                 cnn_kernel=(15, 7, 15), downsampling=(1, 2, 4), embed_channels=(8, 16, 64), vocab_size=97,
                                 decoder_dim=128, joiner_dim=128, unk_id=2)
