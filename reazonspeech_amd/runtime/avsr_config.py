"""Architecture of the AV-HuBERT encoder-decoder behind `reazonspeech.avsr` (pkg/avsr/src/avhubert/configuration_avhubert.py:4-151,
configuration_resnet.py:4-17).  Field names and defaults are the reference's `AVHubertConfig`; only what inference reads is kept
(dropouts, layerdrop, CTC loss settings and the unused wav2vec-style conv_* lists are training / inert fields)."""
from dataclasses import dataclass, replace
from typing import Optional


@dataclass(frozen=True)
class AvsrConfig:
    family: str = "avsr"
    # encoder (configuration_avhubert.py:9-13, 21-26)
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    activation_fn: str = "gelu"
    conv_pos: int = 128
    conv_pos_groups: int = 16
    resnet_relu_type: str = "prelu"
    audio_feat_dim: int = 104
    modality_fuse: str = "concat"
    do_stable_layer_norm: bool = False
    layer_norm_eps: float = 1e-5          # HubertConfig's default (encoder_config / decoder_config do not override it)
    # video front-end (configuration_resnet.py: frontend_nout 64, backend_out 512; modeling_resnet.py:140-178)
    frontend_nout: int = 64
    backend_out: int = 512
    image_size: int = 88                  # feature_extraction_avhubert.py:24 image_crop_size
    # decoder (configuration_avhubert.py:27-40)
    decoder_embed_dim: int = 768
    decoder_ffn_embed_dim: int = 3072
    decoder_layers: int = 6
    decoder_attention_heads: int = 4
    decoder_learned_pos: bool = False
    max_target_positions: int = 2048
    share_decoder_input_output_embed: bool = False
    vocab_size: Optional[int] = 1000
    pad_token_id: int = 1
    bos_token_id: int = 0
    eos_token_id: int = 2
    decoder_start_token_id: int = 2

    def with_(self, **kw):
        return replace(self, **kw)

    @property
    def fused_dim(self):
        return 2 * self.encoder_embed_dim if self.modality_fuse == "concat" else self.encoder_embed_dim

    def validate(self):
        d, dd = self.encoder_embed_dim, self.decoder_embed_dim
        assert self.activation_fn == "gelu" and self.resnet_relu_type in ("prelu", "relu")
        assert self.modality_fuse in ("concat", "add")
        assert not self.do_stable_layer_norm and not self.decoder_learned_pos, "the post-LayerNorm, sinusoidal-position variant is built"
        assert d % self.encoder_attention_heads == 0 and dd % self.decoder_attention_heads == 0
        assert d % 32 == 0 and dd % 32 == 0 and self.encoder_ffn_embed_dim % 32 == 0 and self.decoder_ffn_embed_dim % 32 == 0
        assert d % self.conv_pos_groups == 0 and self.conv_pos % 2 == 0
        assert self.frontend_nout == 64 and self.backend_out == 512 and self.image_size % 8 == 0
        assert self.vocab_size and self.vocab_size >= 4 and d == dd, "cross-attention reads encoder states of the decoder's width"
        return self

    def n_params(self):
        from .avsr_weights import expected_shapes_avsr
        n = 0
        for k, s in expected_shapes_avsr(self).items():
            if k.endswith(("num_batches_tracked", "position_embeddings", "running_mean", "running_var")):      # buffers
                continue
            c = 1
            for x in s:
                c *= x
            n += c
        return n


AVSR_BASE = AvsrConfig()                 # the reference's defaults: 12 x 768 encoder, 6 x 768 decoder, ResNet-18 video trunk (161M with 1000 tokens)
AVSR_TINY = AvsrConfig(encoder_layers=2, encoder_embed_dim=128, encoder_ffn_embed_dim=256, encoder_attention_heads=4, conv_pos=16, conv_pos_groups=4,
                       decoder_embed_dim=128, decoder_ffn_embed_dim=256, decoder_layers=2, decoder_attention_heads=2, vocab_size=61, max_target_positions=64)
