"""Loader of the REFERENCE's own AV-HuBERT model code (pkg/avsr/src/avhubert/{configuration_avhubert,configuration_resnet,
modeling_resnet,decoder,modeling_avhubert}.py), imported UNCHANGED from /root/reference.  TEST INFRASTRUCTURE (oracle/__init__.py),
usable in the build container only: /root/reference does not exist on the GPU box, nothing under tests/ -m gpu, smoke() or
bench.py may call this.  Its one consumer is tests/golden/make_avsr_golden.py, which runs the reference on this repo's seeded
synthetic weights and inputs and commits the outputs as fixtures (tests/golden/avsr_ref_*.npz); tests/test_oracle_avsr.py also
uses it — when the reference is present — to compare oracle/avsr.py with the reference directly on fresh inputs.

The package's __init__ imports the feature extractor, which needs cv2 / mediapipe / librosa / python_speech_features (absent
here); the model modules do not, so they are imported under a stub parent package."""
import os
import sys
import types

REF = "/root/reference/pkg/avsr/src/avhubert"


def available() -> bool:
    return os.path.isdir(REF)


def modules():
    """-> (configuration_avhubert, modeling_avhubert) of the reference"""
    if not available():
        raise RuntimeError("the reference tree is not present (build container only)")
    if "avhubert" not in sys.modules:
        pkg = types.ModuleType("avhubert")
        pkg.__path__ = [REF]
        sys.modules["avhubert"] = pkg
    from avhubert import configuration_avhubert, modeling_avhubert
    return configuration_avhubert, modeling_avhubert


def build(cfg, state_dict):
    """the reference's AVHubertForConditionalGeneration at this repo's AvsrConfig, loaded STRICTLY with `state_dict`"""
    import torch
    conf, model = modules()
    rc = conf.AVHubertConfig(
        encoder_layers=cfg.encoder_layers, encoder_embed_dim=cfg.encoder_embed_dim, encoder_ffn_embed_dim=cfg.encoder_ffn_embed_dim,
        encoder_attention_heads=cfg.encoder_attention_heads, activation_fn=cfg.activation_fn, conv_pos=cfg.conv_pos,
        conv_pos_groups=cfg.conv_pos_groups, resnet_relu_type=cfg.resnet_relu_type, audio_feat_dim=cfg.audio_feat_dim,
        modality_fuse=cfg.modality_fuse, decoder_embed_dim=cfg.decoder_embed_dim, decoder_ffn_embed_dim=cfg.decoder_ffn_embed_dim,
        decoder_layers=cfg.decoder_layers, decoder_attention_heads=cfg.decoder_attention_heads, decoder_learned_pos=cfg.decoder_learned_pos,
        max_target_positions=cfg.max_target_positions, share_decoder_input_output_embed=cfg.share_decoder_input_output_embed,
        do_stable_layer_norm=cfg.do_stable_layer_norm, vocab_size=cfg.vocab_size, pad_token_id=cfg.pad_token_id, bos_token_id=cfg.bos_token_id,
        eos_token_id=cfg.eos_token_id, decoder_start_token_id=cfg.decoder_start_token_id)
    with torch.no_grad():
        m = model.AVHubertForConditionalGeneration(rc).eval()
        missing, unexpected = m.load_state_dict(state_dict, strict=True)
    assert not missing and not unexpected
    # The reference pins transformers <= 4.53.3 (pkg/avsr/pyproject.toml), whose HubertEncoder.forward turns the padding mask into
    # an additive key mask whatever `config._attn_implementation` is (`_update_full_mask`, last branch).  This container has
    # transformers 5.x, where `create_bidirectional_mask` returns None for an unset implementation — and the HubertConfig that
    # `AVHubertConfig.encoder_config` builds on the fly never has it set — so the encoder would silently attend to padded frames.
    # Naming the implementation the reference's own decoder uses ("eager": decoder.py:121-150) restores the pinned version's
    # semantics; no reference source is touched.
    m.avhubert.encoder.config._attn_implementation = "eager"
    return m
