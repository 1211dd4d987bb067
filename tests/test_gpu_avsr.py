"""-m gpu: `reazonspeech.avsr` — the AV-HuBERT encoder-decoder (pkg/avsr/src/avhubert/, SURVEY.md §8f row 4, BASELINE.json configs[4])
through the C ABI (rs_avsr_*; csrc/k_avsr.hip) against THE REFERENCE ITSELF: tests/golden/avsr_ref_{tiny,base}.npz are outputs of the
reference's own modules on this repo's seeded synthetic weights and inputs (generator tests/golden/make_avsr_golden.py; the CPU suite
holds oracle/avsr.py to the same files).  Both sides compute float32.

Stated tolerances (float32 MFMA chains vs the CPU's summation orders):
  video front-end / fused LayerNorm / encoder.layer_norm / encoder layers / last_hidden_state     max |err| <= 5e-4 (O(1) activations)
  teacher-forced logits                                                                           <= 2e-3 (|logits| up to ~20)
  greedy ids, beam-search ids                                                                     IDENTICAL to the reference's generate()
  beam-search scores                                                                              <= 1e-3
  KV-cache decoding == the reference's full-prefix recomputation (the goldens were made without a cache)
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.avsr_config import AVSR_TINY, AVSR_BASE
from reazonspeech_amd.runtime.avsr_synth import synthetic_clips
from reazonspeech_amd.runtime.avsr_weights import synthetic_state_dict_avsr
from reazonspeech_amd.avsr import AVHubertForConditionalGeneration
from oracle import avsr as oa

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {"tiny": AVSR_TINY, "base": AVSR_BASE}
TOL_ENC, TOL_LOGITS, TOL_SCORE = 5e-4, 2e-3, 1e-3


def load(name):
    g = np.load(os.path.join(HERE, "golden", f"avsr_ref_{name}.npz"))
    cfg = CONFIGS[name]
    B, T = int(g["clips"]), int(g["frames"])
    a, v, mask, lens = synthetic_clips(B, T, seed=int(g["input_seed"]), ragged=True, min_frames=max(8, T // 3))
    assert hashlib.sha256(a.tobytes() + v.tobytes() + mask.tobytes()).digest() == bytes(g["input_sha256"].tolist()), "inputs drifted from the golden's"
    sd = synthetic_state_dict_avsr(cfg, int(g["weight_seed"]))
    return g, cfg, sd, a, v, mask


def check_against_reference(name, products="exact"):
    g, cfg, sd, a, v, mask = load(name)
    model = AVHubertForConditionalGeneration(cfg, sd, device="cuda:0", products=products)
    mid = cfg.encoder_layers // 2
    enc, taps = model.dev.encode(a, v, mask, taps=[0, mid])
    torch.cuda.synchronize()
    enc_c = enc.cpu()
    stats = {}
    pairs = [("video", taps["video"][0], g["tap_video"]), ("fused_ln", taps["fused_ln"][0], g["tap_fused_ln"]), ("enc_ln", taps["enc_ln"][0], g["tap_enc_ln"]),
             ("layer0", taps["layers"][0, 0], g["tap_layer0"]), ("layer_mid", taps["layers"][1, 0], g["tap_layer_mid"])]
    for key, got, want in pairs:
        stats[key] = float((got.cpu() - torch.from_numpy(want)).abs().max())
        assert stats[key] <= TOL_ENC, (key, stats[key])
    n_full = g["enc"].shape[0]
    stats["enc"] = float((enc_c[:n_full] - torch.from_numpy(g["enc"])).abs().max())
    R = torch.randn((cfg.encoder_embed_dim, 8), generator=torch.Generator().manual_seed(int(g["proj_seed"]))) / cfg.encoder_embed_dim ** 0.5
    stats["enc_proj_all_clips"] = float(((enc_c @ R) - torch.from_numpy(g["enc_proj"])).abs().max())
    assert stats["enc"] <= TOL_ENC and stats["enc_proj_all_clips"] <= TOL_ENC, stats
    # teacher-forced logits through the KV-cache decoder == the reference's full-prefix forward
    ids = g["greedy"][:, :-1]
    logits = model(input_values=a, pixel_values=v, padding_mask=mask, decoder_input_ids=ids).logits.cpu()
    stats["logits_clips01"] = float((logits[:2] - torch.from_numpy(g["logits"])).abs().max())
    Rv = torch.randn((cfg.vocab_size, 8), generator=torch.Generator().manual_seed(int(g["proj_seed"]) + 1)) / cfg.vocab_size ** 0.5
    stats["logits_proj_all_clips"] = float(((logits @ Rv) - torch.from_numpy(g["logits_proj"])).abs().max())
    assert stats["logits_clips01"] <= TOL_LOGITS and stats["logits_proj_all_clips"] <= TOL_LOGITS, stats
    # generate(): greedy on every clip, beam search on the golden's first clips — ids identical to the reference's
    n_new, K, kb = int(g["new_tokens"]), int(g["beams"]), int(g["beam_clips"])
    greedy = model.generate(input_values=a, pixel_values=v, padding_mask=mask, num_beams=1, max_new_tokens=n_new)
    assert np.array_equal(greedy.numpy(), g["greedy"]), "greedy ids differ from the reference's generate()"
    out = model.generate(input_values=a[:kb], pixel_values=v[:kb], padding_mask=mask[:kb], num_beams=K, max_new_tokens=n_new, return_dict_in_generate=True)
    assert np.array_equal(out.sequences.numpy(), g["beam"]), "beam-search ids differ from the reference's generate()"
    stats["beam_scores"] = float(np.abs(out.sequences_scores.numpy() - g["beam_scores"]).max())
    assert stats["beam_scores"] <= TOL_SCORE, stats
    stats["distinct_greedy_tokens"] = int(len(set(g["greedy"].reshape(-1).tolist())))
    return model, stats, (g, cfg, sd, a, v, mask, enc)


@pytest.mark.parametrize("name", ["tiny", "base"])
def test_three_term_bf16_products_vs_the_reference(gpu_device, name):
    """products="x3" (csrc/k_f32.hip X3: every float32 product of the big GEMMs / convolutions as three bf16 matrix-core terms, float32
    accumulation) against the SAME goldens and the SAME tolerances as the exact mode: taps, encoder, teacher-forced logits, and the ids
    of greedy and beam search identical to the reference's generate().  Its errors are printed next to the exact mode's."""
    model, stats, _ = check_against_reference(name, products="x3")
    assert model.dev.products == "x3"
    print(f"avsr {name} vs reference, x3 products:", stats)
    with pytest.raises(ValueError):
        model.dev.set_products("bf16")


def test_tiny_vs_the_reference(gpu_device):
    model, stats, (g, cfg, sd, a, v, mask, enc) = check_against_reference("tiny")
    print("avsr tiny vs reference:", stats)
    # and against the CPU oracle on fresh inputs (what smoke() and bench.py check on the GPU box, where the reference is absent)
    a2, v2, m2, _ = synthetic_clips(3, 19, seed=77, ragged=True)
    with torch.no_grad():
        want = oa.encode(cfg, sd, torch.from_numpy(a2), torch.from_numpy(v2), torch.from_numpy(m2))
        got = model.avhubert(input_values=a2, pixel_values=v2, padding_mask=m2).last_hidden_state.cpu()
        assert (got - want).abs().max() <= TOL_ENC
        assert np.array_equal(model.generate(input_values=a2, pixel_values=v2, padding_mask=m2, num_beams=4, max_new_tokens=9).numpy(),
                              oa.beam_generate(cfg, sd, want, torch.from_numpy(m2), 4, 9)[0].numpy())
    # one modality only (the reference puts zero FEATURES in the other's place: modeling_avhubert.py:172-177)
    for kw, key in ((dict(input_values=a, padding_mask=mask), "enc_audio_only"), (dict(pixel_values=v, padding_mask=mask), "enc_video_only")):
        got = model.avhubert(**kw).last_hidden_state.cpu()
        assert (got - torch.from_numpy(g[key])).abs().max() <= TOL_ENC, key
    with pytest.raises(ValueError):
        model.avhubert(padding_mask=mask)
    # a clip alone == inside the batch when it is the longest one (no padding differences): bits
    lens = g["lens"]
    b = int(np.argmax(lens))
    if int(lens[b]) == a.shape[1]:
        alone = model.avhubert(input_values=a[b:b + 1], pixel_values=v[b:b + 1], padding_mask=mask[b:b + 1]).last_hidden_state
        assert torch.equal(alone[0], enc[b])


def test_base_161m_vs_the_reference(gpu_device):
    """the reference's default geometry (12 x 768 encoder, ResNet-18 at 88 x 88, 6-layer decoder, 161M parameters): 16 ragged clips"""
    _, stats, _ = check_against_reference("base")
    print("avsr base vs reference:", stats)
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    import json
    json.dump(stats, open(os.path.join(os.path.dirname(HERE), "gpurun_out", "avsr_parity.json"), "w"), indent=1, sort_keys=True)


def test_the_readme_flow_from_a_pretrained_directory(gpu_device, tmp_path):
    """pkg/avsr/README.rst end to end: AVHubertProcessor.from_pretrained(dir) / AVHubertForConditionalGeneration.from_pretrained(dir),
    inputs = processor(raw_audio=, raw_video=), outputs = model.generate(**inputs, num_beams=5, max_new_tokens=...),
    processor.decode(outputs[0], skip_special_tokens=True).  The directory holds config.json AS THE REFERENCE WRITES IT
    (tests/golden/avsr_ref_config_tiny.json), model.safetensors under the reference's parameter names, preprocessor_config.json and a
    PreTrainedTokenizerFast; the result equals the model built from (config, state dict) directly."""
    from safetensors.torch import save_file
    from test_avsr_host import make_processor_dir
    from reazonspeech_amd.avsr import AVHubertProcessor
    cfg = AVSR_TINY
    sd = synthetic_state_dict_avsr(cfg, 0)
    with open(os.path.join(HERE, "golden", "avsr_ref_config_tiny.json"), encoding="utf-8") as fp:
        (tmp_path / "config.json").write_text(fp.read(), encoding="utf-8")
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    make_processor_dir(str(tmp_path), vocab_size=cfg.vocab_size)
    processor = AVHubertProcessor.from_pretrained(str(tmp_path))
    model = AVHubertForConditionalGeneration.from_pretrained(str(tmp_path), device=str(gpu_device))
    assert model.config == cfg
    rng = np.random.default_rng(5)
    clips = [((0.1 * rng.standard_normal(n)).astype(np.float32), rng.integers(0, 256, (n // 640, 96, 96), dtype=np.uint8)) for n in (16000, 9600)]
    inputs = processor(raw_audio=[a for a, _ in clips], raw_video=[v for _, v in clips])
    assert inputs["input_values"].shape == (2, 25, 104) and inputs["padding_mask"][1, 15:].all() and not inputs["padding_mask"][1, :15].any()
    outputs = model.generate(**inputs, num_beams=5, max_new_tokens=8)
    assert outputs.shape[0] == 2 and outputs.shape[1] <= 9 and (outputs[:, 0] == cfg.bos_token_id).all()      # the prompt the reference's generate() starts from (goldens)
    text = processor.decode(outputs[0], skip_special_tokens=True)
    assert isinstance(text, str) and "<s>" not in text and "</s>" not in text
    assert len(processor.batch_decode(outputs, skip_special_tokens=True)) == 2
    direct = AVHubertForConditionalGeneration(cfg, sd, device=str(gpu_device)).generate(**inputs, num_beams=5, max_new_tokens=8)
    assert torch.equal(direct, outputs)
    # generate() options: the neutral values of transformers' defaults pass, max_length counts the bos prompt, anything that would
    # change the search is refused instead of dropped
    assert torch.equal(model.generate(**inputs, num_beams=5, max_length=9, early_stopping=False, repetition_penalty=1.0, use_cache=True), outputs)
    with pytest.raises(NotImplementedError):
        model.generate(**inputs, num_beams=5, max_new_tokens=8, repetition_penalty=1.2)
    with pytest.raises(NotImplementedError):
        model.generate(**inputs, num_beams=5, max_new_tokens=8, do_sample=True)
    with pytest.raises(TypeError):
        model.generate(**inputs, num_beams=5, max_new_tokens=8, logits_processor=[])
    # one clip alone, audio only (the processor substitutes a zero video; README's extractor path for the pretrained encoder)
    solo = processor(raw_audio=clips[0][0])
    enc = model.avhubert(**solo).last_hidden_state
    assert enc.shape == (1, 25, cfg.encoder_embed_dim) and torch.isfinite(enc).all()


def test_long_decoding_crosses_every_attention_form(gpu_device):
    """README.rst asks for max_new_tokens=256: the self-attention of a decoding step then runs over 1 .. 257 cached keys, and a longer
    prompt beyond 512 — the step kernel's one- and two-chunk forms and the general kernel.  Teacher-forced logits over 530 positions
    through the KV cache (and a re-parented cache, the beam-search path) against the oracle's full-prefix forward, toy geometry."""
    cfg = AVSR_TINY.with_(max_target_positions=640)
    sd = synthetic_state_dict_avsr(cfg, 3)
    a, v, mask, _ = synthetic_clips(2, 21, seed=5, ragged=True)
    model = AVHubertForConditionalGeneration(cfg, sd, device=str(gpu_device))
    L = 530
    ids = torch.randint(3, cfg.vocab_size, (2, L), generator=torch.Generator().manual_seed(9))
    ids[:, 0] = cfg.bos_token_id
    got = model(input_values=a, pixel_values=v, padding_mask=mask, decoder_input_ids=ids).logits.cpu()
    with torch.no_grad():
        enc = oa.encode(cfg, sd, torch.from_numpy(a), torch.from_numpy(v), torch.from_numpy(mask))
        want = oa.decode_logits(cfg, sd, enc, torch.from_numpy(mask), ids)
    for lo, hi in ((0, 64), (64, 256), (256, 512), (512, L)):
        err = float((got[:, lo:hi] - want[:, lo:hi]).abs().max())
        assert err <= TOL_LOGITS, (lo, hi, err)
    # beam rows: three hypotheses per clip fed the same tokens, the cache re-gathered by a permutation of a clip's rows every step
    enc_d = model.avhubert(input_values=a, pixel_values=v, padding_mask=mask).last_hidden_state
    dec = model.dev.decoding(enc_d, mask, 3, 300)
    perm = np.array([1, 2, 0, 5, 3, 4])
    for t in range(290):
        lg = dec.step(np.repeat(ids[:, t].numpy(), 3), t, perm if t else None)
    torch.cuda.synchronize()
    lg = lg.cpu().view(2, 3, -1)
    for r in range(3):
        assert float((lg[:, r] - want[:, 289]).abs().max()) <= TOL_LOGITS, r
