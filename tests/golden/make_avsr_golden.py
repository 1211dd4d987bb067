"""Generate tests/golden/avsr_ref_{tiny,base}.npz — outputs of the REFERENCE ITSELF (pkg/avsr/src/avhubert/modeling_avhubert.py
AVHubertForConditionalGeneration: forward :256-314, generate through transformers' GenerationMixin) on this repo's seeded
synthetic weights and inputs.  Run in the BUILD container (CPU; the reference tree must be at /root/reference):

    python tests/golden/make_avsr_golden.py [tiny] [base] [config]

This is the one model family whose checker is the reference's own code: the modules are imported unchanged (oracle/_ref_avsr.py)
under this container's torch / transformers, `load_state_dict(strict=True)` takes runtime/avsr_weights.py's synthetic state dict
(seed in the file), the inputs are runtime/avsr_synth.py's clips.  `use_cache=False` is passed to generate(): the reference
never builds a cache itself (prepare_inputs_for_generation :372-391 re-feeds the whole prefix AND re-runs the encoder every step)
and transformers 5's default DynamicCache cannot be built from AVHubertConfig (no num_hidden_layers).

Stored per configuration (B clips, T frames, ragged lengths):
  cfg_*/seed/lens/input_sha256       what was run
  enc                                 avhubert(...).last_hidden_state [B][T][d]          (tiny: all clips; base: fingerprint + clip 0)
  tap_*                               clip 0: video front-end output, audio projection, fused LayerNorm, post_extract_proj,
                                      encoder.layer_norm (after the positional convolution), encoder layers 0 / mid / last
  enc_proj                            enc @ R [d][8] for every clip (R seeded N(0, 1) / sqrt(d))
  enc_audio_only / enc_video_only     (tiny) last_hidden_state with only one modality passed
  greedy                              generate(num_beams=1, do_sample=False, max_new_tokens=N)  [B][1 + N] (prompt = bos)
  beam / beam_scores                  generate(num_beams=K, ...) sequences and sequences_scores for the first `beam_clips` clips
  logits                              teacher-forced forward(decoder_input_ids = greedy[:, :-1]).logits: clips 0-1 in full, all clips
                                      as logits @ Rv [V][8]
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from reazonspeech_amd.runtime.avsr_config import AVSR_BASE, AVSR_TINY                  # noqa: E402
from reazonspeech_amd.runtime.avsr_synth import synthetic_clips                        # noqa: E402
from reazonspeech_amd.runtime.avsr_weights import synthetic_state_dict_avsr            # noqa: E402
from oracle import _ref_avsr as ra                                                     # noqa: E402

PROJ_SEED = 20240930
RECIPES = {     # name: (config, weight seed, input seed, clips, frames, new tokens, beams, clips in the beam run)
    "tiny": (AVSR_TINY, 0, 11, 5, 24, 10, 3, 5),
    "base": (AVSR_BASE, 0, 4242, 16, 100, 12, 5, 4),
}


def projection(n, seed=PROJ_SEED, dim=8):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((n, dim), generator=g, dtype=torch.float32) / n ** 0.5


def run(name):
    cfg, wseed, iseed, B, T, new_tokens, beams, beam_clips = RECIPES[name]
    sd = synthetic_state_dict_avsr(cfg, wseed)
    model = ra.build(cfg, sd)
    a, v, mask, lens = synthetic_clips(B, T, seed=iseed, ragged=True, min_frames=max(8, T // 3))
    kw = dict(input_values=torch.from_numpy(a), pixel_values=torch.from_numpy(v), padding_mask=torch.from_numpy(mask))
    h = hashlib.sha256(a.tobytes() + v.tobytes() + mask.tobytes()).digest()
    store = {"weight_seed": np.int64(wseed), "input_seed": np.int64(iseed), "clips": np.int64(B), "frames": np.int64(T), "lens": lens,
             "new_tokens": np.int64(new_tokens), "beams": np.int64(beams), "beam_clips": np.int64(beam_clips), "proj_seed": np.int64(PROJ_SEED),
             "input_sha256": np.frombuffer(h, np.uint8)}
    taps = {}
    av = model.avhubert
    mid = cfg.encoder_layers // 2
    hooks = [
        av.feature_extractor_video.register_forward_hook(lambda m, i, o: taps.__setitem__("tap_video", o.transpose(1, 2)[0].clone())),
        av.feature_extractor_audio.register_forward_hook(lambda m, i, o: taps.__setitem__("tap_audio", o.transpose(1, 2)[0].clone())),
        av.layer_norm.register_forward_hook(lambda m, i, o: taps.__setitem__("tap_fused_ln", o[0].clone())),
        av.encoder.layer_norm.register_forward_hook(lambda m, i, o: taps.__setitem__("tap_enc_ln", o[0].clone())),
        av.encoder.layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("tap_layer0", o[0][0].clone())),
        av.encoder.layers[mid].register_forward_hook(lambda m, i, o: taps.__setitem__("tap_layer_mid", o[0][0].clone())),
    ]
    if av.post_extract_proj is not None:
        hooks.append(av.post_extract_proj.register_forward_hook(lambda m, i, o: taps.__setitem__("tap_post_proj", o[0].clone())))
    t0 = time.time()
    with torch.no_grad():
        enc = av(**kw).last_hidden_state
    for hk in hooks:
        hk.remove()
    print(f"[{name}] encoder forward {time.time() - t0:.1f} s, |enc| mean {enc.abs().mean():.3f}", flush=True)
    for k, t in taps.items():
        store[k] = t.numpy()
    R = projection(cfg.encoder_embed_dim)
    store["enc_proj"] = (enc @ R).numpy()
    store["enc"] = enc.numpy() if name == "tiny" else enc[:1].numpy()
    if name == "tiny":                       # a missing modality (modeling_avhubert.py:172-177: zero features in its place)
        with torch.no_grad():
            store["enc_audio_only"] = av(input_values=kw["input_values"], padding_mask=kw["padding_mask"]).last_hidden_state.numpy()
            store["enc_video_only"] = av(pixel_values=kw["pixel_values"], padding_mask=kw["padding_mask"]).last_hidden_state.numpy()
    t0 = time.time()
    with torch.no_grad():
        greedy = model.generate(**kw, num_beams=1, do_sample=False, max_new_tokens=new_tokens, use_cache=False)
    print(f"[{name}] greedy generate {time.time() - t0:.1f} s: {greedy[:3].tolist()}", flush=True)
    store["greedy"] = greedy.numpy().astype(np.int32)
    t0 = time.time()
    kb = {k: t[:beam_clips] for k, t in kw.items()}
    with torch.no_grad():
        out = model.generate(**kb, num_beams=beams, do_sample=False, max_new_tokens=new_tokens, use_cache=False, return_dict_in_generate=True,
                             output_scores=True)
    print(f"[{name}] beam-{beams} generate {time.time() - t0:.1f} s: {out.sequences[:2].tolist()} {out.sequences_scores[:2].tolist()}", flush=True)
    store["beam"] = out.sequences.numpy().astype(np.int32)
    store["beam_scores"] = out.sequences_scores.numpy().astype(np.float32)
    with torch.no_grad():
        logits = model(**kw, decoder_input_ids=greedy[:, :-1], decoder_attention_mask=torch.ones_like(greedy[:, :-1])).logits
    store["logits"] = logits[:2].numpy()
    store["logits_proj"] = (logits @ projection(cfg.vocab_size, PROJ_SEED + 1)).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"avsr_ref_{name}.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes", flush=True)


def write_configs():
    """tests/golden/avsr_ref_config_{tiny,base}.json: `config.json` exactly as the reference's AVHubertConfig serialises itself
    (PretrainedConfig.to_json_string of the model built for the goldens) — what runtime/avsr_weights.read_avsr has to parse"""
    for name, (cfg, wseed, *_rest) in RECIPES.items():
        model = ra.build(cfg, synthetic_state_dict_avsr(cfg, wseed))
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"avsr_ref_config_{name}.json")
        with open(path, "w", encoding="utf-8") as fp:
            fp.write(model.config.to_json_string())
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    if "config" in sys.argv[1:]:
        write_configs()
    for name in ([a for a in sys.argv[1:] if a in RECIPES] or ([] if "config" in sys.argv[1:] else list(RECIPES))):
        run(name)
