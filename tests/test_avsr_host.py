"""Host side of `reazonspeech.avsr` that needs no GPU: what `from_pretrained` reads (config.json as the REFERENCE serialises its
AVHubertConfig — tests/golden/avsr_ref_config_*.json, written by tests/golden/make_avsr_golden.py from the reference's own class —,
model.safetensors, preprocessor_config.json, the PreTrainedTokenizerFast files) and the processor's text path
(pkg/avsr/src/avhubert/processing_avhubert.py:33-89, README.rst's documented calls)."""
import json
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.avsr import AVHubertFeatureExtractor, AVHubertProcessor
from reazonspeech_amd.avsr.feature_extraction import _FastTokenizer, wrap_targets
from reazonspeech_amd.runtime.avsr_config import AVSR_BASE, AVSR_TINY
from reazonspeech_amd.runtime.avsr_weights import read_avsr, synthetic_state_dict_avsr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name,cfg", [("tiny", AVSR_TINY), ("base", AVSR_BASE)])
def test_read_avsr_parses_the_references_own_config_json(tmp_path, name, cfg):
    """config.json written by the reference's AVHubertConfig (PretrainedConfig.to_json_string: lists for tuples, id2label,
    transformers_version, training fields ...) + model.safetensors under the reference's parameter names -> the same AvsrConfig the
    goldens were made with and the same tensors"""
    from safetensors.torch import save_file
    with open(os.path.join(GOLDEN, f"avsr_ref_config_{name}.json"), encoding="utf-8") as fp:
        raw = fp.read()
    assert json.loads(raw)["model_type"] == "avhubert"
    (tmp_path / "config.json").write_text(raw, encoding="utf-8")
    sd = synthetic_state_dict_avsr(cfg, 0) if name == "tiny" else {k: v for k, v in list(synthetic_state_dict_avsr(AVSR_TINY, 0).items())[:3]}
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    got_cfg, got_sd = read_avsr(str(tmp_path))
    assert got_cfg == cfg
    assert set(got_sd) == set(sd) and all(torch.equal(got_sd[k], sd[k]) for k in sd)


def test_read_avsr_rejects_what_is_not_built(tmp_path):
    with open(os.path.join(GOLDEN, "avsr_ref_config_tiny.json"), encoding="utf-8") as fp:
        raw = json.load(fp)
    raw["decoder_learned_pos"] = True
    (tmp_path / "config.json").write_text(json.dumps(raw), encoding="utf-8")
    with pytest.raises(AssertionError):
        read_avsr(str(tmp_path))


def make_processor_dir(path, vocab_size=12):
    """what `processor.save_pretrained` leaves: preprocessor_config.json (FeatureExtractionMixin.to_dict: constructor arguments,
    the printed transforms, bookkeeping keys) and a PreTrainedTokenizerFast's tokenizer.json / tokenizer_config.json /
    special_tokens_map.json — here a small WordLevel vocabulary"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for w in ("こんにちは", "世界", "today", "is", "fine", "mouth", "crop", "lip"):
        vocab[w] = len(vocab)
    while len(vocab) < vocab_size:
        vocab[f"w{len(vocab)}"] = len(vocab)
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tk.add_special_tokens(["<s>", "<pad>", "</s>", "<unk>"])
    tk.save(os.path.join(path, "tokenizer.json"))
    special = {"bos_token": "<s>", "eos_token": "</s>", "pad_token": "<pad>", "unk_token": "<unk>"}
    with open(os.path.join(path, "special_tokens_map.json"), "w") as fp:
        json.dump(special, fp)
    with open(os.path.join(path, "tokenizer_config.json"), "w") as fp:
        json.dump(dict(special, tokenizer_class="PreTrainedTokenizerFast", clean_up_tokenization_spaces=False), fp)
    with open(os.path.join(path, "preprocessor_config.json"), "w") as fp:
        json.dump({"feature_extractor_type": "AVHubertFeatureExtractor", "processor_class": "AVHubertProcessor", "image_crop_size": 88,
                   "image_mean": 0.421, "image_std": 0.165, "landmark_indices": [5, 411, 199, 187], "max_sample_size": None,
                   "min_detection_confidence": 0.5, "min_tracking_confidence": 0.5, "normalize": True, "refine_landmarks": False,
                   "sr": 16000, "stack_order_audio": 4, "static_image_mode": False,
                   "transforms": [{"transforms_type": "ToImage"}, {"transforms_type": "CenterCrop", "size": "(88, 88)"}]}, fp)
    return vocab


def test_processor_from_pretrained_and_the_readme_calls(tmp_path):
    """README.rst: processor = AVHubertProcessor.from_pretrained(dir); inputs = processor(raw_audio=, raw_video=);
    processor.decode(outputs[0], skip_special_tokens=True)"""
    vocab = make_processor_dir(str(tmp_path))
    proc = AVHubertProcessor.from_pretrained(str(tmp_path))
    fe = proc.feature_extractor
    assert (fe.stack_order_audio, fe.image_crop_size, fe.sr, fe.normalize, fe.max_sample_size) == (4, 88, 16000, True, None)
    assert (fe.image_mean, fe.image_std) == (0.421, 0.165)
    rng = np.random.default_rng(3)
    audio = (0.1 * rng.standard_normal(16000)).astype(np.float32)            # 1 s -> 100 filterbank frames -> 25 stacked
    video = rng.integers(0, 256, (25, 96, 96), dtype=np.uint8)               # grey mouth crops at 25 fps
    inputs = proc(raw_audio=audio, raw_video=video)
    assert inputs["input_values"].shape == (1, 25, 104) and inputs["pixel_values"].shape == (1, 25, 1, 88, 88)
    assert inputs["padding_mask"].shape == (1, 25) and not inputs["padding_mask"].any()
    ids = [vocab["</s>"], vocab["today"], vocab["is"], vocab["fine"], vocab["</s>"], vocab["<pad>"]]     # decoder start = eos, then pad
    assert proc.decode(ids, skip_special_tokens=True) == "today is fine"
    assert proc.batch_decode(torch.tensor([ids, ids]), skip_special_tokens=True) == ["today is fine"] * 2
    assert "</s>" in proc.decode(ids, skip_special_tokens=False)
    # the fallback over the `tokenizers` library alone says the same
    lite = _FastTokenizer(str(tmp_path))
    assert lite.decode(ids, skip_special_tokens=True) == "today is fine" and lite.pad_id == vocab["<pad>"]
    # feature extractor on its own (the pretrained encoder's documented path: AutoFeatureExtractor.from_pretrained)
    fe2 = AVHubertFeatureExtractor.from_pretrained(str(tmp_path))
    again = fe2(raw_audio=audio, raw_video=video)
    assert all(np.array_equal(again[k], inputs[k]) for k in inputs)
    with pytest.raises(FileNotFoundError):
        AVHubertProcessor.from_pretrained(str(tmp_path / "nowhere"))


def test_processor_text_targets(tmp_path):
    """processing_avhubert.py:55-89: every text becomes <s> text </s>; with audio / video the call also returns the
    teacher-forcing tensors — decoder_input_ids = ids[:, :-1], labels = ids[:, 1:]"""
    vocab = make_processor_dir(str(tmp_path))
    assert wrap_targets(["a", "<s>a", "a</s>", "<s>a</s>"]) == ["<s>a</s>"] * 4
    proc = AVHubertProcessor.from_pretrained(str(tmp_path))
    rng = np.random.default_rng(4)
    audio = (0.1 * rng.standard_normal(8000)).astype(np.float32)
    video = rng.integers(0, 256, (13, 88, 88), dtype=np.uint8)
    enc = proc(text="<s> lip crop </s>")
    want = [vocab["<s>"], vocab["lip"], vocab["crop"], vocab["</s>"]]
    assert np.asarray(enc["input_ids"]).tolist() == [want]
    both = proc(raw_audio=[audio], raw_video=[video], text=["<s> lip crop </s>"])
    assert np.asarray(both["decoder_input_ids"]).tolist() == [want[:-1]] and np.asarray(both["labels"]).tolist() == [want[1:]]
    assert np.asarray(both["decoder_attention_mask"]).tolist() == [[1, 1, 1]]
    assert both["input_values"].shape[0] == 1
    with pytest.raises(ValueError):
        proc()


def test_colour_frames_use_opencvs_fixed_point_grey():
    """feature_extraction_avhubert.py:72-73 converts un-cropped numpy frames with cv2.COLOR_BGR2GRAY: [UPSTREAM] OpenCV's 8-bit
    path, (1868 B + 9617 G + 4899 R + 8192) >> 14 — known answers of that formula, and a pixel where a float product rounds
    the other way"""
    fe = AVHubertFeatureExtractor()
    px = np.array([[[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0]], [[0, 0, 255], [12, 200, 77], [1, 2, 3], [90, 90, 90]]]], np.uint8)
    got = fe._load_video(px)[0, 0]
    want = np.array([[255, 0, 29, 150], [76, 142, 2, 90]], np.uint8)
    assert np.array_equal(got, want)
    rng = np.random.default_rng(0)
    c = rng.integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    fx = fe._load_video(c)[0, 0].astype(np.int64)
    fl = np.round(c[0, ..., 0] * 0.114 + c[0, ..., 1] * 0.587 + c[0, ..., 2] * 0.299)
    assert np.abs(fx - fl).max() <= 1
