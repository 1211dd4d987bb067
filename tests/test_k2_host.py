"""CPU: the host side of `reazonspeech.k2.asr` (reazonspeech_amd/k2/asr) against the REFERENCE's own files, the CPU oracle's
building blocks, and the ONNX reader.

tests/golden/reference_k2.json holds what /root/reference/pkg/k2-asr/src/{huggingface,transcribe}.py do with stubbed third-party
imports (generator: tests/golden/make_reference_k2_golden.py): which files `load_model` resolves for every (language, precision),
its ValueError messages, and `transcribe()` on the fake recogniser of tests/k2_fake.py."""
import importlib
import json
import os
import warnings

import numpy as np
import pytest
import torch

import k2_fake as fk
from k2_onnx_writer import write_k2_onnx
from reazonspeech_amd.k2.asr import interface, huggingface as hfm
from reazonspeech_amd.k2.asr.model import K2Model, read_tokens, synthetic_tokens
from reazonspeech_amd.runtime.k2_config import ZIPFORMER_TINY, ZIPFORMER_159M
from reazonspeech_amd.runtime import k2_weights as kw, k2_onnx, onnx_lite
from oracle import zipformer as oz, greedy as og

tr = importlib.import_module("reazonspeech_amd.k2.asr.transcribe")
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_k2.json")))


def test_load_model_resolves_the_reference_files_and_raises_like_it():
    for key, want in GOLD["load_model"].items():
        lang, prec = key.split("|")
        repo, files = hfm.repo_files(lang, prec)
        for part in ("tokens", "encoder", "decoder", "joiner"):
            assert want[part] == f"/cache/{repo}/{files[part]}", (key, part)
    for kwargs, msg in GOLD["errors"].items():
        with pytest.raises(ValueError) as e:
            hfm.load_model(**json.loads(kwargs))
        assert str(e.value) == msg
    with pytest.raises(RuntimeError, match="no CPU"):
        hfm.load_model(device="cpu")


def test_transcribe_matches_the_reference_on_a_fake_recogniser():
    assert (tr.PAD_SECONDS, tr.TOO_LONG_SECONDS) == (GOLD["constants"]["PAD_SECONDS"], GOLD["constants"]["TOO_LONG_SECONDS"])
    for name, secs, seed in (("short", 2.0, 1), ("ten", 10.0, 2), ("long", 29.5, 3)):
        model = fk.FakeRecognizer()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            res = tr.transcribe(model, interface.AudioData(fk.audio(secs, seed), 16000))
        want = GOLD["transcribe"][name]
        assert [list(s) for s in model.seen] == want["seen"]                 # (sample rate, padded samples, the padding is silence)
        assert res.text == want["text"] and [[s.token, s.seconds] for s in res.subwords] == want["subwords"]
        assert [str(x.message) for x in w] == want["warnings"]
    import reazonspeech.k2.asr as pub
    assert sorted(n for n in pub.__all__ if n != "transcribe_batch") == sorted(
        ["TranscribeConfig", "load_model", "transcribe", "audio_from_numpy", "audio_from_tensor", "audio_from_path"])
    assert interface.TranscribeConfig().verbose is True


def test_result_conversion_follows_sherpa_onnx_conventions(tmp_path):
    """tokens.txt parsing, U+2581 -> space, byte-fallback pieces joined into UTF-8, timestamps = frame x 0.04 s"""
    p = tmp_path / "tokens.txt"
    p.write_text("<blk> 0\n<sos/eos> 1\n<unk> 2\n▁the 3\nあ 4\n<0xE3> 5\n<0x81> 6\n<0x84> 7\n", encoding="utf-8")
    toks = read_tokens(str(p))
    assert toks[3] == "▁the" and len(toks) == 8
    conv = K2Model.convert.__get__(type("M", (), {"tokens": toks, "cfg": ZIPFORMER_TINY, "symbol": lambda self, i: toks[i].replace("▁", " ")})())
    r = conv([3, 4, 5, 6, 7], [0, 3, 10, 11, 12])
    assert r.tokens == [" the", "あ", "<0xE3>", "<0x81>", "<0x84>"] and r.text == " theあい"
    assert r.timestamps == [0.0, float(np.float32(0.12)), float(np.float32(0.4)), float(np.float32(0.44)), float(np.float32(0.48))]
    assert len(set(synthetic_tokens(97))) == 97 and synthetic_tokens(97)[:3] == ["<blk>", "<sos/eos>", "<unk>"]


# ---- oracle building blocks ([UPSTREAM] forms restated two ways) -------------------------------------------------------------
def test_rel_shift_equals_icefalls_as_strided_form():
    """the position scores are read at column (T - 1) - i + j of the (T, 2T - 1) matrix: the gather of oracle/zipformer.py equals
    the as_strided expression icefall's RelPositionMultiheadAttentionWeights.forward uses"""
    torch.manual_seed(0)
    H, B, T = 3, 2, 9
    ps = torch.randn(H, B, T, 2 * T - 1)
    strided = ps.as_strided((H, B, T, T), (ps.stride(0), ps.stride(1), ps.stride(2) - ps.stride(3), ps.stride(3)), storage_offset=ps.stride(3) * (T - 1))
    i, j = torch.arange(T).unsqueeze(1), torch.arange(T).unsqueeze(0)
    assert torch.equal(strided, ps.gather(3, (j - i + T - 1).expand(H, B, T, T)))


def test_the_oracles_own_tables_equal_the_products_and_have_the_published_values():
    """oracle/zipformer.py builds its window, mel banks and relative-position rows itself (float64, from the published
    kaldi-native-fbank / icefall formulas); the arrays the HIP path uploads (runtime/k2_weights.py) must agree with them to
    float32 resolution, so a wrong mel edge / window exponent / position formula on EITHER side fails here, and a handful of
    values are checked against numbers worked out by hand from the formulas"""
    import inspect
    imports = [ln for ln in inspect.getsource(oz).splitlines() if ln.startswith(("import ", "from "))]
    assert imports and not any("reazonspeech_amd" in ln for ln in imports), "the oracle must not import its constants from the product"
    for cfg in (ZIPFORMER_159M, ZIPFORMER_TINY):
        w64, w32 = oz.povey_window(cfg.frame_length), kw.povey_window(cfg.frame_length)
        assert w64.dtype == np.float64 and np.array_equal(w64.astype(np.float32), w32)
        b64, b32 = oz.kaldi_mel_banks(cfg), kw.kaldi_mel_banks(cfg)
        assert b64.shape == b32.shape and np.abs(b64 - b32).max() < 1e-7 and np.array_equal(b64 > 0, b32 > 0)
        for T in (1, 2, 37, 293, 586):
            p64, p32 = oz.compact_rel_pos_table(cfg, T), kw.compact_rel_pos_table(cfg, T)
            assert p64.shape == p32.shape == (2 * T - 1, cfg.pos_dim) and np.abs(p64 - p32).max() < 1e-5
        for s in range(cfg.n_stacks):
            assert oz.layer_prefix(cfg, s, 1) == kw.layer_prefix(cfg, s, 1)
    cfg = ZIPFORMER_159M
    # povey: (0.5 - 0.5 cos(2 pi i / 399)) ** 0.85 at i = 100 -> (0.5 - 0.5 cos(1.574733..)) ** 0.85
    assert abs(oz.povey_window(400)[100] - (0.5 - 0.5 * np.cos(2 * np.pi * 100 / 399)) ** 0.85) < 1e-15
    assert abs(oz.povey_window(400)[100] - 0.55664) < 1e-5       # (0.50222) ** 0.85
    # mel: 1127 ln(1 + f / 700); 20 Hz -> 31.75, 7600 Hz -> 2787.0 (1127 x ln 11.857 = 1127 x 2.47293); 81 steps of 34.0; bank 0 peaks at mel 65.75 = 42.05 Hz,
    # between FFT bins 1 (31.25 Hz) and 2 (62.5 Hz): both carry weight, bin 0 and bin 3 (93.75 Hz = mel 141.6 > right edge 99.7) none
    assert abs(float(oz.mel_scale(20.0)) - 31.748) < 1e-2 and abs(float(oz.mel_scale(7600.0)) - 2786.99) < 1e-2
    fb = oz.kaldi_mel_banks(cfg)
    assert fb[0, 0] == 0 and fb[0, 1] > 0 and fb[0, 2] > 0 and fb[0, 3] == 0
    m1, lo, step = float(oz.mel_scale(31.25)), float(oz.mel_scale(20.0)), (float(oz.mel_scale(7600.0)) - float(oz.mel_scale(20.0))) / 81
    assert abs(fb[0, 1] - (m1 - lo) / step) < 1e-12
    assert fb[79, 243] > 0 and fb[79, 244] == 0                   # 7600 Hz / 31.25 = bin 243.2: the last filter ends there
    # compact rel-pos: rel 0 -> angle 0 -> cos columns 1, sin columns 0, last column 1; antisymmetric sines
    pe = oz.compact_rel_pos_table(cfg, 5)
    assert np.allclose(pe[4, 0::2][:-1], 1.0) and np.allclose(pe[4, 1::2][:-1], 0.0) and (pe[:, -1] == 1.0).all()
    assert np.allclose(pe[0, 1:-1:2], -pe[8, 1:-1:2]) and np.allclose(pe[0, 0::2], pe[8, 0::2])
    a1 = np.arctan(np.sqrt(48.0) * (np.log(1 + np.sqrt(48.0)) - np.log(np.sqrt(48.0))) / (48 / (2 * np.pi)))
    assert abs(pe[5, 0] - np.cos(a1)) < 1e-15 and abs(pe[5, 3] - np.sin(2 * a1)) < 1e-15


def test_fbank_geometry_and_invariances():
    cfg = ZIPFORMER_159M
    assert cfg.fbank_frames(188800) == 1180 and cfg.embed_frames(1180) == 586 and cfg.enc_frames(1180) == 293
    fb = kw.kaldi_mel_banks(cfg)
    assert fb.shape == (80, 257) and fb[:, 256].max() == 0 and (fb.sum(1) > 0).all() and fb.max() <= 1.0
    assert (fb[:, 0] == 0).all()                                  # low_freq 20 Hz: the DC bin belongs to no filter
    w = kw.povey_window(400)
    assert w[0] == 0 and abs(w[200] - 1.0) < 1e-3 and np.allclose(w, w[::-1], atol=1e-6)
    wav = torch.from_numpy(fk.audio(1.0, 5))
    a = oz.fbank(cfg, wav)
    assert a.shape == (100, 80)
    # remove_dc_offset: a constant offset does not change the features; the reflected edges make the first frame finite
    b = oz.fbank(cfg, wav + 0.25)
    assert (a - b).abs().max() < 2e-3 and torch.isfinite(a).all()
    assert torch.allclose(oz.fbank(cfg, torch.zeros(16000)), torch.full((100, 80), float(np.log(np.float32(1.1920929e-07)))))


def test_simple_downsample_repeats_the_last_frame_and_bias_norm_scales_x():
    x = torch.arange(10, dtype=torch.float32).reshape(5, 2)
    out = oz.simple_downsample(x, torch.zeros(2), 2)
    assert torch.allclose(out, torch.tensor([[1., 2.], [5., 6.], [8., 9.]]))          # the last group is (frame 4, frame 4)
    y = oz.bias_norm(torch.tensor([[3.0, 4.0]]), torch.tensor([0.0, 0.0]), torch.tensor(0.0))
    assert torch.allclose(y, torch.tensor([[3.0, 4.0]]) / (12.5 ** 0.5))


def test_c_greedy_equals_the_torch_restatement_of_sherpa_onnx_greedy_search():
    cfg = ZIPFORMER_TINY
    sd = kw.synthetic_state_dict_k2(cfg, 7)
    sd["joiner.output_linear.bias"][cfg.unk_id] += 5.0             # make <unk> win now and then: it must be skipped like the blank
    torch.manual_seed(1)
    f = torch.randn(3, 60, cfg.joiner_dim)
    lens = [60, 41, 0]
    got = og.k2_greedy(cfg, sd, f.numpy(), lens)
    for b in range(3):
        ids, frames = oz.greedy_search(cfg, sd, f[b, :lens[b]])
        assert (ids, frames) == (got[b][0], got[b][1])
        assert cfg.unk_id not in ids
    assert sum(len(g[0]) for g in got) > 10


def test_oracle_runs_one_utterance_and_the_bf16_recipe_stays_close():
    cfg = ZIPFORMER_TINY
    sd = kw.synthetic_state_dict_k2(cfg, 3)
    wav = np.pad(fk.audio(1.5, 2), 14400)
    a, b = oz.forward(cfg, sd, wav, "fp32"), oz.forward(cfg, sd, wav, "bf16")
    assert a["enc"].shape == (cfg.enc_frames(cfg.fbank_frames(len(wav))), cfg.out_dim)
    assert (a["joint_enc"] - b["joint_enc"]).abs().max() < 0.1


# ---- weights and the ONNX reader ----------------------------------------------------------------------------------------------
def test_parameter_count_and_prepared_layouts():
    assert abs(ZIPFORMER_159M.n_params() / 1e6 - 158.3) < 0.5          # README.rst:27-28: "159M"
    cfg = ZIPFORMER_TINY
    sd = kw.synthetic_state_dict_k2(cfg, 0)
    assert sum(v.numel() for v in sd.values()) == cfg.n_params()
    w = kw.prepare_weights_k2(cfg, sd, 32)
    assert w["S0.L0.attw.pos_proj"].shape == (63, cfg.num_heads[0] * 4) and w["emb.conv2.w"].shape == (64, 192)
    assert w["S0.L0.sa1.out.w"].shape == (64, 64) and torch.all(w["S0.L0.sa1.out.w"][:, 24:] == 0)      # K padded 24 -> 64
    assert abs(float(w["S1.ds.w"][:2].sum()) - 1.0) < 1e-6 and torch.all(w["S1.ds.w"][2:] == 0)
    bad = dict(sd)
    bad["encoder.extra.weight"] = torch.zeros(1)
    with pytest.raises(ValueError, match="no counterpart"):
        kw.prepare_weights_k2(cfg, bad, 32)


def test_onnx_reader_round_trip(tmp_path):
    """three files in the layout icefall's export gives (named conv / bias tensors, anonymous transposed MatMul operands under
    scoped node names, constant-folded BiasNorm scales and down-sampling weights) -> the same configuration and weights"""
    cfg = ZIPFORMER_TINY
    sd = kw.synthetic_state_dict_k2(cfg, 5)
    paths = [str(tmp_path / n) for n in ("encoder.onnx", "decoder.onnx", "joiner.onnx")]
    write_k2_onnx(cfg, sd, *paths)
    m = onnx_lite.load(paths[1])
    assert m.metadata["context_size"] == "2" and "decoder.embedding.weight" in m.initializers
    cfg2, sd2 = k2_onnx.read_k2_onnx(*paths)
    assert cfg2 == cfg.with_(unk_id=cfg2.unk_id) and set(sd2) == set(sd)
    for k in sd:
        if k.endswith("downsample.bias") or k.endswith("downsample_output.bias"):
            assert torch.allclose(torch.softmax(sd[k], 0), torch.softmax(sd2[k], 0), atol=1e-6)
        else:
            assert torch.allclose(sd[k].float(), sd2[k].float(), atol=1e-6), k
    a = kw.prepare_weights_k2(cfg, sd, 16)
    b = kw.prepare_weights_k2(cfg2, sd2, 16)
    assert all(torch.allclose(a[k].float(), b[k].float(), atol=1e-5) for k in a)
    q = onnx_lite.Model(nodes=[onnx_lite.Node("/x/MatMulInteger", "MatMulInteger", ["a", "b"], ["c"])])
    onnx_lite.dump(paths[0], q)
    with pytest.raises(ValueError, match="quantized"):
        k2_onnx.read_k2_onnx(*paths)
    # the reader checks what it recovered: a dropped tensor, a down-sampling constant that is not a softmax, and two
    # different constants landing on one key are refused instead of loading silently
    write_k2_onnx(cfg, {k: v for k, v in sd.items() if k != "encoder.encoders.1.encoder.layers.0.bypass.bypass_scale"}, *paths)
    with pytest.raises(ValueError, match="not found"):
        k2_onnx.read_k2_onnx(*paths)
    write_k2_onnx(cfg, sd, *paths)
    m = onnx_lite.load(paths[0])
    for n in m.nodes:
        if n.name.endswith("encoders.1/downsample/Mul"):
            m.initializers[n.inputs[1]] = m.initializers[n.inputs[1]] * 3.0
    onnx_lite.dump(paths[0], m)
    with pytest.raises(ValueError, match="softmax"):
        k2_onnx.read_k2_onnx(*paths)
    write_k2_onnx(cfg, sd, *paths)
    m = onnx_lite.load(paths[0])
    twin = [n for n in m.nodes if n.name.endswith("layers.0/norm/Mul")][0]
    m.initializers["dup_scale"] = np.asarray(2.5, np.float32)
    m.nodes.append(onnx_lite.Node(twin.name, "Mul", ["x", "dup_scale"], ["y2"]))
    onnx_lite.dump(paths[0], m)
    with pytest.raises(ValueError, match="two different constants"):
        k2_onnx.read_k2_onnx(*paths)
