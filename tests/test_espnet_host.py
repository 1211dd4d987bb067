"""CPU: the host side of `reazonspeech.espnet.asr` (reazonspeech_amd/espnet/asr) against the REFERENCE's own files.

tests/golden/reference_espnet.json holds what /root/reference/pkg/espnet-asr/src/{transcribe,ctc}.py produce on the fake
model of tests/espnet_fake.py (generator: tests/golden/make_reference_espnet_golden.py); the same fake model through this
repo's restatement must give the same windows, texts, segments and timestamps.  The aligner itself (`ctc_segmentation`, a
third-party package the reference imports) is this repo's restatement on BOTH sides: it is tested on its own below."""
import json
import os

import numpy as np
import pytest

import espnet_fake as fk
import importlib

from reazonspeech_amd.espnet.asr import ctc, ctc_segmentation, interface

tr = importlib.import_module("reazonspeech_amd.espnet.asr.transcribe")
from reazonspeech_amd.espnet.asr.model import synthetic_token_list

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_espnet.json")))
CASES = {"short_3s": (3.0, 1), "one_window_19s": (19.0, 2), "long_47s": (47.3, 3), "long_90s": (90.0, 4)}


@pytest.mark.parametrize("name", list(CASES))
def test_transcribe_windowing_and_segments_match_the_reference(name):
    """20 s windows cut at the midpoint of the longest CTC-blank stretch (transcribe.py:59-67), (16000, 8000) padding (:69),
    segments offset by the window position (:72-77)"""
    secs, seed = CASES[name]
    wav = fk.long_audio(secs, seed)
    model = fk.FakeEspnetModel()
    res = tr.transcribe(model, interface.AudioData(wav, 16000), interface.TranscribeConfig(verbose=False))
    want = GOLD["cases"][name]
    assert model.calls == want["windows"]
    assert res.text == want["text"]
    got = [[float(s.start_seconds), float(s.end_seconds), s.text] for s in res.segments]
    assert len(got) == len(want["segments"])
    for a, b in zip(got, want["segments"]):
        assert a[2] == b[2] and abs(a[0] - b[0]) < 1e-9 and abs(a[1] - b[1]) < 1e-9
    blank = ctc.find_blank(model, wav[:20 * 16000])
    assert [int(blank.start), int(blank.end)] == want["find_blank_first_window"]
    assert sum(model.calls) == len(wav)                      # the windows tile the audio exactly


def test_find_blank_matches_the_reference_on_arbitrary_blank_patterns():
    """120 blank-posterior patterns (runs touching either end, ties, fewer samples than frames, all silent / all speech): the
    run-length implementation returns what the reference's per-frame scan (ctc.py:29-58) returned"""
    pats = fk.blank_patterns()
    assert len(pats) == len(GOLD["find_blank"])
    for (n, col), want in zip(pats, GOLD["find_blank"]):
        b = ctc.find_blank(fk.ColumnModel(col), np.zeros(n, np.float32))
        assert [int(b.start), int(b.end)] == want, (n, col.tolist())


def test_find_end_of_segment_matches_the_reference():
    for case in GOLD["find_end_of_segment"]:
        assert ctc.find_end_of_segment(case["text"], case["timings"], case["start"]) == case["end"], case


def test_constants_and_exports():
    assert (tr.WINDOW_SECONDS, tr.PADDING) == (20, (16000, 8000))                         # transcribe.py:9-10
    assert (ctc.PHONEMIC_BREAK, ctc.CHARS_PER_SEGMENT) == (8000, 15)                      # ctc.py:9-10
    assert ctc.TOKEN_EOS == {'。', '?', '!'} and ctc.TOKEN_COMMA == {'、', ','}
    import reazonspeech.espnet.asr as pub
    assert sorted(n for n in pub.__all__ if n != "transcribe_batch") == sorted(
        ["TranscribeConfig", "transcribe", "load_model", "audio_from_numpy", "audio_from_tensor", "audio_from_path"])
    assert interface.TranscribeConfig().verbose is True
    toks = synthetic_token_list(2600, 0)
    assert toks[0] == "<blank>" and toks[-1] == "<sos/eos>" and len(set(toks)) == 2600 and "。" in toks


def test_split_text_falls_back_to_one_segment_when_alignment_fails():
    """the reference's blanket `except Exception` (ctc.py:90-93): text longer than the audio has frames"""
    model = fk.FakeEspnetModel()
    wav = fk.long_audio(0.2, 9)
    assert ctc.split_text(model, wav, "あ" * 50) == [(0, len(wav), "あ" * 50)]


# ---- the aligner: published algorithm (Kürzinger et al. 2020) ---------------------------------------------------------
def _clean_case(T, events, V=8):
    lpz = np.full((T, V), 1e-4, np.float32)
    lpz[:, 0] = 1.0 - 1e-4 * (V - 1)
    for t, k in events:
        lpz[t] = 1e-4
        lpz[t, k] = 1.0 - 1e-4 * (V - 1)
    return lpz


@pytest.mark.parametrize("as_log", [True, False])
def test_ctc_segmentation_recovers_a_planted_alignment(as_log):
    chars = ["<blank>", "<unk>", "あ", "い", "う", "。", "え", "お"]
    events = [(10, 2), (11, 2), (18, 3), (25, 4), (30, 5), (44, 6), (51, 7)]
    lpz = _clean_case(64, events)
    opt = ctc_segmentation.CtcSegmentationParameters(index_duration=0.5, char_list=chars)
    mat, ind = ctc_segmentation.prepare_text(opt, ["あいう。えお"])
    assert ind == [1, 8] and mat.shape == (9, 7) and mat[0, 0] == -1          # '#' is not a token, '·' maps to the blank
    timings, probs, states = ctc_segmentation.ctc_segmentation(opt, np.log(lpz) if as_log else lpz, mat)
    assert (timings[ind[0] + 1:ind[1]] / 0.5).tolist() == [10, 18, 25, 30, 44, 51]
    assert states[18] == "い" and states[5] == "ε"
    # characters outside the token list and the package's excluded characters do not enter the ground truth
    mat2, ind2 = ctc_segmentation.prepare_text(opt, ["あ,Zい"])
    assert ind2 == [1, 4]
    with pytest.raises(AssertionError):
        ctc_segmentation.ctc_segmentation(opt, lpz[:3], mat)                    # "Audio is shorter than text!"


def test_ctc_segmentation_is_a_max_plus_path():
    """brute force over all monotone alignments of a tiny case: the table's best score is the best path's score"""
    import itertools
    rng = np.random.default_rng(0)
    chars = ["<blank>", "a", "b", "c"]
    T = 7
    lp = np.log(rng.dirichlet(np.ones(4), size=T)).astype(np.float32)
    opt = ctc_segmentation.CtcSegmentationParameters(index_duration=1.0, char_list=chars)
    mat, ind = ctc_segmentation.prepare_text(opt, ["ab"])
    timings = ctc_segmentation.ctc_segmentation(opt, lp, mat)[0]
    seq = [int(mat[c, 0]) for c in range(1, mat.shape[0])]        # '·' a b '·'  ->  token ids (blank, a, b, blank)
    best, best_times = -np.inf, None
    for times in itertools.combinations(range(1, T), len(seq)):   # frame at which each symbol is entered
        score = 0.0
        for j, t0 in enumerate(times):
            t1 = times[j + 1] if j + 1 < len(seq) else times[-1] + 1      # the path ends when the last symbol is entered
            score += float(lp[t0, seq[j]])
            for t in range(t0 + 1, t1):
                score += max(float(lp[t, 0]), float(lp[t, seq[j]]))
        if score > best:
            best, best_times = score, times
    assert [int(x) for x in timings[1:]] == list(best_times)


# ---- weights: ESPnet2 keys, the model directory reader, the prepared tensors -------------------------------------------
def test_espnet_model_directory_round_trip_and_strict_config(tmp_path):
    """an ESPnet2 model directory (training config.yaml, *.pth, feats_stats.npz) written in the zoo's layout is read back
    without ESPnet: same configuration, same tensors, GlobalMVN statistics rebuilt from (count, sum, sum_square); a
    configuration the kernels do not compute is refused"""
    import copy
    import torch
    import yaml
    from reazonspeech_amd.runtime.config import ESPNET_TINY, ESPNET_CONFORMER_120M, UnsupportedCheckpoint
    from reazonspeech_amd.runtime import weights_espnet as we
    cfg = ESPNET_TINY
    sd = we.synthetic_state_dict_espnet(cfg, 3)
    toks = synthetic_token_list(cfg.vocab_size, 3)
    we.write_espnet(str(tmp_path), cfg, sd, toks)
    cfg2, sd2, toks2 = we.read_espnet(str(tmp_path))
    assert cfg2 == cfg and toks2 == toks
    assert all(torch.equal(sd[k], sd2[k]) for k in sd if not k.startswith("normalize."))
    assert (sd["normalize.mean"] - sd2["normalize.mean"]).abs().max() < 1e-5 and (sd["normalize.std"] - sd2["normalize.std"]).abs().max() < 1e-5
    prepared = we.prepare_weights_espnet(cfg2, sd2)
    assert prepared["sub.conv1.w"].shape == (cfg.d_model, 9 * cfg.d_model) and prepared["sub.conv1.w"].dtype == torch.bfloat16
    assert prepared["ctc.w"].shape == (cfg.vocab_size, cfg.d_model) and prepared["joint.pred.b"].abs().max() == 0
    # the dense conv's K order is (kernel row, kernel column, input channel)
    w = sd["encoder.embed.conv.2.weight"]
    assert torch.equal(prepared["sub.conv1.w"][5, (1 * 3 + 2) * cfg.d_model + 7], w[5, 7, 1, 2].to(torch.bfloat16))
    # a leftover tensor or a missing one is an error, not a silent skip
    with pytest.raises(UnsupportedCheckpoint):
        we.prepare_weights_espnet(cfg, {**sd, "encoder.encoders.0.extra.weight": torch.zeros(1)})
    with pytest.raises(UnsupportedCheckpoint):
        we.prepare_weights_espnet(cfg, {k: v for k, v in sd.items() if k != "ctc.ctc_lo.bias"})
    doc = yaml.safe_load(open(os.path.join(str(tmp_path), "exp", "asr_train", "config.yaml")))
    for section, key, value in (("encoder_conf", "input_layer", "conv2d6"), ("encoder_conf", "rel_pos_type", "legacy"),
                                ("encoder_conf", "activation_type", "relu"), ("joint_net_conf", "joint_activation_type", "relu"),
                                ("decoder_conf", "rnn_type", "gru"), ("frontend_conf", "htk", True)):
        bad = copy.deepcopy(doc)
        bad[section][key] = value
        with pytest.raises(UnsupportedCheckpoint):
            we.config_from_espnet_yaml(bad)
    bad = copy.deepcopy(doc)
    bad["normalize"] = "utterance_mvn"
    with pytest.raises(UnsupportedCheckpoint):
        we.config_from_espnet_yaml(bad)
    assert abs(ESPNET_CONFORMER_120M.n_params() - 120e6) < 3e6                              # README.rst:39-40: "120M"
    sd120 = None
    assert ESPNET_CONFORMER_120M.enc_frames(ESPNET_CONFORMER_120M.mel_frames(160000 + 24000)) == 358
