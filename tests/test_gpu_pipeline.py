"""-m gpu: stage-level and end-to-end parity of the HIP path (through the C ABI / the Python
boundary) against the CPU oracle and the committed golden fixtures.

Stated tolerances
  front-end features        max |err| <= 2e-3 (normalised log-mel, O(1) values)
  encoder output (bf16 recipe, TINY 2-layer) max |err| <= 6e-2, mean |err| <= 6e-3 on LayerNorm-ed
                            O(1) activations; vs the fp32 HF golden: max <= 0.15
  joint encoder projection  same class
  greedy decode             token ids and emission frames BIT-EXACT against oracle/rnnt_greedy.c
                            when both consume the same joint-encoder tensor
"""
import os

import numpy as np
import pytest
import torch

from reazonspeech_amd.nemo.asr import (load_model, transcribe, transcribe_batch, audio_from_numpy,
                                       TranscribeConfig)
from reazonspeech_amd.runtime.config import TINY
from reazonspeech_amd.runtime.model import AsrModel
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
from reazonspeech_amd.runtime.weights import synthetic_state_dict
from oracle import model as om, greedy as og

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parakeet_tiny.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def tiny(gold, gpu_device):
    sd = synthetic_state_dict(TINY, int(gold["seed"]), blank_bias=float(gold["blank_bias"]))
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0", pad_seconds=0.0)
    return model, sd


def _run_stages(model, audio, lens, want_enc=True):
    waves = [audio[b, :int(lens[b])] for b in range(audio.shape[0])]
    buf = model.stage(waves)
    enc = torch.zeros((buf.B, buf.tp_max, model.cfg.d_model), dtype=torch.float32, device=model.device) \
        if want_enc else None
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    return buf, enc


def test_frontend_matches_oracle_and_hf(tiny, gold):
    model, sd = tiny
    audio, lens = gold["audio"], gold["lengths"]
    buf, _ = _run_stages(model, audio, lens, want_enc=False)
    feats = buf.feats.cpu()
    n = buf.n_frames.cpu().numpy()
    ref, n_ref = om.frontend(TINY, sd, torch.from_numpy(audio), torch.from_numpy(lens))
    assert n.tolist() == n_ref.tolist() == gold["hf_n_frames"].tolist()
    T = ref.shape[1]
    assert (feats[:, :T] - ref).abs().max() <= 2e-3
    assert (feats[:, :T] - torch.from_numpy(gold["hf_feats"])[:, :T]).abs().max() <= 2e-3
    for b in range(len(n)):
        assert torch.all(feats[b, n[b]:] == 0)


def test_frontend_pad_is_folded(gold, gpu_device):
    """pad_audio (audio.py:70-83) folded into the kernel == padding on the host first"""
    sd = synthetic_state_dict(TINY, 3)
    m_fold = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0", pad_seconds=0.5)
    m_none = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0", pad_seconds=0.0)
    audio, lens = synthetic_batch(3, 1.3, seed=5, ragged=True, min_seconds=0.4)
    waves = [audio[b, :lens[b]] for b in range(3)]
    b1 = m_fold.stage(waves)
    m_fold.run_device(b1)
    b2 = m_none.stage([np.pad(w, 8000) for w in waves])
    m_none.run_device(b2)
    torch.cuda.synchronize()
    assert b1.n_frames.cpu().tolist() == b2.n_frames.cpu().tolist()
    T = min(b1.t_max, b2.t_max)
    assert torch.equal(b1.feats.cpu()[:, :T], b2.feats.cpu()[:, :T])
    assert b1.n_ids.cpu().tolist() == b2.n_ids.cpu().tolist()


@pytest.mark.parametrize("fuse_glu", [1, 0])
def test_encoder_matches_oracle(tiny, gold, fuse_glu):
    """fuse_glu = 1 (the default at every batch size): the conv module's GLU in the pw1 GEMM epilogue, against the
    oracle recipe with that rounding point; 0: plain pw1 product, GLU in the depthwise kernel (the other layout)"""
    model, sd = tiny
    audio, lens = gold["audio"], gold["lengths"]
    model.ctx.set_option("fuse_glu", fuse_glu)
    try:
        buf, enc = _run_stages(model, audio, lens)
    finally:
        model.ctx.set_option("fuse_glu", 1)
    taps = {}
    f_ref, el = om.forward_to_joint(TINY, sd, torch.from_numpy(audio), torch.from_numpy(lens),
                                    "bf16-fused-glu" if fuse_glu == 1 else "bf16", taps)
    assert buf.enc_lens.cpu().tolist() == el.tolist() == gold["hf_enc_lens"].tolist()
    enc, f = enc.cpu(), buf.joint_enc.cpu()
    hf = torch.from_numpy(gold["hf_enc"])
    for b in range(len(el)):
        n = int(el[b])
        d = (enc[b, :n] - taps["enc"][b, :n]).abs()
        assert d.max() <= 6e-2 and d.mean() <= 6e-3, (b, d.max().item(), d.mean().item())
        dj = (f[b, :n] - f_ref[b, :n]).abs()
        assert dj.max() <= 6e-2, (b, dj.max().item())
        assert (enc[b, :n] - hf[b, :n]).abs().max() <= 0.15


def test_decode_bit_exact_given_same_encoder_output(tiny, gold):
    model, sd = tiny
    audio, lens = gold["audio"], gold["lengths"]
    buf, _ = _run_stages(model, audio, lens, want_enc=False)
    got = model.collect(buf)
    ref = og.rnnt_greedy(TINY, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    assert got.ids == [r[0] for r in ref]
    assert got.frames == [r[1] for r in ref]


@pytest.mark.parametrize("lookahead,B", [(True, 37), (False, 37), (True, 3), (True, 150)])
def test_decode_bit_exact_many_utterances(gpu_device, monkeypatch, lookahead, B):
    """synthetic joint-encoder tensors straight into rs_rnnt_greedy: ragged lengths, an empty
    utterance, a batch that is not a multiple of the 16-row tile.  Small batches score several frames per utterance and step
    (look-ahead: 37 rows -> 4 frames, 3 rows -> 8, 150 rows -> 1 until the tail); $RS_DECODE_NO_LOOKAHEAD is the plain loop."""
    if not lookahead:
        monkeypatch.setenv("RS_DECODE_NO_LOOKAHEAD", "1")
    cfg = TINY
    sd = synthetic_state_dict(cfg, 11, blank_bias=4.0)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    g = torch.Generator().manual_seed(2)
    Tp = 45
    f = torch.randn((B, Tp, cfg.joint_hidden), generator=g) * 1.5
    lens = torch.randint(1, Tp + 1, (B,), generator=g, dtype=torch.int32)
    lens[B // 2] = 0
    lens[B - 1] = Tp
    u_max = Tp * cfg.max_symbols
    dev = model.device
    ids = torch.zeros((B, u_max), dtype=torch.int32, device=dev)
    frames = torch.zeros_like(ids)
    n_ids = torch.zeros((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((model.ctx.workspace_bytes(B, 16000),), dtype=torch.uint8, device=dev)
    model.ctx.rnnt_greedy(f.to(dev), lens.to(dev), B, Tp, u_max, ids, frames, n_ids, ws,
                          torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = og.rnnt_greedy(cfg, sd, f.numpy(), lens.numpy())
    n = n_ids.cpu().numpy()
    assert sum(len(r[0]) for r in ref) > B, "test inputs should emit tokens"
    for b in range(B):
        assert ids[b, :n[b]].cpu().tolist() == ref[b][0], b
        assert frames[b, :n[b]].cpu().tolist() == ref[b][1], b


def test_end_to_end_ids_vs_oracle(tiny, gold):
    """whole path: ids and emission frames of the HIP path (throughput mode) == the oracle run end to end in the same bf16
    recipe (GLU in the pw1 epilogue) == the HF parakeet golden, EXACTLY, on this fixture.  (The float32 parity mode makes
    the same statement against the float32 oracle at every geometry: tests/test_gpu_fp32_mode.py.)  The flip audit runs on
    top with an ABSOLUTE cap on the joint-projection difference it may use as an excuse (the tolerance of the encoder
    parity test above), so an encoder regression cannot widen its own bound."""
    from oracle import audit
    model, sd = tiny
    audio, lens = gold["audio"], gold["lengths"]
    buf, _ = _run_stages(model, audio, lens, want_enc=False)
    got = model.collect(buf)
    f_hip = buf.joint_enc.cpu().numpy()
    f_ref, el = om.forward_to_joint(TINY, sd, torch.from_numpy(audio), torch.from_numpy(lens), "bf16-fused-glu")
    ref = og.rnnt_greedy(TINY, sd, f_ref.numpy(), el.numpy())
    assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref]
    hf = [[int(x) for x in gold["hf_ids"][b, :gold["hf_n_ids"][b]]] for b in range(2)]
    assert got.ids == hf, "HIP ids drifted from the HF golden"
    audits = [audit.flip_audit(TINY, sd, f_ref[b].numpy(), f_hip[b], int(el[b]), got.ids[b], got.frames[b]) for b in range(2)]
    s = audit.summarize(audits, [True, True])
    assert s["walk_reproduces_hip_path"] and s["local_flips"] == 0, s
    for b in range(2):
        n = int(el[b])
        assert np.abs(f_ref[b, :n].numpy() - f_hip[b, :n]).max() <= 6e-2          # the audit's absolute cap on delta f


def edit_d(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def test_batch_invariance(gpu_device):
    """an utterance decoded alone == the same utterance inside a ragged batch (reference
    semantics: batch_size=1, transcribe.py:48-50)"""
    sd = synthetic_state_dict(TINY, 21, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(5, 3.0, seed=9, ragged=True, min_seconds=0.5)
    waves = [audio[b, :lens[b]] for b in range(5)]
    together = model.transcribe_waveforms(waves)
    for b in range(5):
        alone = model.transcribe_waveforms([waves[b]])
        assert alone.ids[0] == together.ids[b]
        assert alone.frames[0] == together.frames[b]


def test_pipelined_schedule_equals_sequential(gpu_device):
    """encoder(i+1) || decode(i) on two streams gives exactly what the one-stream path gives"""
    sd = synthetic_state_dict(TINY, 31, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    bufs, want = [], []
    for k in range(2):
        audio, lens = synthetic_batch(6, 2.5, seed=40 + k, ragged=True, min_seconds=0.5)
        waves = [audio[b, :lens[b]] for b in range(6)]
        ref = model.transcribe_waveforms(waves)
        want.append((ref.ids, ref.frames))
        bufs.append(model.stage(waves, buf=model.new_buffers(6, 40000)))
    got = {}

    def grab(buf):
        torch.cuda.current_stream().synchronize()
        got.setdefault(id(buf), []).append(model.collect(buf))

    model.run_pipelined(bufs, 5, after_decode=grab)
    for k in range(2):
        runs = got[id(bufs[k])]
        assert len(runs) == (3 if k == 0 else 2)
        for r in runs:
            assert (r.ids, r.frames) == want[k]


def test_two_decode_streams_equal_sequential(gpu_device):
    """decode of consecutive batches on two streams (three resident batches): same hypotheses as the one-stream
    path, and the after-decode hooks fire in batch order whichever lane finishes first"""
    sd = synthetic_state_dict(TINY, 33, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    bufs, want = [], []
    for k in range(3):
        audio, lens = synthetic_batch(5, 3.0 if k != 1 else 0.9, seed=60 + k, ragged=True, min_seconds=0.5)   # batch 1 decodes fastest
        waves = [audio[b, :lens[b]] for b in range(5)]
        ref = model.transcribe_waveforms(waves)
        want.append((ref.ids, ref.frames))
        bufs.append(model.stage(waves, buf=model.new_buffers(5, 48000)))
    order = []

    def grab(buf):
        torch.cuda.current_stream().synchronize()
        k = [id(b) for b in bufs].index(id(buf))
        r = model.collect(buf)
        order.append((k, (r.ids, r.frames) == want[k]))

    model.run_pipelined(bufs, 7, after_decode=grab, dec_streams=2)
    assert [k for k, _ in order] == [i % 3 for i in range(7)]
    assert all(ok for _, ok in order)


def test_three_decode_lanes_equal_sequential(gpu_device):
    """n decode lanes (batch i -> lane i mod n; n + 1 resident batches): same hypotheses as the one-stream path, hooks in
    batch order"""
    sd = synthetic_state_dict(TINY, 34, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    bufs, want = [], []
    for k in range(4):
        audio, lens = synthetic_batch(4, 2.5 if k != 2 else 0.8, seed=80 + k, ragged=True, min_seconds=0.5)
        waves = [audio[b, :lens[b]] for b in range(4)]
        ref = model.transcribe_waveforms(waves)
        want.append((ref.ids, ref.frames))
        bufs.append(model.stage(waves, buf=model.new_buffers(4, 40000)))
    order = []

    def grab(buf):
        torch.cuda.current_stream().synchronize()
        k = [id(b) for b in bufs].index(id(buf))
        r = model.collect(buf)
        order.append((buf.step, k, (r.ids, r.frames) == want[k]))

    model.run_pipelined(bufs, 9, after_decode=grab, dec_streams=3)
    assert [s for s, _, _ in order] == list(range(9)) and [k for _, k, _ in order] == [i % 4 for i in range(9)]
    assert all(ok for _, _, ok in order)


def test_long_list_is_chunked_sorted_and_pipelined(gpu_device):
    """more utterances than max_batch: sorted by length, batches of max_batch through the pipeline,
    results in the caller's order and identical to one-at-a-time decoding"""
    sd = synthetic_state_dict(TINY, 33, blank_bias=4.0)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(11, 2.0, seed=77, ragged=True, min_seconds=0.3)
    waves = [audio[b, :lens[b]] for b in range(11)]
    got = model.transcribe_waveforms(waves, max_batch=4)        # 3 batches: 4 + 4 + 3 (padded with empties)
    for b in (0, 3, 7, 10):
        alone = model.transcribe_waveforms([waves[b]])
        assert got.ids[b] == alone.ids[0] and got.frames[b] == alone.frames[0]
    assert len(got.ids) == 11 and all(x is not None for x in got.ids)


def test_long_form_audio_crosses_key_chunks(gpu_device):
    """25 s of audio: T' = 326 > 160 keys, so attention runs several staged key chunks and two
    workgroups of query blocks; checked against the oracle in the bf16 recipe"""
    sd = synthetic_state_dict(TINY, 35, blank_bias=4.3)
    model = AsrModel(TINY, sd, SyntheticTokenizer(TINY.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(2, 25.0, seed=5, ragged=True, min_seconds=12.0)
    waves = [audio[b, :lens[b]] for b in range(2)]
    buf = model.stage(waves)
    enc = torch.zeros((2, buf.tp_max, TINY.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    padded = np.zeros((2, audio.shape[1] + 16000), np.float32)
    for b in range(2):
        padded[b, 8000:8000 + lens[b]] = waves[b]
    taps = {}
    f_ref, el = om.forward_to_joint(TINY, sd, torch.from_numpy(padded), torch.from_numpy(lens + 16000), "bf16", taps)
    assert buf.enc_lens.cpu().tolist() == el.tolist() and int(el.max()) > 2 * 160 - 64   # > 160 keys: several staged key chunks
    for b in range(2):
        n = int(el[b])
        dlt = (enc.cpu()[b, :n] - taps["enc"][b, :n]).abs()
        assert dlt.max() <= 8e-2 and dlt.mean() <= 8e-3, (dlt.max().item(), dlt.mean().item())
    ref = og.rnnt_greedy(TINY, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    got = model.collect(buf)
    assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref]


def test_long_form_ten_minutes_limited_context(gpu_device):
    """One 10-minute utterance through the whole path (the reference passes a whole file as one utterance,
    pkg/nemo-asr/src/transcribe.py:44-53): T' = 7 507 frames — far past the 1024 rows the position tables start
    with, so they grow on first use — with the limited-context attention variant ([128, 128] + one global token:
    only ~9 of 235 key blocks are visited per query block).  Checked against the bf16-recipe oracle (which
    materialises the full T' x T' score matrix with the same mask) and decoded bit-exactly."""
    cfg = TINY.with_(att_left=128, att_right=128, n_global=1)
    sd = synthetic_state_dict(cfg, 41, blank_bias=3.5)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(1, 600.0, seed=11)
    assert model.pos_cap == 1024
    buf = model.stage([audio[0]])
    assert model.pos_cap >= buf.tp_max > 7000
    enc = torch.zeros((1, buf.tp_max, cfg.d_model), dtype=torch.float32, device=model.device)
    model.run_device(buf, want_enc=enc)
    torch.cuda.synchronize()
    padded = np.zeros((1, audio.shape[1] + 16000), np.float32)
    padded[0, 8000:8000 + lens[0]] = audio[0]
    taps = {}
    f_ref, el = om.forward_to_joint(cfg, sd, torch.from_numpy(padded), torch.from_numpy(lens + 16000), "bf16", taps)
    assert buf.enc_lens.cpu().tolist() == el.tolist()
    n = int(el[0])
    dlt = (enc.cpu()[0, :n] - taps["enc"][0, :n]).abs()
    assert dlt.max() <= 8e-2 and dlt.mean() <= 8e-3, (dlt.max().item(), dlt.mean().item())
    got = model.collect(buf)
    ref = og.rnnt_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), buf.enc_lens.cpu().numpy())
    assert len(got.ids[0]) > 100
    assert got.ids == [r[0] for r in ref] and got.frames == [r[1] for r in ref]
    # a short clip afterwards still works on the grown tables
    short = model.transcribe_waveforms([audio[0][:32000]])
    assert short.ids[0] == model.transcribe_waveforms([audio[0][:32000]]).ids[0]


def test_python_boundary(tiny):
    model, _ = tiny
    audio, lens = synthetic_batch(2, 1.0, seed=1)
    a0 = audio_from_numpy(audio[0], 16000)
    r = transcribe(model, a0)
    assert isinstance(r.text, str) and r.hypothesis is None
    r2 = transcribe(model, a0, TranscribeConfig(raw_hypothesis=True))
    assert r2.hypothesis is not None and r2.hypothesis.y_sequence[0] == TINY.blank_id
    rs = transcribe_batch(model, [a0, audio_from_numpy(np.stack([audio[1], audio[1]]), 16000)])
    assert rs[0].text == r.text
    assert transcribe_batch(model, []) == []
    # 8 kHz input is resampled on the host
    r8 = transcribe(model, audio_from_numpy(audio[0][::2].copy(), 8000))
    assert isinstance(r8.text, str)


def test_evaluation_batch_hook(tiny):
    """the batched evaluation hook == per-example transcribe()"""
    from reazonspeech_amd.evaluation import RSAmdEvaluator
    model, _ = tiny
    audio, lens = synthetic_batch(5, 1.5, seed=3, ragged=True, min_seconds=0.4)
    rows = [{"audio": {"array": audio[b, :lens[b]], "sampling_rate": 16000}, "text": "あ"} for b in range(5)]
    ev = RSAmdEvaluator(model=model, batch_size=3)
    out = ev.evaluate(rows)
    assert [r["prediction"] for r in out] == [transcribe(model, audio_from_numpy(audio[b, :lens[b]], 16000)).text
                                              for b in range(5)]
    assert all(r["length"] == 1 for r in out)


def test_load_model_rejects_cpu():
    with pytest.raises(RuntimeError):
        load_model(device="cpu")


def test_full_size_model_properties(gpu_device):
    """The 619M configuration (d = 1024, 24 layers, 8 heads) through the whole path: kernel instantiations that
    TINY never selects (LayerNorm<4>, the fused norm pair, 8-head attention, 4096-wide FFN GEMMs).  Asserted are
    size-independent properties only — run-to-run determinism, alone == inside a ragged batch, pipelined ==
    sequential; the CPU oracle of this size is timed by bench.py, not compared here."""
    from reazonspeech_amd.runtime.config import FASTCONFORMER_619M as CFG
    sd = synthetic_state_dict(CFG, 0)
    model = AsrModel(CFG, sd, SyntheticTokenizer(CFG.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(6, 3.0, seed=77, ragged=True, min_seconds=1.0)
    waves = [audio[b, :lens[b]] for b in range(6)]
    first = model.transcribe_waveforms(waves)
    again = model.transcribe_waveforms(waves)
    assert first.ids == again.ids and first.frames == again.frames
    assert sum(len(x) for x in first.ids) > 0, "degenerate fixture: nothing was emitted"
    for b in (0, 3):
        alone = model.transcribe_waveforms([waves[b]])
        assert alone.ids[0] == first.ids[b] and alone.frames[0] == first.frames[b]
    bufs = [model.stage(waves, buf=model.new_buffers(6, 48000)) for _ in range(2)]
    got = []

    def grab(buf):
        torch.cuda.current_stream().synchronize()
        got.append(model.collect(buf))

    model.run_pipelined(bufs, 3, after_decode=grab)
    assert len(got) == 3
    for r in got:
        assert (r.ids, r.frames) == (first.ids, first.frames)


@pytest.mark.parametrize("tile", [0, 256, 192, 128, 64])
@pytest.mark.parametrize("option", ["defer_out_norm"])
def test_fused_epilogues_are_bit_identical(gpu_device, option, tile):
    """A fusion that moves work into a GEMM epilogue without changing a bit of the result, checked against its unfused
    form at every GEMM tile height on a ragged batch and on one utterance: `defer_out_norm` — a layer's output LayerNorm
    is applied by the next layer's first residual GEMM from per-row (mean, rstd) instead of being stored.  (The same test
    held `fuse_dwconv` — depthwise conv + SiLU in the pw1 epilogue — bit-identical too; that fusion was measured and
    removed: profiles/r03t_*.)"""
    import ctypes
    from reazonspeech_amd.runtime.config import FASTCONFORMER_619M
    cfg = FASTCONFORMER_619M.with_(n_layers=3)
    sd = synthetic_state_dict(cfg, 5)
    model = AsrModel(cfg, sd, SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
    audio, lens = synthetic_batch(7, 4.0, seed=99, ragged=True, min_seconds=1.0)
    lib = model.ctx.lib
    lib.rs_debug_set_gemm_tile.argtypes = [ctypes.c_int]
    lib.rs_debug_set_gemm_tile.restype = None
    for waves in ([audio[b, :lens[b]] for b in range(7)], [audio[2, :lens[2]]]):
        outs = []
        try:
            lib.rs_debug_set_gemm_tile(tile)
            for on in (1, 0):
                model.ctx.set_option(option, on)
                buf = model.stage(waves)
                enc = torch.zeros((len(waves), buf.tp_max, cfg.d_model), dtype=torch.float32, device=model.device)
                model.run_device(buf, want_enc=enc)
                torch.cuda.synchronize()
                res = model.collect(buf)
                outs.append((enc.clone(), buf.joint_enc.clone(), res.ids, res.frames))
        finally:
            lib.rs_debug_set_gemm_tile(0)
            model.ctx.set_option(option, 1)
        assert torch.isfinite(outs[0][0]).all() and outs[0][0].abs().max() > 0
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert outs[0][2] == outs[1][2] and outs[0][3] == outs[1][3]


def test_hypothesis_exchange_through_rccl_single_rank(gpu_device):
    """The multi-GPU exchange on the hardware that is available: a `nccl` (= RCCL) process group of one rank, with the
    collectives of runtime/dist.py forced on — the MAX all_reduce that agrees on the payload width and the one
    all_gather_into_tensor of (count | encoder length | score bits | ids | frames) run through RCCL on device tensors and
    give back exactly what the local decode produced, for the greedy and the beam-search payload (scores bit for bit).
    (World sizes 2 and 8 are covered on CPU with gloo: tests/test_host_runtime.py.)"""
    import socket
    import torch.distributed as dist
    from reazonspeech_amd.runtime import dist as rdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    assert not dist.is_initialized()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        rdist.use_collectives_with_one_rank(True)
        assert rdist.is_on() and dist.get_backend() == "nccl"
        audio, lens = synthetic_batch(9, 3.0, seed=5, ragged=True, min_seconds=0.5)
        waves = [audio[b, :lens[b]] for b in range(9)]
        for decoding in ("greedy_batch", "alsd"):
            cfg = TINY.with_(decoding=decoding, beam_size=3)
            model = AsrModel(cfg, synthetic_state_dict(cfg, 2), SyntheticTokenizer(cfg.vocab_size), device="cuda:0")
            want = model.transcribe_waveforms(waves)
            got = model.transcribe_waveforms_sharded(waves)
            assert got.ids == want.ids and got.frames == want.frames and list(got.enc_lens) == list(want.enc_lens)
            if decoding == "alsd":
                assert np.array_equal(np.asarray(got.scores, dtype=np.float32).view(np.uint32),
                                      np.asarray(want.scores, dtype=np.float32).view(np.uint32))
            assert sum(len(x) for x in got.ids) > 0
        assert rdist.max_over_ranks(1.25) == 1.25
        rdist.barrier()
    finally:
        rdist.use_collectives_with_one_rank(False)
        dist.destroy_process_group()
