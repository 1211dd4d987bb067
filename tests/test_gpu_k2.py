"""-m gpu: the Zipformer2 transducer of `reazonspeech.k2.asr` (BASELINE.json configs[3]) through the C ABI
against its CPU oracle (oracle/zipformer.py — PARITY UNPINNED against icefall / sherpa-onnx, which cannot run here).

Stated tolerances (bf16 GEMM operands and stored activations, f32 accumulation and residual stream; the oracle's "bf16" recipe
rounds at the same points):
  kaldi-style fbank features                           max |err| <= 5e-3 on log energies (float32 FFT; exact at the floor)
  encoder_embed output, every stack's output           toy and 159M geometry: max <= 0.08, mean <= 0.01 (activations are O(1): BiasNorm)
  encoder output, joiner.encoder_proj                  max <= 0.08, mean <= 0.01
  greedy search (stateless decoder, tanh joiner)       ids and frames BIT-EXACT vs oracle/k2_greedy.c on the same projection
  batch invariance                                     an utterance alone == the same utterance inside a ragged batch, bit for bit
"""
import numpy as np
import pytest
import torch

from reazonspeech_amd.runtime.k2_config import ZIPFORMER_TINY, ZIPFORMER_159M
from reazonspeech_amd.runtime.k2_weights import synthetic_state_dict_k2
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.k2.asr.model import K2Model, synthetic_tokens
from reazonspeech_amd.k2.asr import interface
from oracle import zipformer as oz, greedy as og
import importlib

k2tr = importlib.import_module("reazonspeech_amd.k2.asr.transcribe")

pytestmark = pytest.mark.gpu
PAD = int(0.9 * 16000)


def build(cfg, seed):
    sd = synthetic_state_dict_k2(cfg, seed)
    return K2Model(cfg, sd, synthetic_tokens(cfg.vocab_size, seed), device="cuda:0"), sd


@pytest.fixture(scope="module")
def tiny(gpu_device):
    return build(ZIPFORMER_TINY, 3)


def run(model, waves, taps=True):
    am, cfg = model.am, model.cfg
    buf = am.stage(waves, buf=am.new_buffers(len(waves), max(len(w) for w in waves)))
    B = buf.B
    t3 = cfg.embed_frames(buf.t_max)
    emb = torch.zeros((B, t3, cfg.encoder_dim[0]), dtype=torch.float32, device=am.device)
    stacks = torch.zeros((B * t3 * sum(cfg.encoder_dim),), dtype=torch.float32, device=am.device)
    enc = torch.zeros((B, buf.tp_max, cfg.out_dim), dtype=torch.float32, device=am.device)
    if taps:
        am.ctx.set_k2_taps(emb, stacks)
    try:
        am.run_device(buf, want_enc=enc)
        torch.cuda.synchronize()
    finally:
        am.ctx.set_k2_taps(None, None)
    outs, off = [], 0
    for d in cfg.encoder_dim:
        outs.append(stacks[off:off + B * t3 * d].view(B, t3, d).cpu())
        off += B * t3 * d
    return buf, emb.cpu(), outs, enc.cpu(), am.collect(buf)


def compare(cfg, sd, model, waves, tol_max=0.08, tol_mean=0.01):
    buf, emb, stacks, enc, got = run(model, waves)
    feats = buf.feats.cpu()
    stats = {}
    for b, w in enumerate(waves):
        taps = {}
        ref = oz.forward(cfg, sd, w, "bf16", taps)
        nf = ref["feats"].shape[0]
        assert int(buf.n_frames[b]) == nf
        d = (feats[b, :nf] - ref["feats"]).abs().max().item()
        stats["feats"] = max(stats.get("feats", 0.0), d)
        assert d <= 5e-3, (b, d)
        assert torch.all(feats[b, nf:] == 0)
        t3 = cfg.embed_frames(nf)
        n = ref["enc"].shape[0]
        assert got.enc_lens[b] == n
        pairs = [("embed", emb[b, :t3], taps["embed"])] + [(f"S{s}", stacks[s][b, :t3], taps[f"S{s}"]) for s in range(cfg.n_stacks)]
        pairs += [("enc", enc[b, :n], ref["enc"]), ("joint", buf.joint_enc[b, :n].cpu(), ref["joint_enc"])]
        for name, a, r in pairs:
            e = (a - r).abs()
            stats[name] = max(stats.get(name, 0.0), e.max().item())
            assert e.max() <= tol_max and e.mean() <= tol_mean, (name, b, e.max().item(), e.mean().item())
    same = og.k2_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), np.asarray(got.enc_lens, np.int32))
    assert got.ids == [r[0] for r in same] and got.frames == [r[1] for r in same]
    assert all(len(set(f)) == len(f) for f in got.frames), "one symbol per frame"
    assert all(cfg.unk_id not in ids and cfg.blank_id not in ids for ids in got.ids)
    return stats, got


def ragged_waves(n, seconds, seed, min_seconds):
    audio, lens = synthetic_batch(n, seconds, seed=seed, ragged=True, min_seconds=min_seconds)
    return [np.pad(audio[b, :lens[b]], PAD) for b in range(n)]


def test_tiny_pipeline_vs_oracle(tiny):
    model, sd = tiny
    stats, got = compare(ZIPFORMER_TINY, sd, model, ragged_waves(4, 3.0, 5, 0.7))
    assert sum(len(x) for x in got.ids) > 10
    print("k2 tiny:", stats, [len(x) for x in got.ids])


def test_tiny_batch_invariance_bits(tiny):
    model, sd = tiny
    waves = ragged_waves(5, 3.0, 9, 0.5)
    buf, emb, stacks, enc, together = run(model, waves)
    f_all = buf.joint_enc.cpu()
    for b in (0, 3):
        b1, _, _, e1, alone = run(model, waves[b:b + 1])
        n = alone.enc_lens[0]
        assert n == together.enc_lens[b]
        assert torch.equal(e1[0, :n], enc[b, :n]) and torch.equal(b1.joint_enc[0, :n].cpu(), f_all[b, :n])
        assert alone.ids[0] == together.ids[b] and alone.frames[0] == together.frames[b]


def test_unk_is_never_emitted_and_costs_no_context(tiny):
    """[UPSTREAM] sherpa-onnx's greedy search skips `<unk>` like the blank: with a joiner biased towards <unk> the device and the
    C checker still agree bit for bit and no <unk> appears"""
    cfg = ZIPFORMER_TINY
    sd = synthetic_state_dict_k2(cfg, 4)
    sd["joiner.output_linear.bias"][cfg.unk_id] += 9.0
    model = K2Model(cfg, sd, synthetic_tokens(cfg.vocab_size, 4), device="cuda:0")
    waves = ragged_waves(3, 2.0, 11, 0.8)
    buf, _, _, _, got = run(model, waves, taps=False)
    same = og.k2_greedy(cfg, sd, buf.joint_enc.cpu().numpy(), np.asarray(got.enc_lens, np.int32))
    assert got.ids == [r[0] for r in same] and got.frames == [r[1] for r in same]
    assert all(cfg.unk_id not in ids for ids in got.ids)


def decode_with(model, buf, screen):
    """the greedy search alone on the buffer's joint projection with the screened (bf16 screening product + exact re-evaluation of
    the candidates) or the all-exact joint"""
    am = model.am
    am.ctx.set_option("decode_screen", screen)
    am.ctx.set_option("decode_narrow", 1)
    am.decode(am.ctx, buf, buf.ws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return am.collect(buf)


def test_screened_joint_is_bit_identical_with_the_tanh_joiner(tiny):
    model, sd = tiny
    waves = ragged_waves(6, 3.0, 13, 0.5)
    buf, _, _, _, _ = run(model, waves, taps=False)
    exact, screened = decode_with(model, buf, 0), decode_with(model, buf, 1)
    want = og.k2_greedy(ZIPFORMER_TINY, sd, buf.joint_enc.cpu().numpy(), np.asarray(exact.enc_lens, np.int32))
    assert exact.ids == screened.ids == [r[0] for r in want] and exact.frames == screened.frames == [r[1] for r in want]
    assert sum(len(x) for x in exact.ids) > 10


def test_159m_geometry_vs_oracle(gpu_device):
    """the published shape (6 stacks 192 .. 768 wide, 19 layers, down-sampling 1 / 2 / 4 / 8 / 4 / 2, kernels 31 / 15): two ragged
    utterances with the reference's 0.9 s padding"""
    cfg = ZIPFORMER_159M
    model, sd = build(cfg, 0)
    waves = ragged_waves(2, 3.0, 123, 1.5)
    stats, got = compare(cfg, sd, model, waves)
    print("k2 159M:", stats, [len(x) for x in got.ids])
    # the screened joint over the 10 720-symbol vocabulary (candidates scanned from memory, batches of 1024): same ids and frames
    buf, _, _, _, _ = run(model, waves, taps=False)
    exact, screened = decode_with(model, buf, 0), decode_with(model, buf, 1)
    assert exact.ids == screened.ids == got.ids and exact.frames == screened.frames == got.frames


def test_convnext_pointwise_pair_as_one_kernel_is_bit_identical(gpu_device):
    """csrc/k_zipformer.hip k2_cnx_pw_fused_kernel (weights register-resident over a persistent workgroup's waves, the hidden tile in
    LDS) against the two GEMM launches it replaces, and k2_conv2_fused_kernel (3 x 3 x 32 patches gathered into LDS) against the patch
    matrix + GEMM launch, at the published geometry (embed channels 32 / 128): the encoder projection of ragged utterances, bits"""
    model, sd = build(ZIPFORMER_159M, 0)
    waves = ragged_waves(5, 4.0, 31, 0.7)
    outs = {}
    for form in ((1, 1), (0, 1), (1, 0), (0, 0)):           # (ConvNeXt pair fused, conv2 fused)
        model.am.ctx.set_option("k2_cnx_fused", form[0])
        model.am.ctx.set_option("k2_conv2_fused", form[1])
        buf, _, _, _, _ = run(model, waves, taps=False)
        outs[form] = buf.joint_enc.clone()
    model.am.ctx.set_option("k2_cnx_fused", 1)
    model.am.ctx.set_option("k2_conv2_fused", 1)
    assert all(torch.equal(outs[f], outs[(0, 0)]) for f in outs)
    assert float(outs[(1, 1)].abs().max()) > 0.1


def test_model_object_answers_sherpa_onnx_call_forms(tiny):
    """create_stream / accept_waveform / decode_stream / result.{tokens, timestamps, text} (pkg/k2-asr/src/transcribe.py:36-45) and
    the package's transcribe(): padding of 0.9 s on both sides, timestamps = frame x 0.04 s, text = the tokens joined"""
    model, sd = tiny
    wav = synthetic_batch(1, 4.0, seed=21)[0][0]
    res = k2tr.transcribe(model, interface.AudioData(wav, 16000), interface.TranscribeConfig(verbose=False))
    st = model.create_stream()
    st.accept_waveform(16000, np.pad(wav, PAD))
    model.decode_stream(st)
    assert [s.token for s in res.subwords] == st.result.tokens and [s.seconds for s in res.subwords] == st.result.timestamps
    assert res.text == st.result.text == "".join(st.result.tokens)
    assert all(abs(t / 0.04 - round(t / 0.04)) < 1e-4 for t in st.result.timestamps)
    assert st.result.timestamps == sorted(st.result.timestamps)
    two = k2tr.transcribe_batch(model, [interface.AudioData(wav, 16000), interface.AudioData(wav[:30000], 16000)])
    assert two[0].text == res.text and [s.seconds for s in two[0].subwords] == [s.seconds for s in res.subwords]
    with pytest.warns(UserWarning, match="long audio input"):
        k2tr._prepare(interface.AudioData(np.zeros(16000 * 29, np.float32), 16000))
