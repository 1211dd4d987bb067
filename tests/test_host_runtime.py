"""CPU: host logic of the runtime — model dims, synthetic weights, weight prep layouts, .nemo
round trip, tokenizer, synthetic inputs, sharding helpers, world_size-2 gloo gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from reazonspeech_amd.runtime import dist as rdist
from reazonspeech_amd.runtime.config import FASTCONFORMER_619M, TINY, from_nemo_yaml
from reazonspeech_amd.runtime.synth import synthetic_batch
from reazonspeech_amd.runtime.tokenizer import SyntheticTokenizer
from reazonspeech_amd.runtime import weights as W


def test_config_geometry():
    cfg = FASTCONFORMER_619M
    assert round(cfg.n_params() / 1e6, 1) == 619.2            # README.rst:34-35 "619M"
    assert cfg.mel_frames(176000) == 1100 and cfg.stft_frames(176000) == 1101
    assert cfg.enc_frames(1100) == 138 and cfg.enc_frames(900) == 113
    assert cfg.sub_freq == 10 and cfg.head_dim == 128 and cfg.blank_id == 3000
    assert abs(cfg.hop_length * cfg.sub_factor / cfg.sample_rate - 0.08) < 1e-12   # decode.py:5


def test_synthetic_state_dict_is_deterministic_and_shaped():
    a = W.synthetic_state_dict(TINY, 3)
    b = W.synthetic_state_dict(TINY, 3)
    c = W.synthetic_state_dict(TINY, 4)
    assert a.keys() == b.keys()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["joint.enc.weight"], c["joint.enc.weight"])
    n = sum(v.numel() for k, v in a.items()
            if not k.startswith("preprocessor") and "running" not in k and "num_batches" not in k)
    assert n == TINY.n_params()
    assert torch.all(a["decoder.prediction.embed.weight"][TINY.blank_id] == 0)      # blank_as_pad
    assert a["encoder.layers.1.self_attn.pos_bias_u"].shape == (TINY.n_heads, 128)


def test_prepare_weights_layouts():
    cfg = TINY
    sd = W.synthetic_state_dict(cfg, 1)
    p = W.prepare_weights(cfg, sd, pos_cap=40)
    C, d, F = cfg.sub_channels, cfg.d_model, cfg.sub_freq
    assert p["sub.conv0.w"].shape == (9, C) and p["sub.dw1.w"].dtype == torch.float32
    assert torch.equal(p["sub.conv0.w"][4], sd["encoder.pre_encode.conv.0.weight"][:, 0, 1, 1])
    assert p["sub.pw2.w"].dtype == torch.bfloat16 and p["sub.pw2.w"].shape == (C, C)
    # output Linear: columns permuted from (c, f) to (f, c)
    w = sd["encoder.pre_encode.out.weight"]
    assert p["sub.out.w"].shape == (d, F * C)
    assert p["sub.out.w"][5, 3 * C + 7] == w[5, 7 * F + 3].to(torch.bfloat16)
    assert p["L0.att.qkv.w"].shape == (3 * d, d) and p["L0.att.qkv.b"].shape == (3 * d,)
    assert torch.equal(p["L0.att.qkv.w"][d:2 * d], sd["encoder.layers.0.self_attn.linear_k.weight"].to(torch.bfloat16))
    assert p["L1.conv.dw.w"].shape == (cfg.conv_kernel, d)
    # pointwise_conv1 rows interleaved in blocks of 32: value rows 32j .. 32j+31, then their gate rows d + 32j ..
    w1 = sd["encoder.layers.1.conv.pointwise_conv1.weight"].squeeze(-1).to(torch.bfloat16)
    b1 = sd["encoder.layers.1.conv.pointwise_conv1.bias"]
    assert p["L1.conv.pw1.w"].shape == (2 * d, d)
    assert torch.equal(p["L1.conv.pw1.w"][64 * 3 + 5], w1[32 * 3 + 5]) and torch.equal(p["L1.conv.pw1.w"][64 * 3 + 32 + 5], w1[d + 32 * 3 + 5])
    assert p["L1.conv.pw1.b"][64 * 2 + 31] == b1[32 * 2 + 31] and p["L1.conv.pw1.b"][64 * 2 + 32] == b1[d + 32 * 2]
    assert sorted(W.glu_interleave_index(d).tolist()) == list(range(2 * d))
    H = cfg.pred_hidden
    # decode matrices are fragment-major: [n/16][k/16][lane = 16*kk + li][4]
    wl = torch.cat([sd["decoder.prediction.dec_rnn.lstm.weight_ih_l0"],
                    sd["decoder.prediction.dec_rnn.lstm.weight_hh_l0"]], dim=1)
    assert p["pred.lstm0.w"].shape == (4 * H // 16, 2 * H // 16, 64, 4)
    tn, kb, li, kk, e = 5, 3, 7, 2, 1
    assert p["pred.lstm0.w"][tn, kb, 16 * kk + li, e] == wl[16 * tn + li, 16 * kb + 4 * kk + e]
    jo = p["joint.out.w"]
    assert jo.shape == ((cfg.n_logits + 15) // 16, cfg.joint_hidden // 16, 64, 4)
    assert jo[3, 1, 16 * 3 + 15, 2] == sd["joint.joint_net.2.weight"][63, 16 + 12 + 2]     # last real row
    assert torch.all(jo.view(-1, cfg.joint_hidden // 16, 4, 16, 4)[3, :, :, 15 + 1 - 16:, :][:, :, :0] == 0)
    assert torch.equal(p["pred.lstm1.b"], sd["decoder.prediction.dec_rnn.lstm.bias_ih_l1"] +
                       sd["decoder.prediction.dec_rnn.lstm.bias_hh_l1"])
    assert p["pos.table"].shape == (79, d) and p["pos.table"].dtype == torch.bfloat16
    # row (cap - T) + n of the capped table is row n of the table for T  (slice trick in rs_encoder_forward)
    t13 = torch.from_numpy(W.rel_pos_table(cfg, 13)).to(torch.bfloat16)
    assert torch.equal(p["pos.table"][40 - 13:40 - 13 + 25], t13)
    idx, fw = p["fe.fb_idx"].numpy(), p["fe.fb_w"].numpy()
    fb = sd["preprocessor.featurizer.fb"][0].numpy()
    m = 37
    dense = np.zeros(257, np.float32)
    dense[idx[m, 0]:idx[m, 0] + idx[m, 1]] = fw[m, :idx[m, 1]]
    assert np.array_equal(dense, fb[m])
    tw = p["fe.twiddle"].numpy()
    assert np.allclose(tw[128], [0.0, -1.0], atol=1e-7) and np.allclose(tw[0], [1.0, 0.0])


def test_folded_batchnorm_equals_unfolded():
    cfg = TINY
    sd = W.synthetic_state_dict(cfg, 2)
    p = W.prepare_weights(cfg, sd, pos_cap=8)
    L = "encoder.layers.0.conv."
    x = torch.randn(2, cfg.d_model, 30)
    y = torch.nn.functional.conv1d(x, sd[L + "depthwise_conv.weight"], sd[L + "depthwise_conv.bias"], padding=4,
                                   groups=cfg.d_model)
    y = torch.nn.functional.batch_norm(y, sd[L + "batch_norm.running_mean"], sd[L + "batch_norm.running_var"],
                                       sd[L + "batch_norm.weight"], sd[L + "batch_norm.bias"], False, 0.0, cfg.bn_eps)
    z = torch.nn.functional.conv1d(x, p["L0.conv.dw.w"].t().unsqueeze(1).contiguous(), p["L0.conv.dw.b"], padding=4,
                                   groups=cfg.d_model)
    assert (y - z).abs().max() <= 1e-5


def test_nemo_archive_round_trip(tmp_path):
    cfg = TINY.with_(att_left=64, att_right=64, n_global=1, max_symbols=7)
    sd = W.synthetic_state_dict(cfg, 9)
    path = str(tmp_path / "m.nemo")
    W.write_nemo(path, cfg, sd, tokenizer_model=b"not-a-real-spm")
    cfg2, sd2, tok = W.read_nemo(path)
    assert cfg2 == cfg
    assert tok == b"not-a-real-spm"
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    with pytest.raises(ValueError):
        import tarfile
        bad = str(tmp_path / "bad.nemo")
        with tarfile.open(bad, "w"):
            pass
        W.read_nemo(bad)


def test_from_nemo_yaml_defaults():
    cfg = from_nemo_yaml({"encoder": {"d_model": 1024, "n_heads": 8, "n_layers": 24, "self_attention_model": "rel_pos"},
                          "decoder": {"vocab_size": 3000}})
    assert cfg == FASTCONFORMER_619M


def test_synthetic_tokenizer_and_sentencepiece_rule():
    tok = SyntheticTokenizer(3000, 0)
    assert len(tok.pieces) == len(set(tok.pieces)) == 3000
    assert tok.ids_to_text([0]) == ""                     # bare U+2581 -> dropped by decode.py:53
    assert tok.ids_to_text([1]) == "。"
    text = tok.ids_to_text([10, 0, 11])
    assert " " in text and not text.startswith(" ")
    assert SyntheticTokenizer(3000, 0).pieces == tok.pieces


def test_synthetic_batch():
    a, l = synthetic_batch(3, 1.0, seed=1)
    b, _ = synthetic_batch(3, 1.0, seed=1)
    assert a.shape == (3, 16000) and a.dtype == np.float32 and np.array_equal(a, b)
    assert l.tolist() == [16000] * 3 and 0.001 < np.abs(a).max() <= 1.0
    r, lr = synthetic_batch(6, 2.0, seed=2, ragged=True, min_seconds=0.5)
    assert lr.min() >= 8000 and lr.max() <= 32000
    for i in range(6):
        assert np.all(r[i, lr[i]:] == 0)


def test_shard_helpers():
    assert [rdist.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lens = [50, 10, 40, 20, 30, 60, 5]
    shards = rdist.shard_by_length(lens, 3)
    assert sorted(sum(shards, [])) == list(range(7))
    assert [len(s) for s in shards] == [3, 2, 2]
    assert max(lens[i] for i in shards[0]) <= min(lens[i] for i in shards[1])
    assert rdist.shard_by_length([], 2) == [[], []]
    # plans: equal lengths -> contiguous whole batches; ragged -> snake-dealt chunks, an even number per rank
    sh, batch = rdist.shard_plan([160000] * 2048, 8, 256)
    assert batch == 256 and [len(x) for x in sh] == [256] * 8 and sh[0] == list(range(256))
    sh, batch = rdist.shard_plan(list(range(1000, 1000 + 4096)), 4, 256)
    assert batch == 256 and [len(x) for x in sh] == [1024] * 4 and sh[0][:256] == list(range(256)) and sh[0][256:512] == list(range(7 * 256, 8 * 256))
    sh, batch = rdist.shard_plan([5, 1, 9, 3, 7], 2, 256)
    assert sorted(sum(sh, [])) == list(range(5)) and batch == 2 and sh == [[1, 3, 2], [0, 4]]
    assert rdist.shard_plan([], 3)[0] == [[], [], []]
    with pytest.raises(ValueError):
        rdist.shard_plan([1], 1, mode="nope")
    assert rdist.world_size() == 1 and rdist.rank() == 0 and rdist.max_over_ranks(1.5) == 1.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rdist.init("gloo")
    B, U = 3, 5
    ids = torch.full((B, U), rank * 100, dtype=torch.int32) + torch.arange(U, dtype=torch.int32)
    frames = ids + 1000
    n = torch.tensor([rank + 1, 0, U], dtype=torch.int32)
    g_ids, g_frames, g_n = rdist.gather_hypotheses(ids, frames, n)
    mx = rdist.max_over_ranks(float(rank + 1))
    rdist.barrier()
    torch.save((g_ids, g_frames, g_n, mx), os.path.join(out_dir, f"r{rank}.pt"))
    rdist.shutdown()


def test_gather_hypotheses_gloo_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_gather_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        g_ids, g_frames, g_n, mx = torch.load(os.path.join(str(tmp_path), f"r{r}.pt"))
        assert g_ids.shape == (6, 5) and mx == 2.0
        assert g_ids[:, 0].tolist() == [0, 0, 0, 100, 100, 100]
        assert torch.equal(g_frames, g_ids + 1000)
        assert g_n.tolist() == [1, 0, 5, 2, 0, 5]


def _fake_decode(lengths):
    """a stand-in for the GPU path: the 'hypothesis' of an utterance is a function of its length only"""
    def run_local(indices):
        ids = [[(lengths[i] * 7 + k) % 3000 for k in range(lengths[i] % 11)] for i in indices]
        frames = [[k // 2 for k in range(len(x))] for x in ids]
        return ids, frames, [lengths[i] // 1280 for i in indices]
    return run_local


def _sharded_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rdist.init("gloo")
    lengths = [16000 * 3 + 977 * i for i in (5, 0, 9, 3, 3, 12, 1)]        # 7 utterances: shards of 4 and 3
    calls = []

    def run_local(indices):
        calls.append(list(indices))
        return _fake_decode(lengths)(indices)

    counters = {}
    out = rdist.sharded_decode(lengths, run_local, counters, mode="contiguous")
    empty = rdist.sharded_decode([], lambda idx: ([], [], []))
    torch.save((out, calls, counters, empty), os.path.join(out_dir, f"s{rank}.pt"))
    rdist.shutdown()


def test_sharded_decode_gloo_world2(tmp_path):
    """config #3's mechanism on 2 CPU ranks: shard by length -> decode locally -> ONE gather -> caller order,
    identical on every rank and identical to the single-process answer (ragged shards, an utterance with
    no tokens, uneven shard sizes)"""
    world, port = 2, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    lengths = [16000 * 3 + 977 * i for i in (5, 0, 9, 3, 3, 12, 1)]
    want = rdist.sharded_decode(lengths, _fake_decode(lengths))              # world = 1 path
    assert [len(x) for x in want[0]].count(0) >= 0 and all(x is not None for x in want[0])
    seen = []
    for r in range(world):
        out, calls, counters, empty = torch.load(os.path.join(str(tmp_path), f"s{r}.pt"))
        assert out == want
        assert len(calls) == 1 and counters["collectives"] == 1
        assert empty == ([], [], [], None)
        seen.append(calls[0])
    assert sorted(seen[0] + seen[1]) == list(range(7)) and len(seen[0]) == 4 and len(seen[1]) == 3
    assert max(lengths[i] for i in seen[0]) <= min(lengths[i] for i in seen[1])   # contiguous in sorted order


def _fake_beam_decode(lengths):
    """like _fake_decode, plus a float32 'score' per utterance (what the beam search returns) with awkward bit patterns"""
    import numpy as np

    def run_local(indices):
        ids, frames, el = _fake_decode(lengths)(indices)
        scores = [float(np.float32(-0.1) * np.float32(lengths[i] % 977) - np.float32(1e-7) * np.float32(i)) for i in indices]
        return ids, frames, el, scores
    return run_local


def _sharded8_worker(rank, world, port, out_dir):
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    rdist.init("gloo")
    rng = np.random.default_rng(1235)
    lengths = rng.integers(2 * 16000, 10 * 16000 + 1, size=2048).tolist()      # BASELINE configs[2]: 2048 = 8 x 256, ragged
    counters = {}
    out = rdist.sharded_decode(lengths, _fake_beam_decode(lengths), counters, mode="contiguous")
    few = rdist.sharded_decode(lengths[:5], _fake_beam_decode(lengths[:5]))      # 5 utterances on 8 ranks: three empty shards
    # the default plan for ragged input: balanced chunks; run_local learns the chunk size and which utterances it got
    mine = {}

    def run_local(indices, batch):
        mine["indices"], mine["batch"] = list(indices), batch
        return _fake_beam_decode(lengths)(indices)

    bal = rdist.sharded_decode(lengths, run_local)
    torch.save((mine, bal[0] == out[0] and bal[1] == out[1] and bal[2] == out[2]), os.path.join(out_dir, f"b{rank}.pt"))
    if rank in (0, world - 1):
        torch.save((out, counters, few), os.path.join(out_dir, f"w{rank}.pt"))
    rdist.shutdown()


def test_sharded_decode_gloo_world8_2048_ragged(tmp_path):
    """BASELINE configs[2] on 8 CPU ranks: 2048 ragged utterances dealt as 8 length-sorted shards of 256, one
    all_gather per call, caller order restored, float32 scores bit-exact through the int32 payload, and a call with
    fewer utterances than ranks (empty shards) still completes with the same answer on every rank"""
    import numpy as np
    world, port = 8, _free_port()
    mp.spawn(_sharded8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(1235)
    lengths = rng.integers(2 * 16000, 10 * 16000 + 1, size=2048).tolist()
    want = rdist.sharded_decode(lengths, _fake_beam_decode(lengths))              # world = 1 path
    want_few = rdist.sharded_decode(lengths[:5], _fake_beam_decode(lengths[:5]))
    shards = rdist.shard_by_length(lengths, world)
    assert [len(s) for s in shards] == [256] * 8
    assert all(max(lengths[i] for i in shards[r]) <= min(lengths[i] for i in shards[r + 1]) for r in range(7))
    for r in (0, world - 1):
        out, counters, few = torch.load(os.path.join(str(tmp_path), f"w{r}.pt"))
        assert out[0] == want[0] and out[1] == want[1] and out[2] == want[2]
        assert np.array_equal(np.asarray(out[3], np.float32).view(np.int32), np.asarray(want[3], np.float32).view(np.int32))
        assert counters["collectives"] == 1
        assert few[:3] == want_few[:3] and few[3] == want_few[3]
    # the balanced plan (default for ragged input): same answers, every utterance dealt once, per-rank audio seconds AND
    # per-rank padded work (batch size x longest utterance of each batch) within 5 % of the mean, batches tightly padded
    secs, padded, seen = [], [], []
    for r in range(world):
        mine, same = torch.load(os.path.join(str(tmp_path), f"b{r}.pt"))
        assert same and mine["batch"] == 128
        idx = sorted(mine["indices"], key=lambda i: (lengths[i], i))
        seen += idx
        secs.append(sum(lengths[i] for i in idx) / 16000.0)
        groups = [idx[k:k + mine["batch"]] for k in range(0, len(idx), mine["batch"])]
        padded.append(sum(len(g) * max(lengths[i] for i in g) for g in groups) / 16000.0)
        for g in groups:                                  # a batch spans at most ~1/16 of the length range (+ sampling noise)
            assert max(lengths[i] for i in g) - min(lengths[i] for i in g) <= 0.08 * (10 - 2) * 16000
    assert sorted(seen) == list(range(2048))
    assert max(secs) <= 1.05 * (sum(secs) / world) and min(secs) >= 0.95 * (sum(secs) / world), secs
    assert max(padded) <= 1.05 * (sum(padded) / world), padded
    contiguous = [sum(lengths[i] for i in s) / 16000.0 for s in shards]
    assert max(contiguous) > 2.5 * min(contiguous)         # what the balanced plan replaces


# ---- .nemo archives shaped like NeMo writes them (not produced by this repo's write_nemo) -------------------------
NEMO_YAML = """
sample_rate: 16000
compute_eval_loss: false
log_prediction: true
model_defaults:
  enc_hidden: ${model.encoder.d_model}
  pred_hidden: 128
  joint_hidden: 128
tokenizer:
  dir: ???
  type: bpe
  model_path: nemo:0a1b2c_tokenizer.model
preprocessor:
  _target_: nemo.collections.asr.modules.AudioToMelSpectrogramPreprocessor
  sample_rate: ${model.sample_rate}
  normalize: per_feature
  window_size: 0.025
  window_stride: 0.01
  window: hann
  features: 80
  n_fft: 512
  frame_splicing: 1
  dither: 1.0e-05
  pad_to: 0
encoder:
  _target_: nemo.collections.asr.modules.ConformerEncoder
  feat_in: ${model.preprocessor.features}
  feat_out: -1
  n_layers: 2
  d_model: 256
  use_bias: %(use_bias)s
  subsampling: %(subsampling)s
  subsampling_factor: 8
  subsampling_conv_channels: 64
  causal_downsampling: false
  ff_expansion_factor: 2
  self_attention_model: rel_pos
  n_heads: 2
  att_context_size: [-1, -1]
  att_context_style: regular
  xscaling: true
  untie_biases: true
  pos_emb_max_len: 5000
  conv_kernel_size: 9
  conv_norm_type: %(conv_norm)s
  conv_context_size: null
  dropout: 0.1
decoder:
  _target_: nemo.collections.asr.modules.RNNTDecoder
  normalization_mode: null
  random_state_sampling: false
  blank_as_pad: true
  prednet:
    pred_hidden: ${model.model_defaults.pred_hidden}
    pred_rnn_layers: 2
    t_max: null
    dropout: 0.2
  vocab_size: 63
joint:
  _target_: nemo.collections.asr.modules.RNNTJoint
  log_softmax: null
  fuse_loss_wer: true
  jointnet:
    joint_hidden: ${model.model_defaults.joint_hidden}
    activation: relu
    dropout: %(joint_dropout)s
  num_classes: 63
decoding:
  strategy: %(strategy)s
  greedy:
    max_symbols: 10
  beam:
    beam_size: 4
target: nemo.collections.asr.models.rnnt_bpe_models.EncDecRNNTBPEModel
nemo_version: 2.6.1
"""


def _nemo_like_archive(path, sd, yaml_text, tokenizer_model=None, gz=True, wrap_state_dict=False):
    import io
    import tarfile
    with tarfile.open(path, "w:gz" if gz else "w") as tar:
        def add(name, data):
            info = tarfile.TarInfo(name)
            info.size = len(data)
            tar.addfile(info, io.BytesIO(data))
        add("./model_config.yaml", yaml_text.encode())
        buf = io.BytesIO()
        torch.save({"state_dict": dict(sd)} if wrap_state_dict else dict(sd), buf)
        add("./model_weights.ckpt", buf.getvalue())
        if tokenizer_model is not None:
            add("./0a1b2c_tokenizer.model", tokenizer_model)
            add("./0a1b2c_vocab.txt", b"x\n")
            add("./0a1b2c_tokenizer.vocab", b"x\t0\n")


def _tiny_sentencepiece(tmp_path):
    import sentencepiece as spm
    corpus = tmp_path / "corpus.txt"
    words = ["こんにちは", "世界", "音声", "認識", "です", "ます", "。", "、", "東京", "大阪", "天気", "今日", "明日"]
    rng = np.random.default_rng(0)
    corpus.write_text("\n".join("".join(rng.choice(words, size=6)) for _ in range(400)), encoding="utf-8")
    prefix = str(tmp_path / "spm")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=prefix, vocab_size=63, model_type="unigram",
                                   character_coverage=1.0, minloglevel=2)
    return open(prefix + ".model", "rb").read()


def test_read_nemo_like_archive(tmp_path):
    """OmegaConf interpolations / `???`, hash-prefixed artefacts, gzip, use_bias=false (bias-less encoder linears),
    joint dropout = 0 (the output Linear sits at joint_net.1), extra non-RNNT modules, a real SentencePiece model"""
    from reazonspeech_amd.runtime.config import UnsupportedCheckpoint
    from reazonspeech_amd.runtime.tokenizer import SentencePieceTokenizer
    cfg0 = TINY.with_(use_bias=False)
    sd = W.synthetic_state_dict(TINY, 3)
    sd = {k: v for k, v in sd.items() if not (k.startswith("encoder.layers.") and k.endswith(".bias") and
                                              ("linear" in k or "pointwise" in k or "depthwise" in k))}
    sd["joint.joint_net.1.weight"] = sd.pop("joint.joint_net.2.weight")
    sd["joint.joint_net.1.bias"] = sd.pop("joint.joint_net.2.bias")
    sd["ctc_decoder.decoder_layers.0.weight"] = torch.zeros(4, 4)            # hybrid models carry a CTC head
    sd["encoder.pos_enc.pe"] = torch.zeros(1, 9, 256)
    spm_model = _tiny_sentencepiece(tmp_path)
    fill = dict(use_bias="false", subsampling="dw_striding", conv_norm="batch_norm", joint_dropout="0.0",
                strategy="greedy_batch")
    path = str(tmp_path / "like.nemo")
    _nemo_like_archive(path, sd, NEMO_YAML % fill, spm_model, wrap_state_dict=True)
    cfg, sd2, tok = W.read_nemo(path)
    assert cfg == cfg0 and tok == spm_model
    prepared = W.prepare_weights(cfg, sd2, pos_cap=16)
    assert torch.count_nonzero(prepared["L0.ff1.b1"]) == 0 and prepared["L0.ff1.b1"].shape == (TINY.ff_dim,)
    assert torch.count_nonzero(prepared["L1.ln_ff1.b"]) > 0                    # LayerNorm biases are real
    full = W.prepare_weights(TINY, W.synthetic_state_dict(TINY, 3), pos_cap=16)
    assert torch.equal(prepared["joint.out.w"], full["joint.out.w"])
    t = SentencePieceTokenizer(tok)
    assert t.vocab_size == 63 and isinstance(t.ids_to_text([5, 9, 12]), str)
    assert t.ids_to_text([]) == ""

    # settings the kernels do not implement are refused, not silently mis-computed
    for bad in (dict(fill, subsampling="striding"), dict(fill, conv_norm="layer_norm")):
        _nemo_like_archive(path, sd, NEMO_YAML % bad, spm_model)
        with pytest.raises(UnsupportedCheckpoint):
            W.read_nemo(path)
    # parameters without a counterpart (Longformer separate global projections) are refused at weight prep
    extra = dict(sd)
    extra["encoder.layers.0.self_attn.global_q.weight"] = torch.zeros(256, 256)
    with pytest.raises(UnsupportedCheckpoint, match="global_q"):
        W.prepare_weights(cfg, extra, pos_cap=16)
    missing = {k: v for k, v in sd.items() if k != "encoder.layers.1.conv.batch_norm.running_var"}
    with pytest.raises(UnsupportedCheckpoint, match="running_var"):
        W.prepare_weights(cfg, missing, pos_cap=16)
    # the shipped checkpoint's strategy (ALSD beam search, decode.py:29,38) selects the device beam search ...
    _nemo_like_archive(path, sd, NEMO_YAML % dict(fill, strategy="alsd"), spm_model, gz=False)
    cfg_alsd, _, _ = W.read_nemo(path)
    assert (cfg_alsd.decoding, cfg_alsd.beam_size, cfg_alsd.alsd_max_target_len, cfg_alsd.beam_score_norm) == ("alsd", 4, 2.0, True)
    assert cfg.decoding == "greedy_batch"
    # ... and a search that is not implemented loads with a warning that this path decodes greedily instead
    _nemo_like_archive(path, sd, NEMO_YAML % dict(fill, strategy="maes"), spm_model, gz=False)
    with pytest.warns(RuntimeWarning, match="greedily"):
        assert W.read_nemo(path)[0].decoding == "greedy_batch"


def test_decode_policy_table(monkeypatch):
    """which decode kernel family a call gets (all are bit-identical; the choice is measured):
    one decode stream next to the encoder -> wide tiles + exact joint; two lanes (slack) or an idle chip -> screened
    joint + narrow tiles; small batches -> narrow tiles + exact joint; the environment overrides"""
    from reazonspeech_amd.runtime.model import AsrModel

    class Ctx:
        def __init__(self):
            self.opts = {}

        def set_option(self, k, v):
            self.opts[k] = v

    def pick(B, pipelined, lanes=1):
        c = Ctx()
        AsrModel._decode_policy(None, c, B, pipelined, lanes)
        return c.opts

    monkeypatch.delenv("RS_DECODE_SCREEN", raising=False)
    monkeypatch.delenv("RS_DECODE_NARROW", raising=False)
    assert pick(256, True, 1) == {"decode_narrow": 0, "decode_screen": 0}
    assert pick(256, True, 2) == {"decode_narrow": 1, "decode_screen": 1}
    assert pick(256, False) == {"decode_narrow": 1, "decode_screen": 1}
    assert pick(32, True, 1) == pick(32, True, 2) == pick(32, False) == {"decode_narrow": 1, "decode_screen": 0}
    monkeypatch.setenv("RS_DECODE_SCREEN", "0")
    assert pick(256, True, 2) == {}


def test_gemm_tile_height_rule():
    """the launcher's tile-height rule (host arithmetic inside librs_asr.so, callable without a GPU) gives the choices
    DESIGN.md §4 documents: 256 rows for ffn_up / qkv and 192 for pw1 and the N = 1024 residual family at the benchmark
    batch, the measured optimum of every encoder shape at B = 32, 64-row tiles for a lone utterance"""
    import ctypes
    from reazonspeech_amd.runtime import capi
    lib = capi.load()
    pick = lib.rs_debug_gemm_tile_height
    pick.argtypes = [ctypes.c_int] * 5
    pick.restype = ctypes.c_int
    RES, F32 = capi.GEMM_RESIDUAL | capi.GEMM_OUT_F32 | capi.GEMM_BIAS, capi.GEMM_OUT_F32 | capi.GEMM_BIAS
    shapes = {"ffn_up": (4096, 1024, capi.GEMM_BIAS | capi.GEMM_SILU), "ffn_down": (1024, 4096, RES), "qkv": (3072, 1024, capi.GEMM_BIAS),
              "out": (1024, 1024, RES), "pw1": (2048, 1024, capi.GEMM_BIAS | capi.GEMM_GLU), "sub_out": (1024, 2560, F32)}
    want = {256: dict(ffn_up=256, ffn_down=192, qkv=256, out=192, pw1=192, sub_out=192),
            32: dict(ffn_up=192, ffn_down=128, qkv=256, out=128, pw1=192, sub_out=128),      # profiles/r03a_gemm_b32_tiles.txt
            1: dict(ffn_up=64, ffn_down=64, qkv=64, out=64, pw1=64, sub_out=64)}
    for B, table in want.items():
        for name, bm in table.items():
            N, K, flags = shapes[name]
            assert pick(B * 138, N, K, 256, flags) == bm, (B, name)


# ---- A5: the loader against NeMo's PUBLISHED FastConformer-Transducer configuration (not this repo's writer) ---------
def test_strict_loader_on_nemo_published_fastconformer_xl_config():
    """tests/golden/nemo_fastconformer_xl_transducer_bpe.yaml follows NeMo's own
    examples/asr/conf/fastconformer/fast-conformer_transducer_bpe.yaml key for key (XL sizes): strict mode maps it onto the
    619M architecture, ignores the training-only sections, resolves the interpolations — and refuses ANY setting it has no
    mapping for, and every known variant the kernels do not compute"""
    import copy
    import yaml
    from reazonspeech_amd.runtime.config import from_nemo_yaml, FASTCONFORMER_619M, UnsupportedCheckpoint
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nemo_fastconformer_xl_transducer_bpe.yaml")
    with open(path) as fp:
        doc = yaml.safe_load(fp)
    cfg = from_nemo_yaml(copy.deepcopy(doc))
    assert cfg == FASTCONFORMER_619M.with_(decoding="alsd", beam_size=4)
    assert cfg.n_params() == 619223481                                  # README.rst:34-35: "619M"
    assert from_nemo_yaml({"model": copy.deepcopy(doc)}) == cfg         # training-style nesting

    def broken(section, key, value, sub=None):
        d = copy.deepcopy(doc)
        node = d[section] if sub is None else d[section][sub]
        node[key] = value
        return d

    for args in (("encoder", "mystery_knob", 3), ("preprocessor", "new_flag", True), ("decoder", "extra", 1),
                 ("joint", "whatever", 0), ("decoder", "unknown_prednet_knob", 1, "prednet"), ("joint", "gate", 1, "jointnet"),
                 # known settings whose non-default value changes the computation
                 ("encoder", "untie_biases", False), ("encoder", "subsampling", "striding"), ("encoder", "conv_norm_type", "layer_norm"),
                 ("encoder", "feat_out", 512), ("encoder", "causal_downsampling", True), ("decoder", "normalization_mode", "layer"),
                 ("joint", "num_extra_outputs", 5), ("preprocessor", "normalize", "all_features"), ("preprocessor", "highfreq", 7600),
                 ("preprocessor", "mel_norm", None), ("joint", "activation", "tanh", "jointnet"),
                 ("decoder", "rnn_hidden_size", 1024, "prednet")):
        with pytest.raises(UnsupportedCheckpoint):
            from_nemo_yaml(broken(*args))
    # the same unknown key is ignored when the caller opts out of strictness
    assert from_nemo_yaml(broken("encoder", "mystery_knob", 3), strict=False).d_model == 1024
    # window given in samples
    d = copy.deepcopy(doc)
    d["preprocessor"].pop("window_size"); d["preprocessor"]["n_window_size"] = 400
    assert from_nemo_yaml(d).win_length == 400


def test_beam_decoding_config():
    """decoding = "beam" (the default transducer beam search): allowed for both families, beam 1..64, label cap, scores"""
    from reazonspeech_amd.runtime.config import ESPNET_TINY, TINY
    c = ESPNET_TINY.with_(decoding="beam", beam_size=20).validate()
    assert c.has_scores and c.label_cap(100) == 216
    assert TINY.with_(decoding="beam", beam_size=64).validate().has_scores
    assert not TINY.validate().has_scores and TINY.label_cap(100) == 100 * TINY.max_symbols
    with pytest.raises(AssertionError):
        TINY.with_(decoding="beam", beam_size=65).validate()
    with pytest.raises(AssertionError):
        ESPNET_TINY.with_(decoding="alsd").validate()
    with pytest.raises(AssertionError):
        TINY.with_(decoding="beam", beam_size=8, beam_max_pops=4).validate()
