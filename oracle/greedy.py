"""ctypes wrapper of oracle/rnnt_greedy.c and oracle/rnnt_alsd.c.  TEST INFRASTRUCTURE (oracle/__init__.py)."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.build(force=True) if _build.stale() else _build.OUT
        _lib = ctypes.CDLL(path)
        _lib.rs_oracle_expf.restype = ctypes.c_float
        _lib.rs_oracle_expf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_sigmoidf.restype = ctypes.c_float
        _lib.rs_oracle_sigmoidf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_tanhf.restype = ctypes.c_float
        _lib.rs_oracle_tanhf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_dot.restype = ctypes.c_float
        _lib.rs_oracle_rnnt_greedy.restype = ctypes.c_int
        _lib.rs_oracle_rnnt_alsd.restype = ctypes.c_int
        _lib.rs_oracle_logf.restype = ctypes.c_float
        _lib.rs_oracle_logf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_logaddexpf.restype = ctypes.c_float
        _lib.rs_oracle_logaddexpf.argtypes = [ctypes.c_float, ctypes.c_float]
        _lib.rs_oracle_lse.restype = ctypes.c_float
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def decoder_arrays(cfg, sd):
    """float32 numpy arrays in the layout rnnt_greedy.c expects (the same host-side prep the
    device path applies: W = [W_ih | W_hh], bias = b_ih + b_hh in float32)."""
    if getattr(cfg, "espnet", False):
        return decoder_arrays_espnet(cfg, sd)
    P = "decoder.prediction.dec_rnn.lstm."
    ws, bs = [], []
    for l in range(cfg.pred_layers):
        w = np.concatenate([sd[P + f"weight_ih_l{l}"].numpy(), sd[P + f"weight_hh_l{l}"].numpy()], axis=1)
        ws.append(np.ascontiguousarray(w, dtype=np.float32))
        bs.append((sd[P + f"bias_ih_l{l}"].numpy().astype(np.float32) +
                   sd[P + f"bias_hh_l{l}"].numpy().astype(np.float32)).astype(np.float32))
    c = lambda k: np.ascontiguousarray(sd[k].numpy(), dtype=np.float32)  # noqa: E731
    return dict(embed=c("decoder.prediction.embed.weight"), lstm_w=ws, lstm_b=bs,
                Wp=c("joint.pred.weight"), bp=c("joint.pred.bias"),
                Wo=c("joint.joint_net.2.weight"), bo=c("joint.joint_net.2.bias"))


def decoder_arrays_espnet(cfg, sd):
    """the same arrays from an ESPnet2 state dict ([UPSTREAM] TransducerDecoder: one LSTM module per layer; JointNetwork:
    lin_dec has no bias, lin_enc is hoisted into `f` by the caller)"""
    ws, bs = [], []
    for l in range(cfg.pred_layers):
        P = f"decoder.decoder.{l}."
        ws.append(np.ascontiguousarray(np.concatenate([sd[P + "weight_ih_l0"].numpy(), sd[P + "weight_hh_l0"].numpy()], axis=1), dtype=np.float32))
        bs.append((sd[P + "bias_ih_l0"].numpy().astype(np.float32) + sd[P + "bias_hh_l0"].numpy().astype(np.float32)).astype(np.float32))
    c = lambda k: np.ascontiguousarray(sd[k].numpy(), dtype=np.float32)  # noqa: E731
    return dict(embed=c("decoder.embed.weight"), lstm_w=ws, lstm_b=bs, Wp=c("joint_network.lin_dec.weight"),
                bp=np.zeros((cfg.joint_hidden,), np.float32), Wo=c("joint_network.lin_out.weight"), bo=c("joint_network.lin_out.bias"))


def rnnt_greedy(cfg, sd, f, enc_lens, u_max=None):
    """f float32 [B, Tp, J] (numpy), enc_lens int[B] -> list of (ids, frames) per utterance.  The joint activation follows
    the model family (ReLU for NeMo, tanh for ESPnet); ESPnet's greedy search is max_symbols = 1."""
    L = lib()
    L.rs_oracle_set_joint_act(1 if getattr(cfg, "espnet", False) else 0)
    arr = decoder_arrays(cfg, sd)
    f = np.ascontiguousarray(f, dtype=np.float32)
    B, Tp, J = f.shape
    enc_lens = np.ascontiguousarray(enc_lens, dtype=np.int32)
    if u_max is None:
        u_max = Tp * cfg.max_symbols
    ids = np.zeros((B, u_max), np.int32)
    frames = np.zeros((B, u_max), np.int32)
    n_ids = np.zeros((B,), np.int32)
    PF = ctypes.POINTER(ctypes.c_float)
    wl = (PF * cfg.pred_layers)(*[_fp(w) for w in arr["lstm_w"]])
    bl = (PF * cfg.pred_layers)(*[_fp(b) for b in arr["lstm_b"]])
    rc = L.rs_oracle_rnnt_greedy(_fp(f), _ip(enc_lens), B, Tp, J, cfg.pred_hidden, cfg.pred_layers,
                                 cfg.n_logits, cfg.blank_id, cfg.max_symbols, _fp(arr["embed"]), wl, bl,
                                 _fp(arr["Wp"]), _fp(arr["bp"]), _fp(arr["Wo"]), _fp(arr["bo"]),
                                 u_max, _ip(ids), _ip(frames), _ip(n_ids))
    if rc != 0:
        raise RuntimeError(f"oracle greedy overflowed u_max={u_max}")
    return [(ids[b, :n_ids[b]].tolist(), frames[b, :n_ids[b]].tolist()) for b in range(B)]


def alsd_budget(enc_lens, max_target_len):
    """label budget per utterance (oracle/alsd.py: a float is a multiple of the frame count, an int is absolute)"""
    if isinstance(max_target_len, float):
        return np.asarray([int(max_target_len * int(t)) for t in enc_lens], np.int32)
    return np.full((len(enc_lens),), int(max_target_len), np.int32)


def rnnt_alsd(cfg, sd, f, enc_lens, beam=4, max_target_len=2.0, score_norm=True, recombine="upstream", out_cap=None):
    """f float32 [B, Tp, J] (numpy), enc_lens int[B] -> list of (ids, alignment steps, score) of the best
    hypothesis per utterance, in the fixed float32 evaluation order of rnnt_alsd.c."""
    L = lib()
    arr = decoder_arrays(cfg, sd)
    f = np.ascontiguousarray(f, dtype=np.float32)
    B, Tp, J = f.shape
    enc_lens = np.ascontiguousarray(enc_lens, dtype=np.int32)
    u_max = alsd_budget(enc_lens, max_target_len)
    if out_cap is None:
        out_cap = max(1, int((enc_lens + u_max).max())) if B else 1
    ids = np.zeros((B, out_cap), np.int32)
    steps = np.zeros((B, out_cap), np.int32)
    n_ids = np.zeros((B,), np.int32)
    scores = np.zeros((B,), np.float32)
    PF = ctypes.POINTER(ctypes.c_float)
    wl = (PF * cfg.pred_layers)(*[_fp(w) for w in arr["lstm_w"]])
    bl = (PF * cfg.pred_layers)(*[_fp(b) for b in arr["lstm_b"]])
    L.rs_oracle_set_joint_act(1 if getattr(cfg, "espnet", False) else 0)
    rc = L.rs_oracle_rnnt_alsd(_fp(f), _ip(enc_lens), B, Tp, J, cfg.pred_hidden, cfg.pred_layers, cfg.n_logits,
                               cfg.blank_id, _fp(arr["embed"]), wl, bl, _fp(arr["Wp"]), _fp(arr["bp"]),
                               _fp(arr["Wo"]), _fp(arr["bo"]), int(beam), _ip(u_max), int(bool(score_norm)),
                               int(recombine == "merge"), out_cap, _ip(ids), _ip(steps), _ip(n_ids), _fp(scores))
    if rc != 0:
        raise RuntimeError(f"oracle alsd overflowed out_cap={out_cap}")
    return [(ids[b, :n_ids[b]].tolist(), steps[b, :n_ids[b]].tolist(), float(scores[b])) for b in range(B)]


def espnet_beam(cfg, sd, f, enc_lens, beam=20, score_norm=True, max_pops=None, out_cap=None, with_frames=False, workers=None):
    """f float32 [B, Tp, J] (numpy), enc_lens int[B] -> list of (ids, score, pops) of the best hypothesis per utterance
    under ESPnet's default transducer beam search, in the fixed float32 evaluation order of espnet_beam.c.
    with_frames: (ids, frames, score, pops) — frames = the frame each label was appended at.
    Utterances are independent and the C routine is re-entrant (its only global, the joint activation, is set before the
    calls): rows run on `workers` threads (default: one per core, at most 32) — a whole 358-frame row at beam 20 is ~20 - 40 s
    of scalar C."""
    L = lib()
    L.rs_oracle_set_joint_act(1 if getattr(cfg, "espnet", False) else 0)
    arr = decoder_arrays(cfg, sd)
    f = np.ascontiguousarray(f, dtype=np.float32)
    B, Tp, J = f.shape
    enc_lens = np.ascontiguousarray(enc_lens, dtype=np.int32)
    if max_pops is None:
        max_pops = 8 * int(beam)
    if out_cap is None:
        out_cap = max(1, Tp * 4)
    ids = np.zeros((B, out_cap), np.int32)
    frames = np.zeros((B, out_cap), np.int32)
    n_ids = np.zeros((B,), np.int32)
    scores = np.zeros((B,), np.float32)
    pops = np.zeros((B,), np.int32)
    PF = ctypes.POINTER(ctypes.c_float)
    wl = (PF * cfg.pred_layers)(*[_fp(w) for w in arr["lstm_w"]])
    bl = (PF * cfg.pred_layers)(*[_fp(b) for b in arr["lstm_b"]])
    L.rs_oracle_espnet_beam.restype = ctypes.c_int

    def rows(b0, b1):          # utterances [b0, b1): views into the shared input / output arrays
        return L.rs_oracle_espnet_beam(_fp(f[b0:b1]), _ip(enc_lens[b0:b1]), b1 - b0, Tp, J, cfg.pred_hidden, cfg.pred_layers, cfg.n_logits,
                                       cfg.blank_id, _fp(arr["embed"]), wl, bl, _fp(arr["Wp"]), _fp(arr["bp"]), _fp(arr["Wo"]),
                                       _fp(arr["bo"]), int(beam), int(bool(score_norm)), int(max_pops), int(out_cap), _ip(ids[b0:b1]),
                                       _ip(frames[b0:b1]), _ip(n_ids[b0:b1]), _fp(scores[b0:b1]), _ip(pops[b0:b1]))

    if workers is None:
        workers = min(32, os.cpu_count() or 1)
    if B > 1 and workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(workers, B)) as pool:
            rcs = list(pool.map(lambda b: rows(b, b + 1), range(B)))
        rc = max(rcs)
    else:
        rc = rows(0, B)
    if rc != 0:
        raise RuntimeError(f"oracle espnet beam search overflowed (max_pops={max_pops}, out_cap={out_cap})")
    if with_frames:
        return [(ids[b, :n_ids[b]].tolist(), frames[b, :n_ids[b]].tolist(), float(scores[b]), int(pops[b])) for b in range(B)]
    return [(ids[b, :n_ids[b]].tolist(), float(scores[b]), int(pops[b])) for b in range(B)]


def k2_greedy(cfg, sd, f, enc_lens, u_max=None):
    """Zipformer family (oracle/k2_greedy.c): f float32 [B, Tp, J] = joiner.encoder_proj(encoder output), enc_lens int[B]
    -> list of (ids, frames) per utterance under sherpa-onnx's offline greedy search with icefall's stateless decoder."""
    L = lib()
    L.rs_oracle_k2_greedy.restype = ctypes.c_int
    c = lambda k: np.ascontiguousarray(sd[k].numpy(), dtype=np.float32)  # noqa: E731
    f = np.ascontiguousarray(f, dtype=np.float32)
    B, Tp, J = f.shape
    enc_lens = np.ascontiguousarray(enc_lens, dtype=np.int32)
    if u_max is None:
        u_max = max(Tp, 1)
    ids = np.zeros((B, u_max), np.int32)
    frames = np.zeros((B, u_max), np.int32)
    n_ids = np.zeros((B,), np.int32)
    embed, conv_w = c("decoder.embedding.weight"), c("decoder.conv.weight")
    wp, bp = c("joiner.decoder_proj.weight"), c("joiner.decoder_proj.bias")
    wo, bo = c("joiner.output_linear.weight"), c("joiner.output_linear.bias")
    rc = L.rs_oracle_k2_greedy(_fp(f), _ip(enc_lens), B, Tp, J, cfg.decoder_dim, cfg.vocab_size, cfg.blank_id, cfg.unk_id, _fp(embed),
                               _fp(conv_w), _fp(wp), _fp(bp), _fp(wo), _fp(bo), u_max, _ip(ids), _ip(frames), _ip(n_ids))
    if rc != 0:
        raise RuntimeError(f"oracle k2 greedy overflowed u_max={u_max}")
    return [(ids[b, :n_ids[b]].tolist(), frames[b, :n_ids[b]].tolist()) for b in range(B)]
