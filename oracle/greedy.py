"""ctypes wrapper of oracle/rnnt_greedy.c.  TEST INFRASTRUCTURE (oracle/__init__.py)."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.OUT
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(_build.SRC):
            path = _build.build(force=True)
        _lib = ctypes.CDLL(path)
        _lib.rs_oracle_expf.restype = ctypes.c_float
        _lib.rs_oracle_expf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_sigmoidf.restype = ctypes.c_float
        _lib.rs_oracle_sigmoidf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_tanhf.restype = ctypes.c_float
        _lib.rs_oracle_tanhf.argtypes = [ctypes.c_float]
        _lib.rs_oracle_dot.restype = ctypes.c_float
        _lib.rs_oracle_rnnt_greedy.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def decoder_arrays(cfg, sd):
    """float32 numpy arrays in the layout rnnt_greedy.c expects (the same host-side prep the
    device path applies: W = [W_ih | W_hh], bias = b_ih + b_hh in float32)."""
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


def rnnt_greedy(cfg, sd, f, enc_lens, u_max=None):
    """f float32 [B, Tp, J] (numpy), enc_lens int[B] -> list of (ids, frames) per utterance."""
    L = lib()
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
