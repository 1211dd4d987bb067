"""From the three ONNX files of a `reazonspeech-k2-v2` repository (pkg/k2-asr/src/huggingface.py:41-66) to an icefall-style state
dict + ZipformerConfig, without onnx / onnxruntime / sherpa-onnx (runtime/onnx_lite.py).

[UPSTREAM, unverifiable here — no such file has ever been seen by this code] what icefall's export-onnx.py produces
(torch.onnx.export of OnnxEncoder(encoder, encoder_embed, encoder_proj), OnnxDecoder(decoder, decoder_proj),
OnnxJoiner(output_linear) after convert_scaled_to_non_scaled):
  * convolution weights / biases, Linear biases, embeddings and plain parameters used as they are (bypass_scale, BiasNorm.bias)
    keep their module names as initializer names;
  * a Linear's weight is the constant B operand of a MatMul (stored transposed, [in][out]) under an anonymous name; the NODE name
    carries the module scope ("/encoder/encoders.0/layers.0/feed_forward1/in_proj/MatMul");
  * constant-folded expressions lose their parameter: BiasNorm's exp(log_scale) is a scalar operand of a Mul in the norm's scope,
    SimpleDownsample's softmax(bias) a [ds, 1, 1] operand of a Mul in the downsample's scope (log of it is the bias up to a
    constant, which softmax ignores).
The architecture is not in the metadata of an offline Zipformer2 export: it is derived from the tensor shapes.
Quantized graphs (the "int8" / "int8-fp32" precisions) are refused."""
import math
import re

import numpy as np
import torch

from . import onnx_lite
from .config import UnsupportedCheckpoint
from .k2_config import ZipformerConfig
from .k2_weights import expected_shapes_k2

_RENAME = (("encoder_proj.", "joiner.encoder_proj."), ("decoder_proj.", "joiner.decoder_proj."), ("output_linear.", "joiner.output_linear."))


def _scope(node_name):
    """'/encoder/encoders.0/layers.0/feed_forward1/in_proj/MatMul' -> 'encoder.encoders.0.layers.0.feed_forward1.in_proj'"""
    parts = [p for p in node_name.split("/") if p]
    return ".".join(parts[:-1])


def _canonical(key):
    for a, b in _RENAME:
        if key.startswith(a):
            return b + key[len(a):]
    return key


def _put(sd, seen, key, value):
    """first occurrence wins (the same folded constant may appear at several nodes), a SECOND DIFFERENT value for one key is a
    layout this reader does not understand: refuse instead of loading a wrongly matched constant"""
    seen[key] = seen.get(key, 0) + 1
    if key in sd:
        if tuple(sd[key].shape) != tuple(value.shape) or not torch.equal(sd[key], value):
            raise UnsupportedCheckpoint(f"two different constants in the ONNX graph map to {key!r} (export layout differs from what this reader expects)")
        return
    sd[key] = value


def _collect(model, sd, seen):
    if any(n.op_type in ("DynamicQuantizeLinear", "MatMulInteger", "QLinearMatMul", "ConvInteger", "DequantizeLinear") for n in model.nodes):
        raise UnsupportedCheckpoint("a quantized (int8) ONNX graph: only the float32 files are read")
    init = model.initializers
    for name, arr in init.items():
        if re.search(r"\.(weight|bias|bypass_scale)$", name) and arr.dtype == np.float32:
            sd[_canonical(name)] = torch.from_numpy(np.array(arr))
    for n in model.nodes:
        consts = [i for i in n.inputs if i in init]
        if not consts:
            continue
        scope = _canonical(_scope(n.name) + ".")[:-1]
        if n.op_type == "MatMul" and len(n.inputs) == 2 and n.inputs[1] in init and init[n.inputs[1]].ndim == 2:
            _put(sd, seen, scope + ".weight", torch.from_numpy(np.array(init[n.inputs[1]]).T.copy()))
        elif n.op_type == "Gemm" and len(n.inputs) >= 2 and n.inputs[1] in init:
            _put(sd, seen, scope + ".weight", torch.from_numpy(np.array(init[n.inputs[1]])))
            if len(n.inputs) > 2 and n.inputs[2] in init:
                _put(sd, seen, scope + ".bias", torch.from_numpy(np.array(init[n.inputs[2]])))
        elif n.op_type == "Add" and scope + ".weight" in sd:
            c = init[consts[0]]
            if c.ndim == 1 and c.shape[0] == sd[scope + ".weight"].shape[0]:
                _put(sd, seen, scope + ".bias", torch.from_numpy(np.array(c)))
        elif n.op_type == "Mul":
            c = init[consts[0]]
            if (scope.endswith("norm") or scope.endswith("out_norm")) and c.size == 1 and float(c.reshape(-1)[0]) > 0:
                _put(sd, seen, scope + ".log_scale", torch.tensor(math.log(float(c.reshape(-1)[0])), dtype=torch.float32))
            elif (scope.endswith("downsample") or scope.endswith("downsample_output")) and c.ndim >= 1 and c.size in (2, 4, 8) and np.all(c > 0):
                total = float(np.array(c, np.float64).sum())
                if abs(total - 1.0) > 1e-4:
                    raise UnsupportedCheckpoint(f"the constant under {scope!r} sums to {total:.6f}: not a folded softmax(bias) (export layout differs)")
                _put(sd, seen, scope + ".bias", torch.from_numpy(np.log(np.array(c, np.float64).reshape(-1)).astype(np.float32)))


def derive_config(sd) -> ZipformerConfig:
    """ZipformerConfig from the shapes of an icefall-style state dict"""
    def shape(k):
        if k not in sd:
            raise UnsupportedCheckpoint(f"the ONNX files hold no tensor that maps to {k!r} (export layout differs from what this reader expects)")
        return tuple(sd[k].shape)
    c1 = shape("encoder_embed.conv.0.weight")[0]
    c2 = shape("encoder_embed.conv.4.weight")[0]
    c3 = shape("encoder_embed.conv.7.weight")[0]
    stacks = sorted({int(m.group(1)) for k in sd for m in [re.match(r"encoder\.encoders\.(\d+)\.", k)] if m})
    dims, layers, ffs, heads, kernels, dss = [], [], [], [], [], []
    for s in stacks:
        down = f"encoder.encoders.{s}.downsample.bias" in sd
        base = f"encoder.encoders.{s}." + ("encoder." if down else "") + "layers."
        n = 1 + max(int(m.group(1)) for k in sd for m in [re.match(re.escape(base) + r"(\d+)\.", k)] if m)
        L = base + "0."
        d = shape(L + "feed_forward2.in_proj.weight")[1]
        dims.append(d); layers.append(n)
        ffs.append(shape(L + "feed_forward2.in_proj.weight")[0])
        heads.append(shape(L + "self_attn1.in_proj.weight")[0] // 12)
        kernels.append(shape(L + "conv_module1.depthwise_conv.weight")[-1])
        dss.append(int(sd[f"encoder.encoders.{s}.downsample.bias"].numel()) if down else 1)
    h0 = heads[0]
    in_proj = shape(layer_prefix_from(dss, 0) + "self_attn_weights.in_proj.weight")[0]
    pos_w = shape(layer_prefix_from(dss, 0) + "self_attn_weights.linear_pos.weight")
    pd = pos_w[0] // h0
    qd = (in_proj // h0 - pd) // 2
    V, D = shape("decoder.embedding.weight")
    return ZipformerConfig(embed_channels=(c1, c2, c3), encoder_dim=tuple(dims), num_layers=tuple(layers), ff_dim=tuple(ffs), num_heads=tuple(heads),
                           cnn_kernel=tuple(kernels), downsampling=tuple(dss), query_head_dim=qd, value_head_dim=12, pos_head_dim=pd, pos_dim=pos_w[1],
                           vocab_size=V, decoder_dim=D, joiner_dim=shape("joiner.output_linear.weight")[1],
                           context_size=shape("decoder.conv.weight")[-1]).validate()


def layer_prefix_from(dss, s):
    return f"encoder.encoders.{s}." + ("" if dss[s] == 1 else "encoder.") + "layers.0."


def self_check(cfg, sd):
    """what a wrongly matched constant cannot survive: the recovered keys are EXACTLY the keys of an icefall Zipformer2 transducer
    of the derived architecture, every shape agrees, the parameter count equals cfg.n_params(), BiasNorm log-scales and bypass
    scales are finite and in the range training keeps them in ([UPSTREAM] scaling.py: log_scale limited to +-1.5, bypass_scale
    to [0, 1] with slack)."""
    want = expected_shapes_k2(cfg)
    missing, extra = sorted(set(want) - set(sd)), sorted(set(sd) - set(want))
    if missing or extra:
        raise UnsupportedCheckpoint(f"the ONNX files do not map onto an icefall Zipformer2 transducer of the derived architecture: "
                                    f"{len(missing)} tensor(s) not found ({', '.join(missing[:4])}{' ...' if len(missing) > 4 else ''}), "
                                    f"{len(extra)} unexpected ({', '.join(extra[:4])}{' ...' if len(extra) > 4 else ''})")
    for k, shp in want.items():
        if tuple(sd[k].shape) != shp:
            raise UnsupportedCheckpoint(f"tensor {k!r} has shape {tuple(sd[k].shape)}, the derived architecture needs {shp}")
        if not bool(torch.isfinite(sd[k]).all()):
            raise UnsupportedCheckpoint(f"tensor {k!r} holds non-finite values")
    n = sum(int(v.numel()) for v in sd.values())
    if n != cfg.n_params():
        raise UnsupportedCheckpoint(f"{n} parameters recovered, the derived architecture has {cfg.n_params()}")
    for k, v in sd.items():
        if k.endswith(".log_scale") and abs(float(v)) > 4.0:
            raise UnsupportedCheckpoint(f"{k!r} = {float(v):.3f}: not a BiasNorm log-scale (a wrongly matched Mul constant?)")
        if k.endswith(".bypass_scale") and (float(v.min()) < -0.5 or float(v.max()) > 1.5):
            raise UnsupportedCheckpoint(f"{k!r} outside [-0.5, 1.5]: not a bypass scale")


def read_k2_onnx(encoder_path, decoder_path, joiner_path):
    """-> (ZipformerConfig, icefall-style state dict), checked by `self_check`.  UNVERIFIED against a real export (module
    docstring): compare one transcript with sherpa-onnx before trusting a first real checkpoint."""
    sd, seen = {}, {}
    for path in (encoder_path, decoder_path, joiner_path):
        _collect(onnx_lite.load(path), sd, seen)
    cfg = derive_config(sd)
    self_check(cfg, sd)
    return cfg, sd
