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
from .k2_weights import layer_prefix

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


def _collect(model, sd):
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
            sd.setdefault(scope + ".weight", torch.from_numpy(np.array(init[n.inputs[1]]).T.copy()))
        elif n.op_type == "Gemm" and len(n.inputs) >= 2 and n.inputs[1] in init:
            sd.setdefault(scope + ".weight", torch.from_numpy(np.array(init[n.inputs[1]])))
            if len(n.inputs) > 2 and n.inputs[2] in init:
                sd.setdefault(scope + ".bias", torch.from_numpy(np.array(init[n.inputs[2]])))
        elif n.op_type == "Add" and scope + ".weight" in sd and scope + ".bias" not in sd:
            c = init[consts[0]]
            if c.ndim == 1 and c.shape[0] == sd[scope + ".weight"].shape[0]:
                sd[scope + ".bias"] = torch.from_numpy(np.array(c))
        elif n.op_type == "Mul":
            c = init[consts[0]]
            if (scope.endswith("norm") or scope.endswith("out_norm")) and c.size == 1 and float(c.reshape(-1)[0]) > 0 and scope + ".log_scale" not in sd:
                sd[scope + ".log_scale"] = torch.tensor(math.log(float(c.reshape(-1)[0])), dtype=torch.float32)
            elif (scope.endswith("downsample") or scope.endswith("downsample_output")) and c.ndim >= 1 and c.size in (2, 4, 8) and np.all(c > 0):
                sd.setdefault(scope + ".bias", torch.from_numpy(np.log(np.array(c, np.float64).reshape(-1)).astype(np.float32)))


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


def read_k2_onnx(encoder_path, decoder_path, joiner_path):
    """-> (ZipformerConfig, icefall-style state dict)"""
    sd = {}
    for path in (encoder_path, decoder_path, joiner_path):
        _collect(onnx_lite.load(path), sd)
    cfg = derive_config(sd)
    return cfg, sd


# ---- writer (tests) --------------------------------------------------------------------------------------------------------
def write_k2_onnx(cfg, sd, encoder_path, decoder_path, joiner_path):
    """three files in the layout `read_k2_onnx` is written for (see the module docstring): conv / bias / embedding / bypass tensors
    by name, Linear weights as anonymous transposed MatMul operands under scoped node names, BiasNorm scales and down-sampling
    weights constant-folded"""
    enc, dec, joi = onnx_lite.Model(), onnx_lite.Model(), onnx_lite.Model()
    counter = [0]

    def anon(model, arr):
        counter[0] += 1
        name = f"onnx::MatMul_{counter[0]}"
        model.initializers[name] = np.ascontiguousarray(arr, dtype=np.float32)
        return name

    def put(model, key, onnx_key):
        t = sd[key].detach().to(torch.float32).numpy()
        scope = "/" + onnx_key.rsplit(".", 1)[0].replace(".", "/").replace("/encoders/", "/encoders.").replace("/layers/", "/layers.").replace("/conv/", "/conv.")
        scope = re.sub(r"/(\d+)", r".\1", "/" + "/".join(onnx_key.split(".")[:-1]))
        if key.endswith("log_scale"):
            model.nodes.append(onnx_lite.Node(scope + "/Mul", "Mul", ["x", anon(model, np.exp(t).reshape(()))], ["y"]))
        elif key.endswith("downsample.bias") or key.endswith("downsample_output.bias"):
            e = np.exp(t - t.max())
            model.nodes.append(onnx_lite.Node(scope + "/Mul", "Mul", ["x", anon(model, (e / e.sum()).reshape(-1, 1, 1))], ["y"]))
        elif key.endswith(".weight") and t.ndim == 2 and "embedding" not in key:
            model.nodes.append(onnx_lite.Node(scope + "/MatMul", "MatMul", ["x", anon(model, t.T)], ["y"]))
        else:
            model.initializers[onnx_key] = np.ascontiguousarray(t)

    for key in sd:
        if key.startswith("joiner.encoder_proj."):
            put(enc, key, key[len("joiner."):])
        elif key.startswith("joiner.decoder_proj."):
            put(dec, key, key[len("joiner."):])
        elif key.startswith("joiner.output_linear."):
            put(joi, key, key[len("joiner."):])
        elif key.startswith("decoder."):
            put(dec, key, key)
        else:
            put(enc, key, key)
    enc.metadata.update({"model_type": "zipformer2", "version": "1", "model_author": "k2-fsa", "comment": "non-streaming zipformer2"})
    dec.metadata.update({"context_size": str(cfg.context_size), "vocab_size": str(cfg.vocab_size)})
    joi.metadata.update({"joiner_dim": str(cfg.joiner_dim)})
    onnx_lite.dump(encoder_path, enc)
    onnx_lite.dump(decoder_path, dec)
    onnx_lite.dump(joiner_path, joi)
