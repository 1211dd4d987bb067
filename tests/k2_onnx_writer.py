"""Test-only writer of the three ONNX files in the layout `runtime/k2_onnx.py: read_k2_onnx` is written for (its module docstring):
conv / bias / embedding / bypass tensors by name, Linear weights as anonymous transposed MatMul operands under scoped node names,
BiasNorm scales and down-sampling weights constant-folded.  NOT an icefall export: a stand-in with the documented structure."""
import re

import numpy as np
import torch

from reazonspeech_amd.runtime import onnx_lite


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
