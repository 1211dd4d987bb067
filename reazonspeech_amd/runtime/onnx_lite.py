"""A reader (and, for tests, a writer) of the part of the ONNX file format that carries a model's weights — without `onnx`,
`onnxruntime` or `protobuf`: the reference's k2 package hands three ONNX graphs to sherpa-onnx
(pkg/k2-asr/src/huggingface.py:73-83) and none of those packages exists in this image.

ONNX files are protobuf messages ([UPSTREAM] onnx/onnx.proto, stable field numbers):
  ModelProto   7 graph, 14 metadata_props (StringStringEntryProto: 1 key, 2 value)
  GraphProto   1 node, 2 name, 5 initializer
  NodeProto    1 input, 2 output, 3 name, 4 op_type
  TensorProto  1 dims, 2 data_type, 4 float_data, 5 int32_data, 7 int64_data, 8 name, 9 raw_data
Only these are interpreted; every other field is skipped by its wire type."""
import struct
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

FLOAT, UINT8, INT8, INT32, INT64, FLOAT16 = 1, 2, 3, 6, 7, 10
_DTYPES = {FLOAT: np.float32, UINT8: np.uint8, INT8: np.int8, INT32: np.int32, INT64: np.int64, FLOAT16: np.float16}


def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def fields(buf):
    """iterate (field number, wire type, value) of one message; length-delimited values come back as memoryview slices"""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val, pos = buf[pos:pos + n], pos + n
        elif wt == 5:
            val, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield num, wt, val


def _packed_varints(val, wt):
    if wt == 0:
        return [val]
    out, pos = [], 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


@dataclass
class Node:
    name: str = ""
    op_type: str = ""
    inputs: List[str] = field(default_factory=list)
    outputs: List[str] = field(default_factory=list)


@dataclass
class Model:
    nodes: List[Node] = field(default_factory=list)
    initializers: Dict[str, np.ndarray] = field(default_factory=dict)
    metadata: Dict[str, str] = field(default_factory=dict)


def _tensor(buf):
    dims, dtype, name, raw = [], FLOAT, "", None
    floats, i32, i64 = [], [], []
    for num, wt, val in fields(buf):
        if num == 1:
            dims += [_signed(v) for v in _packed_varints(val, wt)]
        elif num == 2:
            dtype = val
        elif num == 8:
            name = bytes(val).decode()
        elif num == 9:
            raw = bytes(val)
        elif num == 4:
            floats.append(np.frombuffer(bytes(val), "<f4") if wt == 2 else np.frombuffer(val, "<f4"))
        elif num == 5:
            i32 += [_signed(v) for v in _packed_varints(val, wt)]
        elif num == 7:
            i64 += [_signed(v) for v in _packed_varints(val, wt)]
        elif num in (13, 14) and (num == 13 or val == 1):
            raise ValueError(f"initializer {name!r} keeps its data in an external file: not supported")
    if dtype not in _DTYPES:
        raise ValueError(f"initializer {name!r}: unsupported ONNX data type {dtype}")
    if raw is not None:
        arr = np.frombuffer(raw, np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats)
    elif i64:
        arr = np.asarray(i64, np.int64)
    else:
        arr = np.asarray(i32, _DTYPES[dtype] if dtype != FLOAT16 else np.int32)
    return name, arr.astype(_DTYPES[dtype], copy=False).reshape(dims)


def load(path) -> Model:
    with open(path, "rb") as fp:
        data = fp.read()
    m = Model()
    for num, wt, val in fields(data):
        if num == 7:
            for gnum, gwt, gval in fields(val):
                if gnum == 1:
                    n = Node()
                    for nnum, _, nval in fields(gval):
                        if nnum == 1:
                            n.inputs.append(bytes(nval).decode())
                        elif nnum == 2:
                            n.outputs.append(bytes(nval).decode())
                        elif nnum == 3:
                            n.name = bytes(nval).decode()
                        elif nnum == 4:
                            n.op_type = bytes(nval).decode()
                    m.nodes.append(n)
                elif gnum == 5:
                    name, arr = _tensor(gval)
                    m.initializers[name] = arr
        elif num == 14:
            kv = {}
            for pnum, _, pval in fields(val):
                kv[pnum] = bytes(pval).decode()
            m.metadata[kv.get(1, "")] = kv.get(2, "")
    return m


# ---- writer (tests: a file in the layout the reader is written for) ---------------------------------------------------
def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(num, v):
    return _enc_varint(num << 3) + _enc_varint(v)


def dump(path, model: Model):
    g = bytearray()
    for n in model.nodes:
        body = b"".join(_ld(1, s.encode()) for s in n.inputs) + b"".join(_ld(2, s.encode()) for s in n.outputs)
        body += _ld(3, n.name.encode()) + _ld(4, n.op_type.encode())
        g += _ld(1, body)
    g += _ld(2, b"main_graph")
    for name, arr in model.initializers.items():
        arr = np.ascontiguousarray(arr)
        dt = {np.dtype(np.float32): FLOAT, np.dtype(np.int64): INT64, np.dtype(np.int32): INT32}[arr.dtype]
        body = b"".join(_vi(1, int(d)) for d in arr.shape) + _vi(2, dt) + _ld(8, name.encode()) + _ld(9, arr.tobytes())
        g += _ld(5, body)
    out = _vi(1, 8) + _ld(7, bytes(g))
    for k, v in model.metadata.items():
        out += _ld(14, _ld(1, k.encode()) + _ld(2, v.encode()))
    with open(path, "wb") as fp:
        fp.write(out)
