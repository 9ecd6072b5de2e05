"""Minimal ONNX protobuf reader (no `onnx` package needed).

Used to pull initializers (weights) and the node list out of the ONNX files the
reference loads with nvonnxparser (``src/plnet.cpp:87-93,162-168``).  Only the
wire-format fields needed for that are decoded (field numbers: SURVEY.md B.5).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple

import numpy as np

_DT = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64,
       9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


def _varint(b: bytes, i: int) -> Tuple[int, int]:
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b: bytes):
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]; i += ln
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fn, wt, v


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v: bytes) -> List[int]:
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_signed(x))
    return out


def parse_tensor(b: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dt = 1
    name = ""
    raw = None
    floats: List[float] = []
    ints: List[int] = []
    for fn, wt, v in _fields(b):
        if fn == 1:
            dims += _packed_varints(v) if wt == 2 else [_signed(v)]
        elif fn == 2:
            dt = v
        elif fn == 8:
            name = v.decode()
        elif fn == 9:
            raw = v
        elif fn == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fn in (5, 7):
            ints += _packed_varints(v) if wt == 2 else [_signed(v)]
    np_dt = _DT[dt]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dt).newbyteorder("<")).copy()
    elif floats:
        arr = np.array(floats, dtype=np_dt)
    else:
        arr = np.array(ints, dtype=np_dt)
    return name, arr.reshape(dims) if dims else arr.reshape(())


@dataclass
class Node:
    op: str
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, Any] = field(default_factory=dict)


def _parse_attr(b: bytes) -> Tuple[str, Any]:
    name = ""
    val: Any = None
    ints: List[int] = []
    floats: List[float] = []
    for fn, wt, v in _fields(b):
        if fn == 1:
            name = v.decode()
        elif fn == 2:
            val = struct.unpack("<f", v)[0]
        elif fn == 3:
            val = _signed(v)
        elif fn == 4:
            val = v
        elif fn == 5:
            val = parse_tensor(v)[1]
        elif fn == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif fn == 8:
            ints += _packed_varints(v) if wt == 2 else [_signed(v)]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def _parse_node(b: bytes) -> Node:
    n = Node("", "", [], [])
    for fn, _, v in _fields(b):
        if fn == 1:
            n.inputs.append(v.decode())
        elif fn == 2:
            n.outputs.append(v.decode())
        elif fn == 3:
            n.name = v.decode()
        elif fn == 4:
            n.op = v.decode()
        elif fn == 5:
            k, a = _parse_attr(v)
            n.attrs[k] = a
    return n


def _value_info_name(b: bytes) -> str:
    for fn, _, v in _fields(b):
        if fn == 1:
            return v.decode()
    return ""


@dataclass
class Model:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[str]
    outputs: List[str]
    opset: int


def load(path: str) -> Model:
    with open(path, "rb") as f:
        data = f.read()
    graph = None
    opset = 0
    for fn, _, v in _fields(data):
        if fn == 7:
            graph = v
        elif fn == 8:
            for f2, _, v2 in _fields(v):
                if f2 == 2:
                    opset = max(opset, v2)
    if graph is None:
        raise ValueError(f"{path}: no GraphProto")
    nodes: List[Node] = []
    inits: Dict[str, np.ndarray] = {}
    gin: List[str] = []
    gout: List[str] = []
    for fn, _, v in _fields(graph):
        if fn == 1:
            nodes.append(_parse_node(v))
        elif fn == 5:
            k, a = parse_tensor(v)
            inits[k] = a
        elif fn == 11:
            gin.append(_value_info_name(v))
        elif fn == 12:
            gout.append(_value_info_name(v))
    gin = [g for g in gin if g not in inits]
    return Model(nodes, inits, gin, gout, opset)


# --------------------------------------------------------------------------------------- writer (tests / rehearsals only)
def _enc_varint(v: int) -> bytes:
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        out.append(c | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_field(fn: int, wt: int, payload) -> bytes:
    key = _enc_varint((fn << 3) | wt)
    if wt == 0:
        return key + _enc_varint(payload)
    return key + _enc_varint(len(payload)) + payload


def _enc_tensor(name: str, a: np.ndarray) -> bytes:
    code = {np.dtype(v): k for k, v in _DT.items()}[np.dtype(a.dtype)]
    b = b"".join(_enc_field(1, 0, int(d)) for d in a.shape) + _enc_field(2, 0, code) + _enc_field(8, 2, name.encode())
    return b + _enc_field(9, 2, np.ascontiguousarray(a).tobytes())


def save(path: str, nodes: List[Node], initializers: Dict[str, np.ndarray], inputs: List[str], outputs: List[str], opset: int = 17) -> None:
    """Write a ModelProto that `load` reads back (fields of SURVEY.md B.5; integer / float / ints attributes only) — enough to
    build ONNX-SHAPED files for rehearsing tools/onnx_to_pack.py on the model files the reference checkout does not carry."""
    g = b""
    for n in nodes:
        nb = b"".join(_enc_field(1, 2, i.encode()) for i in n.inputs) + b"".join(_enc_field(2, 2, o.encode()) for o in n.outputs)
        nb += _enc_field(3, 2, n.name.encode()) + _enc_field(4, 2, n.op.encode())
        for k, v in n.attrs.items():
            ab = _enc_field(1, 2, k.encode())
            if isinstance(v, float):
                ab += _enc_varint((2 << 3) | 5) + struct.pack("<f", v)
            elif isinstance(v, int):
                ab += _enc_field(3, 0, v)
            else:
                ab += _enc_field(8, 2, b"".join(_enc_varint(int(x)) for x in v))
            nb += _enc_field(5, 2, ab)
        g += _enc_field(1, 2, nb)
    for k, a in initializers.items():
        g += _enc_field(5, 2, _enc_tensor(k, a))
    for name in inputs:
        g += _enc_field(11, 2, _enc_field(1, 2, name.encode()))
    for name in outputs:
        g += _enc_field(12, 2, _enc_field(1, 2, name.encode()))
    model = _enc_field(1, 0, 8) + _enc_field(7, 2, g) + _enc_field(8, 2, _enc_field(2, 0, opset))
    with open(path, "wb") as f:
        f.write(model)
