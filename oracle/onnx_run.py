"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

numpy interpreter for the 22 ONNX op types used by /root/reference/output/plnet_s1.onnx
(opset 17; the graph TensorRT executes at src/plnet.cpp:510).  This is the one piece of the
path with a TRUE known-answer oracle: real graph + real weights from the reference checkout.
ScatterElements follows sequential execution (last writer wins), as ORT/TensorRT do.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from airslam_amd import onnx_lite

_NP = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def run(model: onnx_lite.Model, feeds: Dict[str, np.ndarray], want=None) -> Dict[str, np.ndarray]:
    env: Dict[str, np.ndarray] = dict(model.initializers)
    env.update(feeds)
    for n in model.nodes:
        i = [env[x] if x else None for x in n.inputs]
        a = n.attrs
        op = n.op
        if op == "Constant":
            o = a["value"]
        elif op == "Cast":
            o = i[0].astype(_NP[a["to"]])
        elif op == "Shape":
            o = np.array(i[0].shape, dtype=np.int64)
        elif op == "Gather":
            o = np.take(i[0], i[1].astype(np.int64), axis=a.get("axis", 0))
        elif op == "GatherElements":
            o = np.take_along_axis(i[0], i[1].astype(np.int64), axis=a.get("axis", 0))
        elif op == "ScatterElements":
            o = i[0].copy()
            ax = a.get("axis", 0)
            assert ax == 0 and o.ndim == 1
            idx = i[1].astype(np.int64)
            for k in range(idx.shape[0]):          # sequential: last writer wins
                o[idx[k]] = i[2][k]
        elif op == "Mul":
            o = i[0] * i[1]
        elif op == "Add":
            o = i[0] + i[1]
        elif op == "Sub":
            o = i[0] - i[1]
        elif op == "Floor":
            o = np.floor(i[0])
        elif op == "Clip":
            o = i[0]
            if i[1] is not None:
                o = np.maximum(o, i[1])
            if len(i) > 2 and i[2] is not None:
                o = np.minimum(o, i[2])
        elif op == "Relu":
            o = np.maximum(i[0], 0)
        elif op == "Concat":
            o = np.concatenate(i, axis=a["axis"])
        elif op == "Transpose":
            o = np.transpose(i[0], a["perm"])
        elif op == "Reshape":
            shp = [int(s) for s in i[1]]
            shp = [i[0].shape[k] if s == 0 else s for k, s in enumerate(shp)]
            o = i[0].reshape(shp)
        elif op == "Flatten":
            ax = a.get("axis", 1)
            s = i[0].shape
            o = i[0].reshape(int(np.prod(s[:ax], dtype=np.int64)), int(np.prod(s[ax:], dtype=np.int64)))
        elif op == "Unsqueeze":
            o = i[0]
            for ax in sorted(int(x) for x in np.atleast_1d(i[1])):
                o = np.expand_dims(o, ax)
        elif op == "Slice":
            starts, ends = np.atleast_1d(i[1]), np.atleast_1d(i[2])
            axes = np.atleast_1d(i[3]) if len(i) > 3 and i[3] is not None else np.arange(len(starts))
            steps = np.atleast_1d(i[4]) if len(i) > 4 and i[4] is not None else np.ones(len(starts), np.int64)
            sl = [slice(None)] * i[0].ndim
            for s, e, ax, st in zip(starts, ends, axes, steps):
                s, e, ax, st = int(s), int(e), int(ax), int(st)
                dim = i[0].shape[ax]
                if st > 0:
                    e = min(e, dim)
                    sl[ax] = slice(s, e, st)
                else:
                    # negative step: clamp like ONNX (end below -dim means "through index 0")
                    s = s + dim if s < 0 else min(s, dim - 1)
                    e_eff = None if e < -dim else (e + dim if e < 0 else e)
                    sl[ax] = slice(s, e_eff, st)
            o = i[0][tuple(sl)]
        elif op == "Range":
            o = np.arange(i[0], i[1], i[2]).astype(i[0].dtype)
        elif op == "ConstantOfShape":
            v = a.get("value", np.zeros(1, np.float32))
            o = np.full([int(s) for s in i[0]], v.reshape(-1)[0], dtype=v.dtype)
        elif op == "Gemm":
            x = i[0].T if a.get("transA", 0) else i[0]
            w = i[1].T if a.get("transB", 0) else i[1]
            o = (a.get("alpha", 1.0) * (x.astype(np.float32) @ w.astype(np.float32))).astype(np.float32)
            if len(i) > 2 and i[2] is not None:
                o = o + a.get("beta", 1.0) * i[2]
            o = o.astype(np.float32)
        elif op == "Softmax":
            ax = a.get("axis", -1)
            z = i[0] - i[0].max(axis=ax, keepdims=True)
            e = np.exp(z)
            o = e / e.sum(axis=ax, keepdims=True)
        else:
            raise NotImplementedError(op)
        env[n.outputs[0]] = o
    names = want if want is not None else model.outputs
    return {k: env[k] for k in names}
