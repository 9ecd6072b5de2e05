#!/usr/bin/env python3
"""ONNX initializers -> airfe weight pack (≙ the reference's ONNX -> TensorRT engine step, src/plnet.cpp:24-196).

    python tools/onnx_to_pack.py plnet_s1   /path/output/plnet_s1.onnx                 /path/output/plnet_s1.airfe
    python tools/onnx_to_pack.py superpoint /path/output/superpoint_v1_sim_int32.onnx  /path/output/superpoint_v1_sim_int32.airfe
    python tools/onnx_to_pack.py lightglue  /path/output/superpoint_lightglue.onnx     /path/output/superpoint_lightglue.airfe
    python tools/onnx_to_pack.py superglue  /path/output/superglue_outdoor_sim_int32.onnx ...

Only output/plnet_s1.onnx ships with the reference checkout, so that is the only conversion of a REAL file the tests exercise.
Matching, in this order: exact initializer name; unique shape-compatible name suffix; and, for exports whose initializer names
were anonymised by the exporter / onnx-simplifier ("onnx::MatMul_1234"), GRAPH ORDER: the parameterised nodes (Conv, Gemm,
MatMul [+ the Add that carries its bias], LayerNormalization) are walked in topological order, every parameter tensor is taken
at its FIRST use (the matcher's weights are shared by both images), MatMul operands are transposed back to Linear's [N][K], and
the resulting sequence is laid over the spec's parameter sequence with every shape checked.  Anything that does not fit is
reported, never guessed.  tests/test_onnx_roundtrip_cpu.py rehearses all four specs on ONNX-shaped files written with
airslam_amd.onnx_lite.save in the three naming styles (PyTorch names, prefixed names, anonymised names).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airslam_amd import onnx_lite, weights  # noqa: E402

SPECS = {"plnet_s1": weights.plnet_s1_spec, "superpoint": weights.superpoint_spec, "lightglue": weights.lightglue_spec,
         "superglue": weights.superglue_spec}


def graph_order_parameters(m):
    """[(weight [N, K, ...] in PyTorch layout, bias or None)] of the parameterised nodes in graph order, first use only."""
    seen, seq = set(), []
    consumers = {}
    for n in m.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    for n in m.nodes:
        ins = [i for i in n.inputs]
        if n.op == "Conv" and len(ins) >= 2 and ins[1] in m.initializers:
            key, w = ins[1], m.initializers[ins[1]]
            b = m.initializers.get(ins[2]) if len(ins) > 2 else None
        elif n.op == "Gemm" and len(ins) >= 2 and ins[1] in m.initializers:
            key, w = ins[1], m.initializers[ins[1]]
            if not n.attrs.get("transB", 0):
                w = w.T
            b = m.initializers.get(ins[2]) if len(ins) > 2 else None
        elif n.op == "MatMul" and len(ins) == 2 and ins[1] in m.initializers:
            key, w = ins[1], m.initializers[ins[1]].T                      # x @ W with W [K][N]  ->  Linear weight [N][K]
            b = None
            for c in consumers.get(n.outputs[0], []):                      # bias = the Add fed by this MatMul and an initializer
                if c.op == "Add":
                    other = [i for i in c.inputs if i != n.outputs[0]]
                    if other and other[0] in m.initializers:
                        b = m.initializers[other[0]]
        elif n.op == "LayerNormalization" and len(ins) >= 3 and ins[1] in m.initializers:
            key, w, b = ins[1], m.initializers[ins[1]], m.initializers.get(ins[2])
        else:
            continue
        if key in seen:
            continue
        seen.add(key)
        seq.append((np.ascontiguousarray(w, dtype=np.float32), None if b is None else np.ascontiguousarray(b, dtype=np.float32)))
    return seq


def match_by_graph_order(m, spec):
    """Lay graph_order_parameters over the spec ((name.weight[, name.bias]) groups in spec order); scalars / 1-element tensors that no
    node consumes as a weight (SuperGlue's bin_score) are taken from the initializers by shape."""
    groups, i = [], 0
    while i < len(spec):
        name, shape = spec[i]
        if name.endswith(".weight") and i + 1 < len(spec) and spec[i + 1][0] == name[:-7] + ".bias":
            groups.append((name, shape, spec[i + 1][0], spec[i + 1][1])); i += 2
        else:
            groups.append((name, shape, None, None)); i += 1
    seq = graph_order_parameters(m)
    out, problems, k = {}, [], 0
    for wname, wshape, bname, bshape in groups:
        if not wname.endswith(".weight"):                                  # free-standing parameter (bin_score)
            cands = [v for v in m.initializers.values() if v.size == int(np.prod(wshape)) and v.dtype == np.float32 and v.ndim <= 1]
            if len(cands) == 1:
                out[wname] = cands[0].reshape(wshape).astype(np.float32)
            else:
                problems.append(f"{wname}: {len(cands)} candidates")
            continue
        if k >= len(seq):
            problems.append(f"{wname}: the graph has no more parameterised nodes")
            continue
        w, b = seq[k]; k += 1
        if w.size != int(np.prod(wshape)) or (w.ndim >= 2 and tuple(w.shape[:2]) != tuple(wshape[:2])):
            problems.append(f"{wname}: graph-order tensor {w.shape} does not fit {wshape}")
            continue
        out[wname] = w.reshape(wshape)
        if bname is not None:
            if b is None or b.size != int(np.prod(bshape)):
                problems.append(f"{bname}: no bias of shape {bshape} on that node")
            else:
                out[bname] = b.reshape(bshape)
    if k != len(seq):
        problems.append(f"{len(seq) - k} parameterised nodes of the graph were left over")
    return out, problems


def linear_layout(m):
    """initializer name -> the tensor in Linear's [N][K] layout: a MatMul's weight operand is stored [K][N] (x @ W) and a Gemm's is unless
    transB — for a SQUARE weight a name-and-shape match alone would silently take the transposed matrix."""
    out = dict(m.initializers)
    for n in m.nodes:
        if n.op == "MatMul" and len(n.inputs) == 2 and n.inputs[1] in m.initializers and m.initializers[n.inputs[1]].ndim == 2:
            out[n.inputs[1]] = np.ascontiguousarray(m.initializers[n.inputs[1]].T)
        elif n.op == "Gemm" and len(n.inputs) >= 2 and n.inputs[1] in m.initializers and not n.attrs.get("transB", 0):
            out[n.inputs[1]] = np.ascontiguousarray(m.initializers[n.inputs[1]].T)
    return out


def convert(kind: str, onnx_path: str, out_path: str) -> None:
    m = onnx_lite.load(onnx_path)
    spec = SPECS[kind]()
    out, missing = {}, []
    inits = linear_layout(m) if kind != "plnet_s1" else m.initializers
    for name, shape in spec:
        if kind == "plnet_s1" and name == "sample_t":
            cands = [v.reshape(-1) for v in m.initializers.values() if v.size == 30 and v.dtype == np.float32
                     and v.reshape(-1)[0] < 0.5]
            out[name] = cands[0] if cands else weights.linspace_t()
            continue
        if name in inits and tuple(inits[name].shape) == tuple(shape):
            out[name] = inits[name].astype(np.float32)
            continue
        cands = [k for k, v in inits.items() if k.endswith(name) and tuple(v.shape) == tuple(shape)]
        if len(cands) == 1:
            out[name] = inits[cands[0]].astype(np.float32)
        else:
            missing.append((name, shape, cands))
    if missing and kind != "plnet_s1":
        # Names are no (or only partial) help: go by graph order.  Many tensors share a shape (LightGlue's out_proj / to_qk / to_v / to_out are
        # all [256, 256]), so graph order alone could silently permute them if the export's topological order differed from the spec's: every
        # tensor that WAS placed by name must be the very tensor graph order puts in that slot, or the pack is refused.
        by_name = out
        out, problems = match_by_graph_order(m, spec)
        for name, t in by_name.items():
            if name in out and not np.array_equal(np.asarray(out[name], np.float32).reshape(-1), np.asarray(t, np.float32).reshape(-1)):
                problems.append(f"{name}: the tensor with that name is not the one graph order assigns (export order differs from the spec order)")
        if problems:
            for pr in problems:
                print(f"graph-order matching: {pr}", file=sys.stderr)
            raise SystemExit(f"{len(missing)} tensors could not be matched by name and graph-order matching failed too")
        print(f"{len(missing)} tensors placed by graph order; {len(by_name)} name matches agree with it")
        missing = []
    if missing:
        for name, shape, cands in missing:
            print(f"unplaced: {name} {shape} candidates={cands}", file=sys.stderr)
        raise SystemExit(f"{len(missing)} tensors could not be matched; write a name map for this export")
    if kind in ("superpoint", "plnet_s0") and "conv1a.weight" in out:
        # fp16 activation range (include/airfe.h "activation range"): layers whose calibration maxima exceed 2048 hand on a power-of-two fraction of their
        # activations, undone at the fp32 heads — exact, and a no-op for a network whose activations are already small (every layer reports factor 1)
        out, report = weights.fold_activation_scales(out)
        print("activation maxima over the calibration frames -> power-of-two factors: " + ", ".join(f"{k} {m:.3g} x{c:g}" for k, (m, c) in report.items()))
    weights.save_pack(out_path, out)
    print(f"{out_path}: {len(out)} tensors, {sum(v.size for v in out.values())} parameters")


if __name__ == "__main__":
    if len(sys.argv) != 4 or sys.argv[1] not in SPECS:
        raise SystemExit(__doc__)
    convert(*sys.argv[1:])
