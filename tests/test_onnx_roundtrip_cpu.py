"""Rehearsal of the real-ONNX path for the three model families whose files the reference checkout does not carry (SuperPoint,
LightGlue, SuperGlue): ONNX-SHAPED files are written with airslam_amd.onnx_lite.save the way exporters lay such models out — Conv
nodes with their weights, Linear layers as MatMul (weight TRANSPOSED to [K][N]) + Add, LayerNormalization, weights shared by both
images referenced twice — in three naming styles, pushed through tools/onnx_to_pack.py, and the resulting pack must equal the
weights that went in, tensor for tensor.  (plnet_s1.onnx, the one real file, has its own golden test.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from airslam_amd import onnx_lite, weights
from airslam_amd.onnx_lite import Node
from conftest import ROOT


def _namer(style):
    cnt = [0]

    def name(pt_name, op):
        if style == "pytorch":
            return pt_name
        if style == "prefixed":
            return "model.backbone." + pt_name
        cnt[0] += 1
        return f"onnx::{op}_{1000 + 7 * cnt[0]}"           # what exporters / onnx-simplifier leave of a parameter's name
    return name


def _export(w, spec, style, twice=False):
    """Nodes + initializers of a straight-line graph over `spec`'s parameters (in spec order; `twice` = every layer applied to both
    images, weights shared, as a matcher's graph does)."""
    name = _namer(style)
    nodes, inits, cur = [], {}, "x"
    names = {}
    groups, i = [], 0
    while i < len(spec):
        n, shp = spec[i]
        if n.endswith(".weight") and i + 1 < len(spec) and spec[i + 1][0] == n[:-7] + ".bias":
            groups.append((n, spec[i + 1][0])); i += 2
        else:
            groups.append((n, None)); i += 1
    for rep in range(2 if twice else 1):
        cur = f"x{rep}"
        for gi, (wn, bn) in enumerate(groups):
            a = w[wn]
            out = f"t{rep}_{gi}"
            if not wn.endswith(".weight"):                                       # bin_score: consumed by a Concat, not by a weight node
                if rep == 0:
                    names[wn] = name(wn, "Concat"); inits[names[wn]] = a
                nodes.append(Node("Concat", f"cat{rep}_{gi}", [cur, names[wn]], [out]))
            elif a.ndim == 1:                                                    # LayerNorm
                if rep == 0:
                    names[wn] = name(wn, "LayerNormalization"); names[bn] = name(bn, "LayerNormalization")
                    inits[names[wn]] = a; inits[names[bn]] = w[bn]
                nodes.append(Node("LayerNormalization", f"ln{rep}_{gi}", [cur, names[wn], names[bn]], [out], {"axis": -1, "epsilon": 1e-5}))
            elif a.ndim == 4 or (a.ndim == 2 and wn.startswith(("kenc", "gnn", "final_proj"))):       # Conv2d / Conv1d(k = 1)
                if rep == 0:
                    names[wn] = name(wn, "Conv"); inits[names[wn]] = a if a.ndim == 4 else a[:, :, None]
                    if bn:
                        names[bn] = name(bn, "Conv"); inits[names[bn]] = w[bn]
                nodes.append(Node("Conv", f"conv{rep}_{gi}", [cur, names[wn]] + ([names[bn]] if bn else []), [out],
                                  {"kernel_shape": [3, 3] if a.ndim == 4 and a.shape[-1] == 3 else [1]}))
            else:                                                                # Linear: MatMul by the TRANSPOSED weight, then Add
                if rep == 0:
                    names[wn] = name(wn, "MatMul"); inits[names[wn]] = np.ascontiguousarray(a.T)
                    if bn:
                        names[bn] = name(bn, "Add"); inits[names[bn]] = w[bn]
                nodes.append(Node("MatMul", f"mm{rep}_{gi}", [cur, names[wn]], [out + "_mm" if bn else out]))
                if bn:
                    nodes.append(Node("Add", f"add{rep}_{gi}", [names[bn], out + "_mm"] if gi % 2 else [out + "_mm", names[bn]], [out]))
            cur = out
    inits["onnx::Reshape_5"] = np.array([1, -1, 256], np.int64)                  # exporter noise: shape constants
    return nodes, inits


CASES = {"superpoint": (weights.superpoint_spec, weights.synthetic_superpoint, False),
         "lightglue": (weights.lightglue_spec, weights.synthetic_lightglue, True),
         "superglue": (weights.superglue_spec, weights.synthetic_superglue, True)}


@pytest.mark.parametrize("style", ["pytorch", "prefixed", "anonymous"])
@pytest.mark.parametrize("kind", list(CASES))
def test_synthetic_export_round_trips_through_onnx_to_pack(kind, style, tmp_path):
    spec_fn, gen, twice = CASES[kind]
    w = gen(77)
    spec = spec_fn()
    nodes, inits = _export(w, spec, style, twice)
    onnx_path, pack_path = str(tmp_path / f"{kind}.onnx"), str(tmp_path / f"{kind}.airfe")
    onnx_lite.save(onnx_path, nodes, inits, ["x0", "x1"] if twice else ["x0"], [nodes[-1].outputs[0]])
    m = onnx_lite.load(onnx_path)                                                # the writer and the reader agree
    assert set(m.initializers) == set(inits) and len(m.nodes) == len(nodes)
    for k in inits:
        np.testing.assert_array_equal(m.initializers[k], inits[k])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "onnx_to_pack.py"), kind, onnx_path, pack_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    back = weights.load_pack(pack_path)
    weights.check_spec(back, spec)
    for name, _ in spec:
        np.testing.assert_array_equal(back[name], w[name], err_msg=name)


def test_graph_order_matching_refuses_a_graph_that_does_not_fit(tmp_path):
    w = weights.synthetic_lightglue(5, n_layers=2)
    spec = weights.lightglue_spec(2)
    nodes, inits = _export(w, spec, "anonymous", True)
    del nodes[3]                                                                 # one Linear missing: every later tensor would shift by one
    onnx_path = str(tmp_path / "broken.onnx")
    onnx_lite.save(onnx_path, nodes, inits, ["x0", "x1"], [nodes[-1].outputs[0]])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "onnx_to_pack.py"), "lightglue", onnx_path, str(tmp_path / "o.airfe")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "graph-order matching" in r.stderr
