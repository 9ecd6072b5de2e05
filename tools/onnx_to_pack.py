#!/usr/bin/env python3
"""ONNX initializers -> airfe weight pack (≙ the reference's ONNX -> TensorRT engine step, src/plnet.cpp:24-196).

    python tools/onnx_to_pack.py plnet_s1   /path/output/plnet_s1.onnx                 /path/output/plnet_s1.airfe
    python tools/onnx_to_pack.py superpoint /path/output/superpoint_v1_sim_int32.onnx  /path/output/superpoint_v1_sim_int32.airfe
    python tools/onnx_to_pack.py lightglue  /path/output/superpoint_lightglue.onnx     /path/output/superpoint_lightglue.airfe
    python tools/onnx_to_pack.py superglue  /path/output/superglue_outdoor_sim_int32.onnx ...

Only output/plnet_s1.onnx ships with the reference checkout, so that is the only conversion exercised by the tests.
For the others the tool matches initializers to the pack spec by exact name, then by unique shape-compatible suffix;
anything it cannot place is reported instead of guessed (simplified ONNX exports often anonymise names).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airslam_amd import onnx_lite, weights  # noqa: E402

SPECS = {"plnet_s1": weights.plnet_s1_spec, "superpoint": weights.superpoint_spec, "lightglue": weights.lightglue_spec,
         "superglue": weights.superglue_spec}


def convert(kind: str, onnx_path: str, out_path: str) -> None:
    m = onnx_lite.load(onnx_path)
    spec = SPECS[kind]()
    out, missing = {}, []
    for name, shape in spec:
        if kind == "plnet_s1" and name == "sample_t":
            cands = [v.reshape(-1) for v in m.initializers.values() if v.size == 30 and v.dtype == np.float32
                     and v.reshape(-1)[0] < 0.5]
            out[name] = cands[0] if cands else weights.linspace_t()
            continue
        if name in m.initializers and tuple(m.initializers[name].shape) == tuple(shape):
            out[name] = m.initializers[name].astype(np.float32)
            continue
        cands = [k for k, v in m.initializers.items() if k.endswith(name) and tuple(v.shape) == tuple(shape)]
        if len(cands) == 1:
            out[name] = m.initializers[cands[0]].astype(np.float32)
        else:
            missing.append((name, shape, cands))
    if missing:
        for name, shape, cands in missing:
            print(f"unplaced: {name} {shape} candidates={cands}", file=sys.stderr)
        raise SystemExit(f"{len(missing)} tensors could not be matched; write a name map for this export")
    weights.save_pack(out_path, out)
    print(f"{out_path}: {len(out)} tensors, {sum(v.size for v in out.values())} parameters")


if __name__ == "__main__":
    if len(sys.argv) != 4 or sys.argv[1] not in SPECS:
        raise SystemExit(__doc__)
    convert(*sys.argv[1:])
