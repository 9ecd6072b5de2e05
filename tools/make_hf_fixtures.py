#!/usr/bin/env python3
"""tests/golden/hf_pin.npz: outputs of Hugging Face transformers' SuperPoint / LightGlue / SuperGlue ports on the seeded inputs of
tests/hf_cases.py with the seeded weights of airslam_amd.weights (oracle/hf_pin.py).  Run in the build container (transformers is
installed there); the GPU twin tests/test_gpu_hf_pin.py compares the HIP library with this file where transformers is missing.

    python tools/make_hf_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import hf_cases                                   # noqa: E402
from airslam_amd import weights                   # noqa: E402
from oracle import hf_pin                         # noqa: E402


def main():
    out = {"transformers_version": np.array(hf_pin.transformers_version())}
    sp = weights.synthetic_superpoint(1234)
    for h, w, seed in hf_cases.SP_IMAGES:
        _, x = hf_cases.sp_input(h, w, seed)
        nms, desc = hf_pin.superpoint_maps(sp, x)
        idx = np.flatnonzero(nms).astype(np.int32)             # the suppressed map is sparse: (index, value) pairs
        out[f"sp_{h}_{w}_{seed}_nms_idx"] = idx
        out[f"sp_{h}_{w}_{seed}_nms_val"] = nms.reshape(-1)[idx]
        out[f"sp_{h}_{w}_{seed}_desc"] = np.ascontiguousarray(desc[:, ::hf_cases.DESC_STRIDE, ::hf_cases.DESC_STRIDE])
    lg = weights.synthetic_lightglue(1234)
    for n0, n1, seed in hf_cases.LG_PAIRS:
        _, _, a, b = hf_cases.lg_input(n0, n1, seed)
        out[f"lg_{n0}_{n1}_{seed}"] = hf_pin.lightglue_scores(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
    sg = weights.synthetic_superglue(1234)
    for n0, n1, seed in hf_cases.SG_PAIRS:
        _, _, a, b = hf_cases.sg_input(n0, n1, seed)
        out[f"sg_{n0}_{n1}_{seed}"] = hf_pin.superglue_scores(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:])
    path = os.path.join(ROOT, "tests", "golden", "hf_pin.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
