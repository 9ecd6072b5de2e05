"""Generate the committed golden fixtures under tests/golden/ from the REAL reference artefact
/root/reference/output/plnet_s1.onnx (the only model file present in the reference checkout).

Run in the build container only (the GPU box has no /root/reference):
    python tools/make_golden.py
Outputs
    tests/golden/plnet_s1.airfe        real stage-1 weights re-packed as an airfe weight pack (fc2.* tensors)
    tests/golden/plnet_s1_golden.npz   outputs of the real ONNX graph (numpy interpreter oracle/onnx_run.py)
                                       on seeded synthetic stage-0 tensors (airslam_amd.synth.plnet_stage0_lines)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from airslam_amd import onnx_lite, synth, weights  # noqa: E402
from oracle import onnx_run, ref_post  # noqa: E402

ONNX = "/root/reference/output/plnet_s1.onnx"
CASES = [(5, 300), (6, 1500), (7, 40), (8, 1)]


def main():
    m = onnx_lite.load(ONNX)
    w = {k: v for k, v in m.initializers.items() if k.startswith("fc2")}
    w["sample_t"] = m.initializers["onnx::Mul_1141"].reshape(-1)
    assert np.array_equal(w["sample_t"], weights.linspace_t())
    assert np.array_equal(m.initializers["onnx::Mul_1142"].reshape(-1), np.float32(1) - w["sample_t"])
    weights.check_spec(w, weights.plnet_s1_spec())
    gd = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gd, exist_ok=True)
    weights.save_pack(os.path.join(gd, "plnet_s1.airfe"), w)
    out = {}
    for seed, nl in CASES:
        s0 = synth.plnet_stage0_lines(seed, n_lines=nl)
        keep, inv, pairs = ref_post.wireframe_matcher(s0["iskeep"], s0["idx_junc_to_end_min"], s0["idx_junc_to_end_max"])
        feeds = dict(juncs_pred=s0["juncs_pred"], lines_pred=s0["lines_pred"],
                     idx_lines_for_junctions=pairs.astype(np.float32), inverse=inv.astype(np.float32)[:, None],
                     iskeep_index=keep.astype(np.float32)[:, None], loi_features=s0["loi_features"],
                     loi_features_thin=s0["loi_features_thin"], loi_features_aux=s0["loi_features_aux"])
        r = onnx_run.run(m, feeds)
        out[f"s{seed}_n{nl}_lines_adjusted"] = r["lines_adjusted"]
        out[f"s{seed}_n{nl}_scores_line"] = r["scores_line"]
    np.savez_compressed(os.path.join(gd, "plnet_s1_golden.npz"), **out)
    print("wrote", gd)


if __name__ == "__main__":
    main()
