"""The oracle pinned to the REFERENCE'S OWN CODE (missing #1).

oracle/_ref/libairslam_ref.so is the reference's front end compiled UNCHANGED from /root/reference (oracle/Makefile: feature_detector.cc,
point_matcher.cc, plnet.cpp, super_point.cpp, light_glue.cpp, super_glue.cpp, line_processor.cc:1-180; TensorRT engines = a callback).
tests/golden/ref_pin.npz holds ITS outputs on the seeded cases of tests/ref_cases.py (tools/make_ref_fixtures.py).

  * test_restatement_equals_the_reference_fixtures   oracle/ref_post.py == fixtures, bit for bit (everywhere, no library needed)
  * test_restatement_equals_the_live_reference       the same against the library itself, incl. what the engines were fed
  * test_fixtures_regenerate                         the committed file is what the library produces today
Bit for bit means: every index, coordinate, score, line, distance AND every descriptor (the summation order of Eigen's normalize() is restated
on both sides, shim/stubs/Eigen/Core / ref_post._eigen_sse2_sum); the one licence is the order of EQUAL scores behind std::sort (unstable)."""
import os

import numpy as np
import pytest

import ref_cases as rc
from conftest import GOLDEN, ROOT
from oracle import ref_lib

FIX = os.path.join(GOLDEN, "ref_pin.npz")
CASES = [(fam, name) for fam, (table, _) in rc.FAMILIES.items() for name in table]
needs_ref = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref is not built (no /root/reference here and no prebuilt library)")


def _fixture(fam, name):
    z = np.load(FIX)
    pre = f"{fam}/{name}/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


@pytest.mark.parametrize("fam,name", CASES)
def test_restatement_equals_the_reference_fixtures(fam, name):
    case = rc.FAMILIES[fam][1](name)
    want = _fixture(fam, name)
    assert want, "no fixture for this case: run tools/make_ref_fixtures.py"
    got = rc.run_post(fam, case)
    rc.assert_same(fam, name, got, want, case)


@needs_ref
@pytest.mark.parametrize("fam,name", CASES)
def test_restatement_equals_the_live_reference(fam, name, tmp_path):
    case = rc.FAMILIES[fam][1](name)
    want = rc.run_ref(fam, case, str(tmp_path))
    got = rc.run_post(fam, case)
    rc.assert_same(fam, name, got, want, case)
    if fam == "detect":          # process_image's `float(px) / 255.0` (double division, narrowed) on the stand-in resize
        assert np.array_equal(got["fed_input"], want["fed_input"])


@needs_ref
def test_fixtures_regenerate():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_ref_fixtures", os.path.join(ROOT, "tools", "make_ref_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    new, old = m.generate(), np.load(FIX)
    assert sorted(new) == sorted(old.files)
    for k in old.files:
        if k.endswith("ties_above_top_k/feat"):                      # std::sort's order of equal scores is the library's business
            assert np.array_equal(new[k][:, 0], old[k][:, 0])
            continue
        assert np.array_equal(new[k], old[k], equal_nan=new[k].dtype.kind == "f"), k


@needs_ref
def test_the_library_is_the_references_code_not_ours():
    """The recipe compiles the reference's sources where they lie: nothing of /root/reference is copied into the repository, the generated
    line-range extract and the header farm live under the git-ignored oracle/_ref/."""
    assert "src/plnet.cpp" in ref_lib.sources() and "src/line_processor.cc:1-180" in ref_lib.sources()
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    for src in ("feature_detector.cc", "point_matcher.cc", "plnet.cpp", "super_point.cpp", "light_glue.cpp", "super_glue.cpp"):
        assert src in mk
    assert "$(REF)/src/%.cpp" in mk and "$(REF)/src/%.cc" in mk                      # compiled from the reference tree itself
    ign = open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref/" in ign
    assert "oracle/_ref/" not in open(os.path.join(ROOT, ".gpurunignore")).read().split()   # the built library travels to the GPU box
    inc = os.path.join(ROOT, "oracle", "_ref", "inc")
    if os.path.isdir(inc):
        for h in ("plnet.h", "feature_detector.h", "point_matcher.h", "read_configs.h"):
            assert os.path.realpath(os.path.join(inc, h)).startswith("/root/reference/include/"), h


@needs_ref
def test_reference_reads_its_own_config_files(tmp_path):
    """include/read_configs.h (unchanged) parses the reference's own configs/visual_odometry/vo_euroc.yaml through the yaml stand-in: the
    thresholds the parity tests use are the reference's, read by the reference's code."""
    y = "/root/reference/configs/visual_odometry/vo_euroc.yaml"
    if not os.path.exists(y):
        pytest.skip("no reference tree")
    ref_lib.set_engines({})
    det = ref_lib.FeatureDetector(str(tmp_path / "m"), yaml=y)
    assert det.cfg == dict(use_superpoint=1, max_keypoints=400, keypoint_threshold=pytest.approx(0.004), remove_borders=4,
                           line_threshold=pytest.approx(0.75), line_length_threshold=pytest.approx(50.0))
    det.close()


@needs_ref
def test_all_six_detect_overloads_of_the_reference(tmp_path):
    """src/feature_detector.cc:36-105 on constant engines: which network each overload runs, left gets junctions / right does not (:100-101),
    lines are appended, `good_infer_left & good_infer_right`, an empty image fails."""
    case = rc.plnet_case("typical")
    sp = rc.detect_case("dense_752x480")
    fh = rc.R // 4
    z = lambda *s: np.zeros(s, np.float32)
    log = []
    eng = {"superpoint": lambda ins: (log.append("sp"), dict(scores=sp["heat"], descriptors=sp["desc"]))[1],
           "plnet_s0": lambda ins: (log.append("s0"), dict(scores=case["heat"], descriptors=case["desc"], juncs_pred=case["juncs_pred"],
                                                           lines_pred=case["lines_pred"], iskeep=case["iskeep"], idx_junc_to_end_min=case["idx_min"],
                                                           idx_junc_to_end_max=case["idx_max"], loi_features=z(1, 128, fh, fh),
                                                           loi_features_thin=z(1, 4, fh, fh), loi_features_aux=z(1, 4, fh, fh)))[1],
           "plnet_s1": lambda ins: (log.append("s1"), dict(zip(("lines_adjusted", "scores_line"),
                                                                rc.stage1_stub(ins["juncs_pred"], ins["idx_lines_for_junctions"], case["seed"]))))[1]}
    ref_lib.set_engines(eng)
    img = case["image"]
    for use_sp in (0, 1):
        det = ref_lib.FeatureDetector(str(tmp_path / "m"), use_superpoint=use_sp)
        want_pts = 400 if use_sp else len(rc.run_post("plnet", case)["feat"])
        for ov, nets in ((0, ["sp"] if use_sp else ["s0", "s1"]), (1, ["s0", "s1"]), (2, ["s0", "s1"]),
                         (3, ["sp", "sp"] if use_sp else ["s0", "s1"] * 2), (4, ["s0", "s1"] * 2), (5, ["s0", "s1"] * 2)):
            log.clear()
            r = det.detect(ov, img, img if ov >= 3 else None, lines_in=np.array([[9.0, 9.0, 9.0, 9.0]]))
            assert r["ok"] and log == nets, (ov, log)
            assert len(r["feat_l"]) == (want_pts if ov in (0, 3) else len(rc.run_post("plnet", case)["feat"]))
            assert (len(r["feat_r"]) > 0) == (ov >= 3)
            assert (len(r["junc"]) > 0) == (ov in (2, 5))                               # junctions: left image only, only when asked
            if ov in (1, 2, 4, 5):
                assert len(r["lines_l"]) > 1 and r["lines_l"][0, 0] != 9.0              # appended, and the caller's line is rescaled with the rest
            if ov >= 4:
                assert np.array_equal(r["lines_r"], r["lines_l"][1:])
        assert not det.detect(1, img[:0])["ok"]                                          # empty image -> false (src/plnet.cpp:247)
        det.close()
