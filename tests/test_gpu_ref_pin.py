"""The HIP path held to the REFERENCE'S OWN CODE (VERDICT r03 "next round" #1: the route from parity 'partial' to 'green' for the index rows).

oracle/_ref/libairslam_ref.so = /root/reference's front-end sources compiled unchanged (oracle/Makefile); it travels to the GPU box with the
snapshot.  Two kinds of test:

  LIVE (skipped when the library did not travel): the device's own engine tensors — NMS'd heat map, dense descriptors, the ten stage-0 tensors,
  the stage-1 outputs, the LightGlue / SuperGlue score matrices — are read back and handed, as the "TensorRT outputs", to the reference's
  FeatureDetector::Detect / PointMatcher::MatchingPoints; what the reference's host code makes of them must be what the device returned:
  keypoints (score, x, y), lines (float64), junctions, match lists EXACT; sampled descriptors within 2e-6 (summation order of two fp32 norms).

  FIXTURES (always): tests/golden/ref_pin.npz holds the reference's outputs on the seeded cases of tests/ref_cases.py; the kernels that can
  be fed host tensors (filter_matches, decode, AssignPointsToLines, MatchLines, NormalizeKeypoints) must reproduce them bit for bit."""
import os

import numpy as np
import pytest

import ref_cases as rc
from airslam_amd import api, synth, weights
from conftest import GOLDEN
from gpu_common import diag
from oracle import ref_lib

pytestmark = pytest.mark.gpu
live = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libairslam_ref.so did not travel to this machine")
FIX = os.path.join(GOLDEN, "ref_pin.npz")
S1 = os.path.join(GOLDEN, "plnet_s1.airfe")
_C = {}


def _fixture(fam, name):
    z = np.load(FIX)
    pre = f"{fam}/{name}/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def _ctx(kind, **kw):
    key = (kind, tuple(sorted(kw.items())))
    if key not in _C:
        if kind == "plnet":
            _C[key] = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=S1, max_batch=2, enc_chunk=2, **kw)
        elif kind == "sp":
            _C[key] = api.Context(superpoint=weights.synthetic_superpoint(1234), max_batch=2, enc_chunk=2, **kw)
        elif kind == "lg":
            _C[key] = api.Context(lightglue=weights.synthetic_lightglue(1234), max_batch=1, max_keypoints=1024, **kw)
        elif kind == "sg":
            _C[key] = api.Context(superglue=weights.synthetic_superglue(1234), matcher=1, max_batch=1, max_keypoints=1024, **kw)
    return _C[key]


# ======================================================================================================== LIVE: PLNet::infer, whole host chain
@live
@pytest.mark.parametrize("h,w,seed,lt,ll", [(480, 752, 0, 0.75, 50.0), (480, 640, 3, 0.5, 20.0), (720, 1280, 5, 0.75, 50.0)])
def test_plnet_infer_equals_the_reference_on_the_devices_own_tensors(tmp_path, h, w, seed, lt, ll):
    ctx = _ctx("plnet", line_threshold=lt, line_length_threshold=ll)
    img = synth.gabor_image(h, w, seed)
    feat, lines, junc = ctx.detect_plnet(img, None, want_junctions=True)
    _, nms, desc = ctx.detector_maps(1)
    s0 = ctx.debug_plnet_stage0()
    la, sc = ctx.debug_plnet_s1(s0)
    seen = {}

    def eng_s0(ins):
        return dict(scores=nms[0], descriptors=np.ascontiguousarray(desc[0].transpose(2, 0, 1)), juncs_pred=s0["juncs_pred"],
                    lines_pred=s0["lines_pred"], iskeep=s0["iskeep"], idx_junc_to_end_min=s0["idx_junc_to_end_min"],
                    idx_junc_to_end_max=s0["idx_junc_to_end_max"], loi_features=s0["loi_features"], loi_features_thin=s0["loi_features_thin"],
                    loi_features_aux=s0["loi_features_aux"])

    def eng_s1(ins):
        # the reference's wireframe_matcher (src/plnet.cpp:272-307) ran on the device's iskeep / idx maps: its unique pairs, in its order,
        # must be the device's — then the device's stage-1 outputs are the engine's outputs line for line
        pairs = ins["idx_lines_for_junctions"].astype(np.int64)
        seen["m2"] = len(pairs)
        seen["la_ref"] = np.concatenate([s0["juncs_pred"][pairs[:, 0]], s0["juncs_pred"][pairs[:, 1]]], 1) if len(pairs) else np.zeros((0, 4), np.float32)
        assert len(pairs) == len(la)
        return dict(lines_adjusted=la, scores_line=sc)

    ref_lib.set_engines({"plnet_s0": eng_s0, "plnet_s1": eng_s1})
    det = ref_lib.FeatureDetector(str(tmp_path / "m"), use_superpoint=0, max_keypoints=400, keypoint_threshold=0.004, remove_borders=4,
                                  line_threshold=lt, line_length_threshold=ll)
    r = det.detect(2, img)
    det.close()
    assert r["ok"]
    dd = float(np.abs(r["feat_l"][:, 3:] - feat[:, 3:]).max()) if r["feat_l"].shape == feat.shape else -1.0
    dj = float(np.abs(r["junc"][:, 3:] - junc[:, 3:]).max()) if r["junc"].shape == junc.shape and len(junc) else 0.0
    diag(f"refpin_plnet_{w}x{h}_{seed}", n_points=len(feat), n_lines=len(lines), n_junc=len(junc), m2=seen.get("m2", -1),
         ref_points=len(r["feat_l"]), ref_lines=len(r["lines_l"]), ref_junc=len(r["junc"]), desc_maxdiff=dd, junc_desc_maxdiff=dj)
    np.testing.assert_array_equal(seen["la_ref"], la)                           # wireframe_matcher + the stage-1 gather: exact
    assert len(lines) >= 20 and len(junc) >= 10 and len(feat) >= 100
    np.testing.assert_array_equal(r["feat_l"][:, :3], feat[:, :3])              # detect_point + rescale: exact, same order
    assert dd <= rc.DESC_TOL
    np.testing.assert_array_equal(r["lines_l"], lines)                          # line filter + rescale: exact doubles
    np.testing.assert_array_equal(r["junc"][:, :3], junc[:, :3])                # junction map + junction_detector: exact
    assert dj <= rc.DESC_TOL


@live
@pytest.mark.parametrize("h,w,seed", [(480, 752, 1), (480, 640, 2)])
def test_superpoint_infer_equals_the_reference_on_the_devices_own_maps(tmp_path, h, w, seed):
    ctx = _ctx("sp")
    img = synth.gabor_image(h, w, seed)
    feat = ctx.detect_points(img)
    _, nms, desc = ctx.detector_maps(1)
    calls = ref_lib.set_engines({"superpoint": lambda ins: dict(scores=nms[0], descriptors=np.ascontiguousarray(desc[0].transpose(2, 0, 1)))})
    det = ref_lib.FeatureDetector(str(tmp_path / "m"), use_superpoint=1)
    r = det.detect(0, img)
    det.close()
    assert r["ok"] and len(feat) >= 100
    np.testing.assert_array_equal(r["feat_l"][:, :3], feat[:, :3])
    assert float(np.abs(r["feat_l"][:, 3:] - feat[:, 3:]).max()) <= rc.DESC_TOL
    # what the reference's process_input fed its engine = its cv::resize (a stand-in) + `float(px) / 255.0`; the device's pre-process agrees
    np.testing.assert_array_equal(calls[0][1]["input"][0, 0], ctx.debug_preprocess(img))


# ======================================================================================================== LIVE: MatchingPoints
@live
@pytest.mark.parametrize("matcher,seed", [(0, 4), (0, 9), (1, 4)])
def test_matching_points_equals_the_reference_on_the_devices_own_scores(tmp_path, matcher, seed):
    sp = _ctx("sp")
    left, right = synth.stereo_pair(480, 752, seed)
    f0, f1 = sp.detect_points(left), sp.detect_points(right)
    ctx = _ctx("sg" if matcher else "lg")
    pm = api.PointMatcher(ctx, 752, 480, matcher)
    cnt, matches = pm.MatchingPoints(np.asfortranarray(f0.T), np.asfortranarray(f1.T))
    n0 = api.PointMatcher.NormalizeKeypoints(np.asfortranarray(f0.T), 752, 480, 0.7 if matcher else 0.5)
    n1 = api.PointMatcher.NormalizeKeypoints(np.asfortranarray(f1.T), 752, 480, 0.7 if matcher else 0.5)
    if matcher:
        scores = ctx.superglue_scores(np.ascontiguousarray(n0.T), np.ascontiguousarray(n1.T))
    else:
        scores = ctx.lightglue_scores(np.ascontiguousarray(n0[1:].T), np.ascontiguousarray(n1[1:].T))
    calls = ref_lib.set_engines({"superglue" if matcher else "lightglue": lambda ins: dict(scores=scores)})
    rpm = ref_lib.PointMatcher(str(tmp_path / "m"), matcher, 752, 480)
    rcnt, q, t, d = rpm.matching_points(f0, f1)
    rpm.close()
    diag(f"refpin_match_{matcher}_{seed}", dev=cnt, ref=int(rcnt))
    assert cnt >= 50 and rcnt == cnt
    np.testing.assert_array_equal(q, np.array([m[0] for m in matches], np.int32))
    np.testing.assert_array_equal(t, np.array([m[1] for m in matches], np.int32))
    from conftest import host_expf_is_the_restated_glibc_routine
    if host_expf_is_the_restated_glibc_routine():          # (the compiled reference links THIS host's libm; on another expf build the distances are within an ulp, not equal)
        np.testing.assert_array_equal(d, np.array([m[2] for m in matches], np.float32))   # 1 - exp(score): the device's exp == glibc's expf here
    else:
        np.testing.assert_allclose(d, np.array([m[2] for m in matches], np.float32), atol=2e-7, rtol=0)
    fed = calls[0][1]                                                                       # NormalizeKeypoints + process_input of the reference
    np.testing.assert_array_equal(fed["keypoints_0"][0], n0[1:3].T)
    np.testing.assert_array_equal(fed["keypoints_1"][0], n1[1:3].T)


# ======================================================================================================== FIXTURES
@pytest.mark.parametrize("name", list(rc.MATCH))
def test_filter_matches_kernel_equals_the_reference_fixtures(name):
    case, want = rc.match_case(name, "lg"), _fixture("lg", name)
    if min(case["scores"].shape) == 0:
        pytest.skip("MatchingPoints returns before the matcher (src/point_matcher.cc:53-55)")
    idx, sc = _ctx("lg").debug_lg_filter(case["scores"])
    np.testing.assert_array_equal(idx[:, 0], want["query"])
    np.testing.assert_array_equal(idx[:, 1], want["train"])
    np.testing.assert_array_equal((1.0 - sc.astype(np.float64)).astype(np.float32), want["distance"])
    nk = api.PointMatcher.NormalizeKeypoints(np.asfortranarray(case["f0"].T), case["width"], case["height"], 0.5)
    np.testing.assert_array_equal(nk[:4].T, want["norm0_head"])


@pytest.mark.parametrize("name", list(rc.MATCH))
def test_decode_kernel_equals_the_reference_fixtures(name):
    case, want = rc.match_case(name, "sg"), _fixture("sg", name)
    if min(case["scores"].shape) <= 1:
        pytest.skip("MatchingPoints returns before the matcher (src/point_matcher.cc:53-55)")
    i0, i1, m0, m1 = _ctx("sg").debug_sg_decode(case["scores"])
    ms = [(i, int(i0[i]), 1.0 - (m0[i] + m1[i0[i]]) / 2.0) for i in range(len(i0)) if 0 <= i0[i] < len(i1) and i1[i0[i]] == i]   # point_matcher.cc:82-91
    np.testing.assert_array_equal(np.array([m[0] for m in ms], np.int32), want["query"])
    np.testing.assert_array_equal(np.array([m[1] for m in ms], np.int32), want["train"])
    np.testing.assert_array_equal(np.array([m[2] for m in ms], np.float64).astype(np.float32), want["distance"])


@pytest.mark.parametrize("name", list(rc.LINES))
def test_line_kernels_equal_the_reference_fixtures(name):
    case, want = rc.lines_case(name), _fixture("lines", name)
    ctx = _ctx("lg")
    rels = []
    for side in "01":
        rel = ctx.assign_points_to_lines(case["lines" + side], case["feat" + side])
        off = np.cumsum([0] + [len(r) for r in rel]).astype(np.int32)
        np.testing.assert_array_equal(off, want["off" + side])
        np.testing.assert_array_equal(np.array([k for r in rel for k in r], np.int32), want["idx" + side])
        np.testing.assert_array_equal(np.array([r[k] for r in rel for k in r], np.float64), want["dist" + side])
        rels.append(rel)
    lm = ctx.match_lines(rels[0], rels[1], list(zip(case["query"].tolist(), case["train"].tolist())), len(case["feat0"]), len(case["feat1"]))
    np.testing.assert_array_equal(np.array(lm, np.int32), want["line_matches"])


@pytest.mark.parametrize("name", list(rc.BOW))
def test_bow_kernel_equals_the_reference_fixtures(name):
    """airfe_bow_transform against what the vendored DBoW2 + src/bow/FSuperpoint.cc, compiled unchanged, returned for the same tree and features
    (TemplatedVocabulary.h:1313-1352 as Database::FrameToBow calls it, database.cc:57-89) — exact ties and near-ties included."""
    from oracle import ref_post
    case, want = rc.bow_case(name), _fixture("bow", name)
    n = len(case["feat"])
    ctx = api.Context(superpoint=None, max_batch=2, max_keypoints=max(n, 16))
    ctx.bow_load(case["voc"])
    words, w = ctx.bow_transform(case["feat"])
    np.testing.assert_array_equal(words, want["word_of_features"])
    np.testing.assert_array_equal(w, want["weight_of_features"])
    bow, _ = ref_post.frame_to_bow(words, w)
    np.testing.assert_array_equal(np.array(list(bow), np.uint32), want["bow_ids"])
    np.testing.assert_allclose(np.array(list(bow.values())), want["bow_values"], rtol=0, atol=1e-16)
    ctx.close()
