"""airfe_stereo_keyframe: ONE call for what map_builder.cc:85-86 does in two façade calls (Detect(left, right, ..., junctions) = PLNet::infer twice,
feature_detector.cc:97-108, then MatchingPoints) — both images as one detector batch, the line path beside LightGlue.  Per image and per pair it must
return the bits the separate batch-1 entries return: it is a different QUEUE of the same kernels, not a different computation."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ctx(W, H, **kw):
    return api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"),
                       lightglue=weights.synthetic_lightglue(1234), max_batch=2, enc_chunk=2, max_keypoints=400, image_width=W, image_height=H,
                       precision=1, matcher_precision=1, **kw)


@pytest.mark.parametrize("W,H,seed", [(752, 480, 1000), (752, 480, 1003), (640, 480, 7)])
def test_one_call_equals_the_three_calls(W, H, seed):
    left, right = synth.stereo_pair(H, W, seed)
    ctx = _ctx(W, H)
    det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, W, H, 0)
    ll, lr = [], []
    ok, fl, jl = det.DetectLines(left, None, ll, junction_detection=True)
    ok2, fr, _ = det.DetectLines(right, None, lr, junction_detection=False)
    n, matches = pm.MatchingPoints(fl, fr)
    assert ok and ok2 and n > 50 and len(ll) >= 50 and len(lr) >= 50 and jl.shape[1] >= 50
    for rep in range(2):                                                           # (twice: the second call reuses every block the first one grew)
        k = ctx.stereo_keyframe(left, right)
        np.testing.assert_array_equal(k["featL"], fl.T)
        np.testing.assert_array_equal(k["featR"], fr.T)
        np.testing.assert_array_equal(k["linesL"], np.array(ll))
        np.testing.assert_array_equal(k["linesR"], np.array(lr))
        np.testing.assert_array_equal(k["juncL"], jl.T)
        np.testing.assert_array_equal(k["idx"], np.array([(m[0], m[1]) for m in matches], np.int32))
        np.testing.assert_array_equal((np.float32(1.0) - k["score"]).astype(np.float32), np.array([m[2] for m in matches], np.float32))
    # the two-call form (the 7-argument Detect overload + MatchingPoints)
    l2, r2 = [], []
    ok, fl2, fr2, jl2 = det.DetectKeyframe(left, right, l2, r2)
    assert ok and l2 == ll and r2 == lr
    np.testing.assert_array_equal(fl2, fl)
    np.testing.assert_array_equal(fr2, fr)
    np.testing.assert_array_equal(jl2, jl)
    # without junction detection: same points and lines
    k = ctx.stereo_keyframe(left, right, match=False, want_junctions=False)
    assert "idx" not in k and len(k["juncL"]) == 0
    np.testing.assert_array_equal(k["featL"], fl.T)
    np.testing.assert_array_equal(k["linesR"], np.array(lr))
    ctx.close()


def test_strided_views_and_argument_errors():
    W, H = 752, 480
    left, right = synth.stereo_pair(H, W, 1001)
    ctx = _ctx(W, H)
    want = ctx.stereo_keyframe(left, right)
    big = np.zeros((H, 2 * W + 64), np.uint8)                                      # both images as views into one wider buffer (cv::Mat ROIs)
    big[:, :W] = left
    big[:, W + 64:] = right
    got = ctx.stereo_keyframe(big[:, :W], big[:, W + 64:])
    for key in want:
        np.testing.assert_array_equal(got[key], want[key])
    with pytest.raises(api.AirfeError):
        ctx.stereo_keyframe(left, right[:-8])
    with pytest.raises(api.AirfeError):
        ctx.stereo_keyframe(left, np.zeros((0, 0), np.uint8))
    with pytest.raises(api.AirfeError, match="do not fit"):
        ctx.stereo_keyframe(left, right, cap_lines=8)
    # and the context is usable afterwards
    again = ctx.stereo_keyframe(left, right)
    np.testing.assert_array_equal(again["idx"], want["idx"])
    ctx.close()


def test_needs_an_arena_of_two_images():
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=1, max_keypoints=400,
                      image_width=752, image_height=480, precision=1)
    left, right = synth.stereo_pair(480, 752, 1000)
    with pytest.raises(api.AirfeError, match="arena of two images"):
        ctx.stereo_keyframe(left, right, match=False)
    ctx.close()


def test_graph_replay_gives_the_same_bits():
    """airfe_tuning::kf_graph = 1: call 1 runs plainly, call 2 captures the queue as a hipGraph, later calls replay it — on other images, with other entries of the same
    context in between (their host-side flags must not leak into the replay, nor the replay's into them)."""
    W, H = 752, 480
    pairs = [synth.stereo_pair(H, W, 1000 + i) for i in range(3)]
    plain = _ctx(W, H)
    want = [plain.stereo_keyframe(*p) for p in pairs]
    want_pts = plain.detect_points(pairs[1][0])
    plain.close()
    ctx = _ctx(W, H, tuning={"kf_graph": 1})
    for i in range(7):
        got = ctx.stereo_keyframe(*pairs[i % 3])
        for key in want[i % 3]:
            np.testing.assert_array_equal(got[key], want[i % 3][key], err_msg=f"call {i} {key}")
        if i in (2, 4):
            np.testing.assert_array_equal(ctx.detect_points(pairs[1][0]), want_pts)
    # another configuration (no junctions, no match) re-captures
    for i in range(3):
        got = ctx.stereo_keyframe(*pairs[i], match=False, want_junctions=False)
        np.testing.assert_array_equal(got["featR"], want[i]["featR"])
        np.testing.assert_array_equal(got["linesL"], want[i]["linesL"])
    ctx.close()


def test_track_frame_equals_detect_plus_match():
    """airfe_track_frame ≙ map_builder.cc:94-101: Detect(image, features) + MatchingPoints(features_last_keyframe, features) as one call, the keyframe's
    features kept on the device between frames."""
    W, H = 752, 480
    ctx = _ctx(W, H)
    det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, W, H, 0)
    key, _ = synth.stereo_pair(H, W, 1000)
    frames = [synth.stereo_pair(H, W, 1000)[1], synth.stereo_pair(H, W, 1001)[0], synth.stereo_pair(H, W, 1000)[0]]
    ok, fk = det.Detect(key)
    assert ok and fk.shape[1] >= 100
    for i, img in enumerate(frames):
        ok, f = det.Detect(img)
        n, matches = pm.MatchingPoints(fk, f)
        feat, idx, sc = ctx.track_frame(img, ref_feat=fk.T if i == 0 else None)      # reference uploaded once, then kept on the device
        np.testing.assert_array_equal(feat, f.T)
        np.testing.assert_array_equal(idx, np.array([(m[0], m[1]) for m in matches], np.int32).reshape(-1, 2))
        np.testing.assert_array_equal((np.float32(1.0) - sc).astype(np.float32), np.array([m[2] for m in matches], np.float32))
        assert i == 1 or n > 50                                                      # (frame 1 is another scene: few or no matches is fine)
    # a new keyframe replaces the reference; an empty reference gives no matches (point_matcher.cc:53-55)
    ok, f2 = det.Detect(frames[1])
    feat, idx, sc = ctx.track_frame(frames[1], ref_feat=f2.T)
    assert len(feat) >= len(idx) >= 100 and (idx[:, 0] == idx[:, 1]).all()          # a frame against itself: every match is (i, i)
    feat, idx, sc = ctx.track_frame(frames[1], ref_feat=np.zeros((0, 259), np.float32))
    assert len(idx) == 0 and len(feat) >= 100
    ctx.close()
    fresh = _ctx(W, H)
    with pytest.raises(api.AirfeError, match="no reference features"):
        fresh.track_frame(key)
    fresh.close()


def test_no_keypoints_at_all():
    """a detector that finds nothing (threshold above every score): the one-call entries return empty features, no matches (point_matcher.cc:53-55),
    whatever lines there are — and do not hang on zero-length sequences"""
    W, H = 752, 480
    left, right = synth.stereo_pair(H, W, 1000)
    ctx = _ctx(W, H, keypoint_threshold=2.0)
    k = ctx.stereo_keyframe(left, right)
    assert len(k["featL"]) == 0 and len(k["featR"]) == 0 and len(k["idx"]) == 0 and len(k["score"]) == 0
    feat, idx, sc = ctx.track_frame(left, ref_feat=np.zeros((0, 259), np.float32))
    assert len(feat) == 0 and len(idx) == 0
    ctx.close()
    ok_ctx = _ctx(W, H)
    want = ok_ctx.stereo_keyframe(left, right)
    feat, idx, sc = ok_ctx.track_frame(left, ref_feat=want["featL"])
    np.testing.assert_array_equal(feat, want["featL"])
    assert len(idx) >= 100 and (idx[:, 0] == idx[:, 1]).all()
    feat, idx, sc = ok_ctx.track_frame(left, ref_feat=np.zeros((0, 259), np.float32))     # an empty reference: features, no matches
    assert len(feat) == len(want["featL"]) and len(idx) == 0
    ok_ctx.close()


def test_one_call_entries_are_deterministic_over_many_calls():
    """two streams, copies beside kernels, blocks reused from call to call: 300 keyframes and 300 tracked frames, alternating between two image pairs,
    must return the first call's bits every time"""
    W, H = 752, 480
    ctx = _ctx(W, H)
    pairs = [synth.stereo_pair(H, W, 1000), synth.stereo_pair(H, W, 1002)]
    want = [ctx.stereo_keyframe(*p) for p in pairs]
    want_t = [ctx.track_frame(p[1], ref_feat=want[0]["featL"]) for p in pairs]
    for i in range(300):
        k = ctx.stereo_keyframe(*pairs[i & 1])
        for key in want[i & 1]:
            assert np.array_equal(k[key], want[i & 1][key]), (i, key)
        t = ctx.track_frame(pairs[i & 1][1], ref_feat=want[0]["featL"] if i % 7 == 0 else None)
        for a, b in zip(t, want_t[i & 1]):
            assert np.array_equal(a, b), i
    ctx.close()


def test_rows_beyond_the_speculative_copy_take_the_second_round_trip():
    """the entry copies the first 1024 line rows per image and 512 junction rows back before it knows the counts; more than that (here: more than 64,
    airfe_tuning::kf_spec_rows) is fetched in a second round trip — same results either way"""
    W, H = 752, 480
    left, right = synth.stereo_pair(H, W, 1003)
    ctx = _ctx(W, H)
    want = ctx.stereo_keyframe(left, right)
    ctx.close()
    assert len(want["linesL"]) > 64 and len(want["juncL"]) > 64
    ctx = _ctx(W, H, tuning={"kf_spec_rows": 64})
    for _ in range(2):
        got = ctx.stereo_keyframe(left, right)
        for key in want:
            np.testing.assert_array_equal(got[key], want[key])
    ctx.close()


def test_keyframe_with_the_temporal_match_in_the_same_forward():
    """airfe_stereo_keyframe_tracked: map_builder.cc:85-86 and :96 — the stereo pair (left, right) and the temporal pair (last keyframe, left) are two
    independent MatchingPoints calls; here they are pairs 0 and 1 of ONE LightGlue forward.  Per pair: the bits of the separate calls."""
    W, H = 752, 480
    ctx = _ctx(W, H)
    det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, W, H, 0)
    key_l, _ = synth.stereo_pair(H, W, 1000)
    ok, fk = det.Detect(key_l)                                                       # "the last keyframe"
    for i, seed in enumerate((1000, 1002, 1000)):
        left, right = synth.stereo_pair(H, W, seed)
        want = ctx.stereo_keyframe(left, right)
        nt, tmatches = pm.MatchingPoints(fk, np.asfortranarray(want["featL"].T))
        got = ctx.stereo_keyframe(left, right, track=True, ref_feat=fk.T if i == 0 else None)
        for key in want:
            np.testing.assert_array_equal(got[key], want[key], err_msg=key)
        np.testing.assert_array_equal(got["track_idx"], np.array([(m[0], m[1]) for m in tmatches], np.int32).reshape(-1, 2))
        np.testing.assert_array_equal((np.float32(1.0) - got["track_score"]).astype(np.float32), np.array([m[2] for m in tmatches], np.float32))
        assert i == 1 or nt > 50
    # the reference stays on the device for airfe_track_frame too
    feat, idx, sc = ctx.track_frame(key_l)
    assert len(idx) >= 100 and (idx[:, 0] == idx[:, 1]).all()
    ctx.close()
    fresh = _ctx(W, H)
    with pytest.raises(api.AirfeError, match="no reference features"):
        fresh.stereo_keyframe(key_l, key_l, track=True)                              # nothing was ever uploaded
    with pytest.raises(api.AirfeError, match="exceeds max_keypoints"):
        fresh.stereo_keyframe(key_l, key_l, track=True, ref_feat=np.zeros((401, 259), np.float32))
    got = fresh.stereo_keyframe(key_l, key_l, track=True, ref_feat=np.zeros((0, 259), np.float32))      # an empty keyframe: no temporal matches
    assert len(got["track_idx"]) == 0 and len(got["idx"]) >= 100
    fresh.close()
    small = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), lightglue=weights.synthetic_lightglue(1234),
                        max_batch=1, max_keypoints=400, image_width=W, image_height=H, precision=1, matcher_precision=1)
    with pytest.raises(api.AirfeError, match="max_batch >= 2"):
        small.stereo_keyframe(key_l, key_l, track=True, ref_feat=fk.T)
    small.close()
