"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/); only tests/ and bench.py's cpu_baseline leg use it.

CPU restatement (fp32 PyTorch networks of ref_nets + the numpy routines of ref_post / ref_chain) of what `MapBuilder::ExtractFeatureThread` runs per
stereo frame, src/map_builder.cc:55-147, with `use_superpoint: 1` as every shipped configuration has it (configs/visual_odometry/*.yaml:2):

    :83-92   keyframe candidate: Detect(l, r, ..., lines, junctions) = PLNet::infer x2 (feature_detector.cc:97-108) + MatchingPoints(l, r)
    :93-97   normal frame:       Detect(left, features)              = SuperPoint::infer     (feature_detector.cc:36-41)
    :99-121  temporal MatchingPoints(last keyframe, frame) + AddKeyframeCheck (:429-466) + promotion (:104-108) + insert-next bookkeeping
    :122-130 initialisation
    :139-141 `_last_keyframe_feature = frame`

It is written independently of airslam_amd/seq.py (the product-side driver of the same loop): the two agree only if both read the reference the same way.
`step(..., follow=...)` makes the chain take the branches a device run took, so that per-frame outputs stay comparable after a decision that sits on a
threshold; the oracle's OWN decision is recorded beside it.

PARITY UNPINNED for the network bodies (ref_nets.py: the ONNX files are absent upstream); the host routines are pinned to the reference's compiled code
(ref_post.py's header).  Not restated: the F-matrix RANSAC behind MatchingPoints(..., true) (src/point_matcher.cc:95-104).
"""
from __future__ import annotations

import numpy as np

from . import ref_chain, ref_nets, ref_post

F = np.float32
NORMAL, KEYFRAME, INIT = 0, 1, 2              # include/map_builder.h:40-44


def add_keyframe_check(ref_feat, cur_feat, idx, min_num_match=30, max_num_match=80, tracking_point_rate=0.65, tracking_parallax_rate=0.1, width=752, height=480):
    """src/map_builder.cc:429-466 without the IMU branch (:438-441)."""
    match_num = len(idx)
    if match_num < min_num_match:                                             # :431
        return 0
    rate = F(tracking_point_rate)
    if F(match_num) / F(ref_feat.shape[0]) < rate or F(match_num) / F(cur_feat.shape[0]) < rate or match_num < max_num_match:   # :443
        return 1
    ref_kp = np.stack([ref_feat[i, 1:3] for i, _ in idx], 1).astype(F)        # Matrix2Xf, columns = matches (:447-455)
    cur_kp = np.stack([cur_feat[j, 1:3] for _, j in idx], 1).astype(F)
    parallax = ref_kp - cur_kp
    average_parallax = float((parallax @ parallax.T).sum(dtype=F)) / match_num      # :458
    if average_parallax > float(height * width) * tracking_parallax_rate * tracking_parallax_rate:     # :459-463
        return 1
    return 2


def add_right_features_count(feat_left, feat_right, idx, min_x_diff=1.0, max_x_diff=200.0, max_y_diff=5.0):
    """the return value of Frame::AddRightFeatures, src/frame.cc:141-172"""
    good = 0
    for il, ir in idx:
        dx = abs(float(F(feat_left[il, 1]) - F(feat_right[ir, 1])))
        dy = abs(float(F(feat_left[il, 2]) - F(feat_right[ir, 2])))
        if not (dx > min_x_diff and dx < max_x_diff and dy <= max_y_diff):     # :153
            continue
        parallax = float(F(feat_left[il, 1]) - F(feat_right[ir, 1]))          # :165
        if parallax < max_x_diff and parallax > min_x_diff:                   # :167
            good += 1
    return good


class Chain:
    def __init__(self, sp_plnet: dict, sp_superpoint: dict, s1: dict, lg: dict, width=752, height=480, max_keypoints=400, policy: dict = None):
        """sp_plnet: the PLNet stage-0 pack (point branch + line.*), sp_superpoint: the SuperPoint pack, s1: stage-1 weights, lg: LightGlue weights;
        policy: keyword overrides of add_keyframe_check / add_right_features_count / min_init_stereo_feature (vo_euroc.yaml:16-22)."""
        self.pl, self.sp, self.s1, self.lg = sp_plnet, sp_superpoint, s1, lg
        self.W, self.H, self.K = width, height, max_keypoints
        p = dict(policy or {})
        self.min_init = p.pop("min_init_stereo_feature", 90)
        self.band = {k: p.pop(k) for k in ("min_x_diff", "max_x_diff", "max_y_diff") if k in p}
        self.kf_args = dict(p, width=width, height=height)
        self.init = False
        self.insert_next = False
        self.ref = None

    # -- the two façade calls
    def superpoint(self, img):
        """SuperPoint::infer, src/super_point.cpp:103-144"""
        x, ws, hs = ref_post.process_image(img)
        heat, desc = ref_nets.superpoint_forward(self.sp, x[None])
        return ref_post.keypoints_decoder(ref_post.simple_nms(heat[0], 4), desc[0], 0.004, 4, self.K, ws, hs)

    def match(self, f0, f1):
        """PointMatcher::MatchingPoints with LightGlue, src/point_matcher.cc:50-107 (without the RANSAC of :95-104) -> (idx [m,2], score [m], log-assignment)"""
        if f0.shape[0] < 1 or f1.shape[0] < 1:                                # :53-55
            return np.zeros((0, 2), np.int32), np.zeros((0,), F), None
        a = ref_post.normalize_keypoints(f0, self.W, self.H, 0.5)
        b = ref_post.normalize_keypoints(f1, self.W, self.H, 0.5)
        s = ref_nets.lightglue_forward(self.lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
        idx, sc = ref_post.filter_matches(s, 0.1)
        return np.asarray(idx, np.int32).reshape(-1, 2), np.asarray(sc, F), s

    def step(self, left, right, follow: dict = None):
        """One frame.  follow = {"candidate": bool, "promoted": bool, "frame_type": int, "dropped": bool} makes the chain take those branches (and adopt that
        frame type); the decisions it would have taken itself are returned as own_*."""
        cand_own = (not self.init) or self.insert_next                        # :83
        cand = follow["candidate"] if follow else cand_own
        out = dict(candidate=cand, own_candidate=cand_own, promoted=False, dropped=False, enough_match=-1, good_stereo_point=0, scores_t=None, scores_s=None)
        if cand:
            L = ref_chain.plnet_infer(self.pl, self.s1, left, want_junctions=True, top_k=self.K)       # feature_detector.cc:100
            R = ref_chain.plnet_infer(self.pl, self.s1, right, want_junctions=False, top_k=self.K)     # :101
            out.update(features_left=L["features"], features_right=R["features"], lines_left=L["lines"], lines_right=R["lines"], junctions=L["junctions"])
            out["stereo_idx"], out["stereo_score"], out["scores_s"] = self.match(L["features"], R["features"])     # map_builder.cc:86
            out["good_stereo_point"] = add_right_features_count(L["features"], R["features"], out["stereo_idx"], **self.band)
            frame_type = KEYFRAME if self.init else INIT                      # :88
        else:
            out["features_left"] = self.superpoint(left)                      # :94
            frame_type = NORMAL
        own_type = frame_type
        if self.init:
            out["matches_idx"], out["matches_score"], out["scores_t"] = self.match(self.ref, out["features_left"])     # :100-101
            out["enough_match"] = em = add_keyframe_check(self.ref, out["features_left"], out["matches_idx"], **self.kf_args)
            promote_own = em == 0 and frame_type == NORMAL
            promote = follow["promoted"] if follow else promote_own
            out["own_promoted"] = promote_own
            if promote:                                                       # :104-109
                fr = self.superpoint(right)
                out["features_right"] = fr
                out["stereo_idx"], out["stereo_score"], out["scores_s"] = self.match(out["features_left"], fr)
                out["good_stereo_point"] = add_right_features_count(out["features_left"], fr, out["stereo_idx"], **self.band)
                out["promoted"] = True
            if em == 0:
                if out["good_stereo_point"] < 10:                             # :111-117
                    insert_next, own_type = True, NORMAL
                else:
                    insert_next, own_type = False, KEYFRAME
            else:
                insert_next = (em == 1) and (frame_type == NORMAL)            # :119
        else:
            insert_next = self.insert_next
            if out["good_stereo_point"] < self.min_init:                      # :122-125
                out["dropped"] = True
        out["own_frame_type"] = own_type
        out["own_dropped"] = out["dropped"]
        if follow:                                                            # take the device's decision from here on
            own_type, out["dropped"] = follow["frame_type"], follow["dropped"]
            insert_next = follow.get("insert_next", insert_next)
        out["frame_type"] = own_type
        if out["dropped"]:
            return out
        if not self.init:
            self.init = True                                                  # :127-128
        self.insert_next = insert_next
        if own_type != NORMAL:
            self.ref = out["features_left"]                                   # :139-141
        return out
