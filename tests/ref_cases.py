"""Seeded inputs for pinning the oracle (oracle/ref_post.py) and the HIP kernels to the REFERENCE'S OWN CODE (oracle/_ref, built by
oracle/Makefile from /root/reference), and the two runners: `run_ref` drives the compiled reference, `run_post` the numpy restatement.

A case is a dict of numpy arrays + scalars that numpy regenerates bit for bit from its seed (PCG64 `random` / `integers` only), so the committed
fixtures (tests/golden/ref_*.npz, written by tools/make_ref_fixtures.py) hold only the case names and the reference's OUTPUTS.

Families (what in the reference each one goes through — all of it compiled unchanged):
  detect   FeatureDetector::Detect(image, features) with use_superpoint = 1 -> SuperPoint::infer -> detect_point, extract_descriptors, rescale
  plnet    FeatureDetector::Detect(image, features, lines, junctions) -> PLNet::infer -> wireframe_matcher, the stage-1 feed, line filter,
           junction map, detect_point, extract_descriptors, junction_detector, rescales
  lg / sg  PointMatcher::MatchingPoints -> NormalizeKeypoints, process_input, filter_matches / decode, the DMatch loops
  lines    AssignPointsToLines, MatchLines
  bow      TemplatedVocabulary::transform per feature + BowVector::addWeight / normalize, as Database::FrameToBow runs them (vendored DBoW2 + FSuperpoint.cc)
"""
from __future__ import annotations

import os

import numpy as np

F = np.float32
R = 512          # the reference's internal resolution (src/plnet.cpp:17-21)


# ================================================================================================================ detect
DETECT = {       # name: (seed, kind, thr, border, top_k, H, W)
    "dense_752x480": (11, "dense", 0.004, 4, 400, 480, 752),
    "sparse_raster_order": (12, "sparse", 0.004, 4, 400, 480, 752),
    "exactly_top_k": (13, "exact", 0.004, 4, 48, 480, 640),
    "ties_below_top_k": (14, "ties_sparse", 0.004, 4, 400, 480, 752),
    "ties_above_top_k": (15, "ties_dense", 0.004, 4, 64, 480, 752),
    "borders_inclusive": (16, "border", 0.004, 4, 400, 512, 512),
    "border_zero": (17, "border0", 0.004, 0, 128, 512, 512),
    "nothing_above_threshold": (18, "empty", 0.5, 4, 400, 480, 752),
    "at_the_threshold": (19, "threshold", 0.015625, 4, 400, 480, 752),
    "zero_descriptors": (20, "zero_desc", 0.004, 4, 64, 720, 1280),
}


def detect_case(name: str) -> dict:
    seed, kind, thr, border, top_k, H, W = DETECT[name]
    rng = np.random.default_rng(seed)
    heat = np.zeros((R, R), F)
    desc = (rng.random((256, R // 8, R // 8), dtype=F) - F(0.5)).astype(F)

    def scatter(n, lo=0.01, hi=1.0, quant=None, region=None):
        y0, y1, x0, x1 = region or (0, R, 0, R)
        ys = rng.integers(y0, y1, n); xs = rng.integers(x0, x1, n)
        v = (rng.random(n, dtype=F) * F(hi - lo) + F(lo)).astype(F)
        if quant:
            v = (np.floor(v * F(quant)) / F(quant)).astype(F)
        heat[ys, xs] = v

    if kind == "dense":
        scatter(3000)
    elif kind == "sparse":
        scatter(60)
    elif kind == "exact":
        ys, xs = np.divmod(rng.permutation(np.arange(40 * 40))[:top_k], 40)
        heat[ys * 8 + 100, xs * 8 + 100] = (rng.random(top_k, dtype=F) * F(0.9) + F(0.05)).astype(F)
    elif kind == "ties_sparse":
        scatter(120, quant=16)
    elif kind == "ties_dense":
        scatter(2000, lo=0.1, quant=64)      # 58 distinct values over 2000 points: ties inside and across the top-k cut
    elif kind in ("border", "border0"):
        b = border
        for v in (0, 1, b - 1, b, b + 1, R - b - 1, R - b, R - b + 1, R - 1):
            if 0 <= v < R:
                heat[v, 37:R:61] = (rng.random(len(range(37, R, 61)), dtype=F) * F(0.5) + F(0.2)).astype(F)
                heat[41:R:67, v] = (rng.random(len(range(41, R, 67)), dtype=F) * F(0.5) + F(0.2)).astype(F)
        scatter(40)
    elif kind == "empty":
        scatter(500, lo=0.0, hi=0.4)
    elif kind == "threshold":
        scatter(200, lo=0.0, hi=0.03, quant=256)       # multiples of 1/256 around thr = 4/256: == thr is kept, one step below is not
    elif kind == "zero_desc":
        scatter(200)
        desc[:, 10:30, 10:40] = 0                      # keypoints whose four taps are all zero: normalize() must leave them zero
    image = rng.integers(0, 256, (H, W), dtype=np.uint8)
    return dict(name=name, heat=heat, desc=desc, thr=thr, border=border, top_k=top_k, image=image)


# ================================================================================================================ plnet
PLNET = {        # name: (seed, n_keep, line_threshold, line_length_threshold, border, H, W)
    "typical": (31, 1200, 0.75, 50.0, 4, 480, 752),
    "many_duplicates": (32, 6000, 0.75, 50.0, 4, 480, 752),
    "no_proposal_kept": (33, 0, 0.75, 50.0, 4, 480, 752),
    "one_proposal": (34, 1, 0.5, 1.0, 4, 512, 512),
    "short_lines_and_thresholds": (35, 800, 0.75, 20.0, 8, 480, 640),
}


def plnet_case(name: str) -> dict:
    seed, n_keep, lthr, llen, border, H, W = PLNET[name]
    rng = np.random.default_rng(seed)
    fh = R // 4
    base = detect_case("sparse_raster_order")
    heat, desc = base["heat"].copy(), base["desc"]
    # junctions in feature coordinates: mostly quarter-pixel grid values (what x4 turns into whole 512-pixel coordinates, the case the
    # reference's `(int)(x + 0.1)` is written for), some just below / above a whole pixel, some near and inside the border
    juncs = (rng.integers(0, 4 * (fh - 1), (300, 2)).astype(F) / F(4)).astype(F)
    juncs[:40] += (rng.random((40, 2), dtype=F) * F(0.2) - F(0.1)).astype(F)
    juncs[40:60] = (rng.integers(0, 3 * border, (20, 2)).astype(F) / F(4)).astype(F)
    juncs[60:80] = (F(fh) - F(0.25) - rng.integers(0, 3 * border, (20, 2)).astype(F) / F(4)).astype(F)
    juncs = np.clip(juncs, F(0), F(fh) - F(0.3)).astype(F)
    lines_pred = (rng.random((3 * fh * fh, 4), dtype=F) * F(fh)).astype(F)
    iskeep = np.zeros(3 * fh * fh, F)
    idx_min = np.zeros(3 * fh * fh, F)
    idx_max = np.zeros(3 * fh * fh, F)
    if n_keep:
        where = rng.choice(3 * fh * fh, n_keep, replace=False)
        iskeep[where] = (rng.random(n_keep, dtype=F) + F(0.001)).astype(F)         # > 0: kept (any positive value)
        iskeep[rng.choice(3 * fh * fh, 500, replace=False)] -= F(0.5)               # and some negative / smaller ones elsewhere
        npairs = max(1, n_keep // (6 if "dup" in name else 1))
        pa = rng.integers(0, 120, npairs); pb = rng.integers(0, 120, npairs)          # 120 of the 300 junctions are line ends (keeps the fixtures small)
        pick = rng.integers(0, npairs, 3 * fh * fh)
        a, b = pa[pick], pb[pick]
        idx_min[:] = np.minimum(a, b); idx_max[:] = np.maximum(a, b)                # incl. a == b (a degenerate "line")
    for j in range(0, 300, 7):                                                       # junction pixels that carry a heat value (junction scores)
        heat[min(int(juncs[j, 1] * 4 + 0.1), R - 1), min(int(juncs[j, 0] * 4 + 0.1), R - 1)] = F(0.25) + F(j) / F(1024)
    image = rng.integers(0, 256, (H, W), dtype=np.uint8)
    return dict(name=name, heat=heat, desc=desc, juncs_pred=juncs, lines_pred=lines_pred, iskeep=iskeep.reshape(1, 3, fh, fh),
                idx_min=idx_min.reshape(1, 3, fh, fh), idx_max=idx_max.reshape(1, 3, fh, fh), thr=0.004, border=border, top_k=400,
                line_threshold=lthr, line_length_threshold=llen, image=image, seed=seed)


def stage1_stub(juncs: np.ndarray, pairs: np.ndarray, seed: int):
    """What stands in for the stage-1 engine in the plnet family: `lines_adjusted` exactly as plnet_s1.onnx computes it (a gather of the two
    junctions) and a seeded score per unique line that lands ON the reference's thresholds (0.5, line_threshold) for some."""
    pairs = np.asarray(pairs).astype(np.int64).reshape(-1, 2)
    la = np.concatenate([juncs[pairs[:, 0]], juncs[pairs[:, 1]]], axis=1).astype(F)
    h = (pairs[:, 0] * 7919 + pairs[:, 1] * 104729 + seed * 13) % 1000
    sc = (h.astype(F) / F(999)).astype(F)
    sc[h % 17 == 0] = F(0.5)
    sc[h % 19 == 0] = F(0.75)
    sc[h % 23 == 0] = np.nextafter(F(0.75), F(0))
    sc[h % 29 == 0] = np.nextafter(F(0.5), F(0))
    return la, sc


# ================================================================================================================ matchers
MATCH = {        # name: (seed, n0, n1, kind, width, height)
    "planted_400": (51, 400, 400, "planted", 752, 480),
    "ragged_317_1024": (52, 317, 1024, "planted", 1280, 720),
    "ties": (53, 64, 80, "ties", 752, 480),
    "floors_inf_nan": (54, 48, 40, "floors", 752, 480),
    "threshold_ulps": (55, 32, 32, "ulps", 640, 480),
    "one_by_one": (56, 1, 1, "planted", 752, 480),
    "empty_side": (57, 0, 25, "planted", 752, 480),
}


def match_case(name: str, family: str) -> dict:
    """family 'lg': scores [n0][n1] (log assignment, threshold exp > 0.1); 'sg': Z [n0+1][n1+1] incl. dustbins (exp > 0.2)."""
    seed, n0, n1, kind, width, height = MATCH[name]
    rng = np.random.default_rng(seed + (1000 if family == "sg" else 0))

    def feats(n):
        f = np.zeros((n, 259), F)
        f[:, 0] = rng.random(n, dtype=F)
        f[:, 1] = (rng.random(n, dtype=F) * F(width)).astype(F)
        f[:, 2] = (rng.random(n, dtype=F) * F(height)).astype(F)
        d = (rng.random((n, 256), dtype=F) - F(0.5)).astype(F)
        f[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True).astype(F)
        return f

    f0, f1 = feats(n0), feats(n1)
    thr = 0.1 if family == "lg" else 0.2
    s = (-(rng.random((n0, n1), dtype=F) * F(12) + F(3))).astype(F)                     # background: log-scores in [-15, -3]
    k = min(n0, n1)
    if k and kind in ("planted", "ties", "floors", "ulps"):
        rows = rng.permutation(n0)[:max(1, (2 * k) // 3)]
        cols = rng.permutation(n1)[:len(rows)]
        s[rows, cols] = (-(rng.random(len(rows), dtype=F) * F(3))).astype(F)            # planted mutual maxima, exp in (0.05, 1]
    if kind == "ties":
        s = (np.round(s * F(2)) / F(2)).astype(F)                                        # heavy ties: first maximum must win
        s[5, :] = s[5, 0]; s[:, 7] = s[0, 7]
    if kind == "floors":
        s[3, :] = -np.inf; s[:, 4] = -np.inf                                             # nothing above the floor: the value-initialised pair
        s[6, :] = np.nan
        s[8, :] = -np.finfo(F).max
        s[10, 2] = np.inf
        s[:, 0] = np.where(np.arange(n0) % 5 == 0, F(-0.1), s[:, 0])
    if kind == "ulps":
        t = F(np.log(F(thr)))
        vals = [t, np.nextafter(t, F(0)), np.nextafter(t, F(-10)), np.nextafter(np.nextafter(t, F(0)), F(0)), F(0), F(-1e-7)]
        for i in range(min(n0, n1)):
            s[i, :] = F(-20); s[:, i] = np.minimum(s[:, i], F(-20))
        for i in range(min(n0, n1)):
            s[i, i] = vals[i % len(vals)]
    if family == "sg":
        z = (-(rng.random((n0 + 1, n1 + 1), dtype=F) * F(2) + F(1))).astype(F)         # dustbin row / column: never read by decode
        z[:n0, :n1] = s
        s = z
    return dict(name=name, f0=f0, f1=f1, scores=s, width=width, height=height, matcher=0 if family == "lg" else 1)


# ================================================================================================================ lines
LINES = {"typical": (71, 60, 400), "dense": (72, 240, 400), "one_line": (73, 1, 50), "degenerate_line": (74, 8, 120)}


def lines_case(name: str) -> dict:
    seed, nl, n = LINES[name]
    rng = np.random.default_rng(seed)

    def frame():
        p1 = rng.random((nl, 2)) * [752, 480]
        ang = rng.random(nl) * np.pi
        ln = rng.random(nl) * 200 + 20
        p2 = p1 + np.stack([np.cos(ang), np.sin(ang)], 1) * ln[:, None]
        lines = np.concatenate([p1, p2], 1).astype(np.float64)
        if "degenerate" in name:
            lines[0, 2:] = lines[0, :2]                     # zero length: D = 0, the distance is 0/0 or x/0
            lines[1, 3] = lines[1, 1]                       # horizontal
            lines[2, 2] = lines[2, 0]                       # vertical
        f = np.zeros((n, 259), F)
        t = rng.random(n)
        which = rng.integers(0, nl, n)
        on = lines[which, :2] * (1 - t[:, None]) + lines[which, 2:] * t[:, None]
        off = (rng.random((n, 2)) - 0.5) * np.array([8.0, 8.0]) * (rng.random((n, 1)) < 0.7)
        pts = on + off
        pts[: n // 10] = lines[which[: n // 10], :2] + (rng.random((n // 10, 2)) - 0.5) * 6      # around end points (side <= 9 rule)
        pts[n // 10: n // 5] = rng.random((n // 5 - n // 10, 2)) * [752, 480]
        f[:, 1:3] = pts.astype(F)
        return lines, f

    l0, f0 = frame()
    l1, f1 = frame()
    m = rng.permutation(n)[: (2 * n) // 3]
    query = np.sort(m).astype(np.int32)
    train = rng.permutation(n)[: len(query)].astype(np.int32)
    return dict(name=name, lines0=l0, feat0=f0, lines1=l1, feat1=f1, query=query, train=train)


# ================================================================================================================ BoW
BOW = {"k10_L4_400": (10, 4, 400, 91), "k8_L3_1024": (8, 3, 1024, 92), "k3_L2_1": (3, 2, 1, 93), "exact_ties": (2, 2, 4, 94)}


def bow_case(name: str) -> dict:
    """A vocabulary tree (airslam_amd.weights.synthetic_vocabulary: voc/point_voc_L4.bin is absent upstream) + feature rows near its leaves; `exact_ties`
    is a hand-built tree in which a descriptor is EXACTLY as far from two children (the first must win) and a leaf is a stopped word (weight 0)."""
    from airslam_amd import weights
    k, L, n, seed = BOW[name]
    if name == "exact_ties":
        e = np.eye(256, dtype=F)
        voc = dict(desc=np.stack([0 * e[0], e[0], e[1], e[0] + e[2], e[0] - e[2]]).astype(F), first_child=np.array([1, 3, 0, 0, 0], np.int32),
                   n_children=np.array([2, 2, 0, 0, 0], np.int32), word_id=np.array([0, 0, 7, 3, 4], np.int32), weight=np.array([0, 0, 2.0, 0.0, 1.5]), k=2, L=2)
        feat = np.zeros((4, 259), F)
        feat[:, 3:] = np.stack([e[1], e[0] + F(0.5) * e[2], e[0] - F(0.5) * e[2], F(0.5) * (e[0] + e[1])])
        return dict(name=name, voc=voc, feat=feat)
    voc = weights.synthetic_vocabulary(1234, k=k, L=L)
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["n_children"] == 0)[0]
    pick = leaves[rng.integers(0, len(leaves), n)]
    d = (voc["desc"][pick] + (rng.random((n, 256), dtype=F) - F(0.5)) * F(0.5)).astype(F)
    feat = np.zeros((n, 259), F)
    feat[:, 0] = rng.random(n, dtype=F)
    feat[:, 3:] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    return dict(name=name, voc=voc, feat=feat)


# ================================================================================================================ runners
def _scales(image):
    h, w = image.shape
    return F(F(w) / F(R)), F(F(h) / F(R))


def run_ref(family: str, case: dict, tmp: str) -> dict:
    """The compiled reference (oracle/_ref) on a case.  Returns the outputs + what the engines were fed (`fed`)."""
    from oracle import ref_lib
    md = os.path.join(tmp, "ref_models")
    if family == "detect":
        calls = ref_lib.set_engines({"superpoint": lambda ins: dict(scores=case["heat"], descriptors=case["desc"]),
                                     "plnet_s0": None, "plnet_s1": None})
        det = ref_lib.FeatureDetector(md, use_superpoint=1, max_keypoints=case["top_k"], keypoint_threshold=case["thr"], remove_borders=case["border"])
        r = det.detect(0, case["image"])
        det.close()
        return dict(ok=r["ok"], feat=r["feat_l"], fed_input=calls[0][1]["input"])
    if family == "plnet":
        s1_in = {}

        def s0(ins):
            fh = R // 4
            return dict(scores=case["heat"], descriptors=case["desc"], juncs_pred=case["juncs_pred"], lines_pred=case["lines_pred"],
                        iskeep=case["iskeep"], idx_junc_to_end_min=case["idx_min"], idx_junc_to_end_max=case["idx_max"],
                        loi_features=np.zeros((1, 128, fh, fh), F), loi_features_thin=np.zeros((1, 4, fh, fh), F),
                        loi_features_aux=np.zeros((1, 4, fh, fh), F))

        def s1(ins):
            s1_in.update({k: v.copy() for k, v in ins.items() if k in ("idx_lines_for_junctions", "inverse", "iskeep_index", "juncs_pred")})
            la, sc = stage1_stub(ins["juncs_pred"], ins["idx_lines_for_junctions"], case["seed"])
            return dict(lines_adjusted=la, scores_line=sc)

        ref_lib.set_engines({"plnet_s0": s0, "plnet_s1": s1})
        det = ref_lib.FeatureDetector(md, use_superpoint=0, max_keypoints=case["top_k"], keypoint_threshold=case["thr"], remove_borders=case["border"],
                                      line_threshold=case["line_threshold"], line_length_threshold=case["line_length_threshold"])
        pre = np.array([[1.0, 2.0, 3.0, 4.0]])
        r = det.detect(2, case["image"], lines_in=pre)            # the reference APPENDS to the caller's vector (src/plnet.cpp:544)
        det.close()
        return dict(ok=r["ok"], feat=r["feat_l"], lines=r["lines_l"], junc=r["junc"],
                    pairs=s1_in.get("idx_lines_for_junctions", np.zeros((0, 2), F)).astype(np.int32),
                    inverse=s1_in.get("inverse", np.zeros((0, 1), F))[:, 0].astype(np.int32),
                    iskeep_index=s1_in.get("iskeep_index", np.zeros((0, 1), F))[:, 0].astype(np.int32))
    if family in ("lg", "sg"):
        kind = "lightglue" if family == "lg" else "superglue"
        calls = ref_lib.set_engines({kind: lambda ins: dict(scores=case["scores"])})
        pm = ref_lib.PointMatcher(md, case["matcher"], case["width"], case["height"])
        cnt, q, t, d = pm.matching_points(case["f0"], case["f1"])
        norm0 = pm.normalize_keypoints(case["f0"], case["width"], case["height"], 0.5 if family == "lg" else 0.7)
        pm.close()
        out = dict(count=cnt, query=q, train=t, distance=d, norm0_head=norm0[:, :4], norm0_tail_unchanged=np.array_equal(norm0[:, 3:], case["f0"][:, 3:]))
        if calls:                                                  # what process_input handed to the engine (descriptors: 16 columns' worth)
            fed = calls[0][1]
            out["fed_kp0"] = fed["keypoints_0"][0]
            out["fed_desc0_sample"] = fed["descriptors_0"][0][::16] if family == "lg" else fed["descriptors_0"][0][:, ::16]
            out["fed_desc0_is_the_input"] = np.array_equal(fed["descriptors_0"][0] if family == "lg" else fed["descriptors_0"][0].T, case["f0"][:, 3:])
            if family == "sg":
                out["fed_scores0"] = fed["scores_0"][0]
        return out
    if family == "lines":
        o0, i0, d0 = ref_lib.assign_points_to_lines(case["lines0"], case["feat0"])
        o1, i1, d1 = ref_lib.assign_points_to_lines(case["lines1"], case["feat1"])
        lm = ref_lib.match_lines(o0, i0, o1, i1, case["query"], case["train"], len(case["feat0"]), len(case["feat1"]))
        return dict(off0=o0, idx0=i0, dist0=d0, off1=o1, idx1=i1, dist1=d1, line_matches=lm)
    if family == "bow":
        w, wt, ids, vals = ref_lib.bow_frame_to_bow(case["voc"], case["feat"])
        return dict(word_of_features=w, weight_of_features=wt, bow_ids=ids, bow_values=vals)
    raise KeyError(family)


def run_post(family: str, case: dict) -> dict:
    """The numpy restatement (oracle/ref_post.py) on a case, in the shape of run_ref's outputs."""
    from oracle import ref_post
    if family == "detect":
        ws, hs = _scales(case["image"])
        x, _, _ = ref_post.process_image(case["image"])
        return dict(ok=True, feat=ref_post.keypoints_decoder(case["heat"], case["desc"], case["thr"], case["border"], case["top_k"], ws, hs),
                    fed_input=x[None, None])
    if family == "plnet":
        ws, hs = _scales(case["image"])
        keep, inv, pairs = ref_post.wireframe_matcher(case["iskeep"], case["idx_min"], case["idx_max"])
        la, sc = stage1_stub(case["juncs_pred"], pairs, case["seed"])
        lines512, jmap = ref_post.line_filter(la, sc, case["border"], case["line_threshold"], case["line_length_threshold"])
        lines = np.concatenate([np.array([[1.0, 2.0, 3.0, 4.0]]), lines512])            # the pre-existing line is rescaled too (:577-582)
        return dict(ok=True, feat=ref_post.keypoints_decoder(case["heat"], case["desc"], case["thr"], case["border"], case["top_k"], ws, hs),
                    lines=ref_post.rescale_lines(lines, ws, hs), junc=ref_post.junction_detector(case["heat"], case["desc"], jmap, case["border"], ws, hs),
                    pairs=pairs.astype(np.int32), inverse=inv.astype(np.int32), iskeep_index=keep.astype(np.int32))
    if family in ("lg", "sg"):
        n0, n1 = len(case["f0"]), len(case["f1"])
        scale = 0.5 if family == "lg" else 0.7
        norm0 = ref_post.normalize_keypoints(case["f0"], case["width"], case["height"], scale)
        out = dict(norm0_head=norm0[:, :4], norm0_tail_unchanged=np.array_equal(norm0[:, 3:], case["f0"][:, 3:]))
        if n0 < 1 or n1 < 1:                                        # src/point_matcher.cc:53-55
            out.update(count=0, query=np.zeros(0, np.int32), train=np.zeros(0, np.int32), distance=np.zeros(0, F))
            return out
        if family == "lg":
            idx, sc = ref_post.filter_matches(case["scores"], 0.1)
            out.update(count=len(idx), query=idx[:, 0].astype(np.int32), train=idx[:, 1].astype(np.int32),
                       distance=(1.0 - sc.astype(np.float64)).astype(F))                 # `1.0 - matches_score(i)`: double, narrowed by DMatch
            out.update(fed_kp0=norm0[:, 1:3], fed_desc0_sample=norm0[::16, 3:], fed_desc0_is_the_input=True)
        else:
            i0, i1, m0, m1 = ref_post.superglue_decode(case["scores"], 0.2)
            ms = ref_post.superglue_matches(i0, i1, m0, m1)
            out.update(count=len(ms), query=np.array([m[0] for m in ms], np.int32), train=np.array([m[1] for m in ms], np.int32),
                       distance=np.array([m[2] for m in ms], np.float64).astype(F))
            out.update(fed_kp0=norm0[:, 1:3], fed_desc0_sample=np.ascontiguousarray(norm0[:, 3:].T)[:, ::16], fed_desc0_is_the_input=True,
                       fed_scores0=norm0[:, 0])
        return out
    if family == "lines":
        def csr(rel):
            off = np.zeros(len(rel) + 1, np.int32)
            idx, dist = [], []
            for i, r in enumerate(rel):
                for k in sorted(r):
                    idx.append(k); dist.append(r[k])
                off[i + 1] = len(idx)
            return off, np.array(idx, np.int32), np.array(dist, np.float64)
        r0 = ref_post.assign_points_to_lines(case["lines0"], case["feat0"])
        r1 = ref_post.assign_points_to_lines(case["lines1"], case["feat1"])
        o0, i0, d0 = csr(r0); o1, i1, d1 = csr(r1)
        lm = ref_post.match_lines(r0, r1, list(zip(case["query"].tolist(), case["train"].tolist())), len(case["feat0"]), len(case["feat1"]))
        return dict(off0=o0, idx0=i0, dist0=d0, off1=o1, idx1=i1, dist1=d1, line_matches=np.array(lm, np.int32))
    if family == "bow":
        w, wt = ref_post.bow_transform(case["voc"], case["feat"][:, 3:])
        bow, _ = ref_post.frame_to_bow(w, wt)
        return dict(word_of_features=w, weight_of_features=wt, bow_ids=np.array(list(bow), np.uint32), bow_values=np.array(list(bow.values()), np.float64))
    raise KeyError(family)


FAMILIES = {"detect": (DETECT, detect_case), "plnet": (PLNET, plnet_case), "lg": (MATCH, lambda n: match_case(n, "lg")),
            "sg": (MATCH, lambda n: match_case(n, "sg")), "lines": (LINES, lines_case), "bow": (BOW, bow_case)}
# descriptors go through Eigen's colwise().normalize(): the summation order of squaredNorm() is Eigen's own (SSE packets in a real-Eigen build,
# index order in the stand-in) and numpy's pairwise — the ONE place where the restatement may differ from the compiled reference, by summation order
DESC_TOL = 2e-6


def assert_same(family: str, name: str, got: dict, want: dict, case: dict = None, desc_tol: float = 0.0):
    """got == want output by output: integers / indices / coordinates / scores exact; descriptor columns within desc_tol (0 = exact).
    The one licence: `detect_point` orders more-than-top_k candidates with std::sort (src/plnet.cpp:336, unstable) — candidates of EQUAL score
    may come in any order and, at the cut, any of them may be the ones kept."""
    for k in want:
        a, b = np.asarray(got[k]), np.asarray(want[k])
        assert a.shape == b.shape, f"{family}/{name}/{k}: shape {a.shape} != {b.shape}"
        if k in ("feat", "junc") and a.size:
            if not np.array_equal(a[:, :3], b[:, :3]):
                assert k == "feat" and case is not None, f"{family}/{name}/{k}: score / x / y differ"
                _assert_same_up_to_ties(a, b, case, f"{family}/{name}/{k}")
                continue
            d = np.abs(a[:, 3:] - b[:, 3:]).max()
            assert d <= desc_tol, f"{family}/{name}/{k}: descriptors differ by {d} (> {desc_tol})"
        elif a.dtype.kind == "f":
            assert np.array_equal(a, b, equal_nan=True), f"{family}/{name}/{k}: differs (max {np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))})"
        else:
            assert np.array_equal(a, b), f"{family}/{name}/{k}: differs"


def _assert_same_up_to_ties(a, b, case, what):
    ws, hs = _scales(case["image"])
    assert np.array_equal(a[:, 0], b[:, 0]), f"{what}: score column differs"
    cut = a[-1, 0]
    ra = {tuple(r) for r in a[a[:, 0] > cut]}; rb = {tuple(r) for r in b[b[:, 0] > cut]}
    assert ra == rb, f"{what}: rows above the cut score differ as sets"
    for r in a[a[:, 0] == cut]:                                      # at the cut: any candidate of that score
        x, y = int(round(float(r[1] / ws))), int(round(float(r[2] / hs)))
        assert case["heat"][y, x] == cut, f"{what}: ({x}, {y}) is not a candidate of the cut score"
    xy = {(float(r[1]), float(r[2])) for r in a}
    assert len(xy) == len(a), f"{what}: duplicate keypoints"
