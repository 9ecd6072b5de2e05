"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

numpy restatement of every CPU routine on the reference's detect/match path.
Each function cites the reference file:line it follows (paths under
/root/reference).  Arithmetic is carried out in float32 op by op wherever the
C++ does float arithmetic, so results match the C++ up to FMA contraction.

PINNED TO THE REFERENCE'S OWN CODE (round 4): the reference ships no tests, golden vectors or fixtures for this path,
but its host code compiles: oracle/Makefile builds /root/reference's src/{feature_detector.cc, point_matcher.cc, plnet.cpp, super_point.cpp,
light_glue.cpp, super_glue.cpp} and line_processor.cc:1-180 UNCHANGED into oracle/_ref/libairslam_ref.so (stand-ins only for Eigen, OpenCV,
yaml-cpp and TensorRT, whose engines become a callback), and tests/test_ref_pin_cpu.py holds every function below that has a counterpart there
to it bit for bit — live, and through the committed outputs tests/golden/ref_pin.npz.  Still pinned by reading only: resize_linear_u8 /
remap_linear_u8 (OpenCV absent), simple_nms (inside the absent ONNX graphs).  bow_transform / frame_to_bow are pinned to the vendored DBoW2 +
src/bow/FSuperpoint.cc compiled unchanged (oracle/ref_bow.cpp).
"""
from __future__ import annotations

import numpy as np

F = np.float32


# ------------------------------------------------------------------ pre-process
def _resize_coeffs(dsize: int, ssize: int):
    """OpenCV 4.x resize() INTER_LINEAR coefficient table for 8-bit images
    (imgproc/src/resize.cpp: `fx = (float)((dx+0.5)*scale_x - 0.5)`, edge clamps,
    `saturate_cast<short>(cbuf[k]*INTER_RESIZE_COEF_SCALE)` with scale 2048)."""
    scale = float(ssize) / float(dsize)
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(F)).astype(F)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= ssize - 1
    f[hi] = 0
    s[hi] = ssize - 1
    a0 = np.rint((F(1.0) - f) * F(2048.0)).astype(np.int64)   # cvRound: half to even
    a1 = np.rint(f * F(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, a0, a1


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize(src, dst, Size(dw, dh)) for CV_8UC1, default INTER_LINEAR, as called at
    src/plnet.cpp:258 and src/super_point.cpp:116 (fixed-point path: HResizeLinear with
    11-bit coefficients, VResizeLinear `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`)."""
    sh, sw = src.shape
    if sw == dw and sh == dh:
        return src.copy()
    xs, xs1, xa0, xa1 = _resize_coeffs(dw, sw)
    ys, ys1, yb0, yb1 = _resize_coeffs(dh, sh)
    s = src.astype(np.int64)
    hp = s[:, xs] * xa0[None, :] + s[:, xs1] * xa1[None, :]            # [sh, dw]
    r0 = hp[ys, :]
    r1 = hp[ys1, :]
    out = (((yb0[:, None] * (r0 >> 4)) >> 16) + ((yb1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def process_image(image: np.ndarray, rw: int = 512, rh: int = 512):
    """PLNet::process_image src/plnet.cpp:246-270 (≡ SuperPoint src/super_point.cpp:111-116,146-165):
    resize to 512x512, `float(px) / 255.0` (double division, stored as float)."""
    h, w = image.shape
    w_scale = F(F(w) / F(rw))
    h_scale = F(F(h) / F(rh))
    r = resize_linear_u8(image, rw, rh)
    x = (r.astype(np.float64) / 255.0).astype(F)
    return x, w_scale, h_scale


# ------------------------------------------------------------------ BoW quantisation
def bow_transform(voc: dict, desc: np.ndarray, return_margin: bool = False):
    """TemplatedVocabulary::transform(feature, word_id, weight) (3rdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1313-1352) per row of
    desc [N,256]: descend from the root, at every node to the child with the smallest squared L2 distance (FSuperpoint::distance,
    src/bow/FSuperpoint.cc:45-49; FIRST minimum, strict '<'), to a leaf.  Returns (word_of_features uint32 with UINT_MAX where the
    leaf weight is <= 0 — src/bow/database.cc:77-83 —, weights float64[, the smallest relative gap between the two nearest children
    seen on the way: rows with a tiny gap are decided by float summation order])."""
    d = desc.astype(np.float32)
    n = d.shape[0]
    words = np.zeros(n, np.uint32); weights = np.zeros(n, np.float64); margin = np.full(n, np.inf)
    for i in range(n):
        node = 0
        while voc["n_children"][node] > 0:
            c0, nc = int(voc["first_child"][node]), int(voc["n_children"][node])
            # FSuperpoint::distance: `diff = a - b; return diff.transpose() * diff;` in float (Matrix<float, 256, 1>), the inner product reduced in
            # Eigen's packet order (_eigen_sse2_sum), widened to double only on return — pinned against the compiled DBoW2 (tests/test_ref_pin_cpu.py)
            diff = (d[i][None] - voc["desc"][c0:c0 + nc].astype(F)).astype(F)
            dist = _eigen_sse2_sum(np.ascontiguousarray((diff * diff).astype(F).T)).astype(np.float64)
            j = int(np.argmin(dist))                                   # first minimum (`if (d < best_d)`, TemplatedVocabulary.h:1336)
            if nc > 1:
                s2 = np.partition(dist, 1)[:2]
                margin[i] = min(margin[i], (s2[1] - s2[0]) / max(s2[1], 1e-30))
            node = c0 + j
        w = float(voc["weight"][node])
        weights[i] = w
        words[i] = np.uint32(voc["word_id"][node]) if w > 0 else np.uint32(0xFFFFFFFF)
    return (words, weights, margin) if return_margin else (words, weights)


def frame_to_bow(words: np.ndarray, weights: np.ndarray):
    """The rest of Database::FrameToBow (src/bow/database.cc:66-97) on the per-feature (word, weight) pairs: BowVector::addWeight (sum per
    word, std::map order), word_features[id] = feature indices, then L1 normalisation (DBoW2's default scoring, L1_NORM)."""
    bow, wf = {}, {}
    for i, (wid, w) in enumerate(zip(words.tolist(), weights.tolist())):
        if w > 0:
            bow[wid] = bow.get(wid, 0.0) + w
            wf.setdefault(wid, []).append(i)
    tot = 0.0
    for k in sorted(bow):                      # BowVector::normalize walks the std::map in ascending word id (3rdparty/DBoW2/src/BowVector.cpp)
        tot += abs(bow[k])
    if tot > 0:
        bow = {k: v / tot for k, v in bow.items()}
    return dict(sorted(bow.items())), dict(sorted(wf.items()))


# ------------------------------------------------------------------ rectification
def remap_linear_u8(src: np.ndarray, mapx: np.ndarray, mapy: np.ndarray) -> np.ndarray:
    """cv::remap(src, dst, map1, map2, cv::INTER_LINEAR) for CV_8UC1 with CV_32FC1 maps and the default BORDER_CONSTANT(0), as
    Camera::UndistortImage calls it (src/camera.cc:161-182).  Restated from OpenCV 4.x imgproc/src/imgwarp.cpp (RemapInvoker,
    initInterTab2D, remapBilinear<FixedPtCast<int, uchar, INTER_REMAP_COEF_BITS = 15>>) — PARITY UNPINNED: OpenCV is absent here.
      sx = cvRound(mapx * 32) (float multiply, round half to even); integer part sx >> 5 saturated to short, fraction sx & 31
      weights = bilinear products * 2^15 — exact integers, except entry (0, 0): saturate_cast<short>(32768) = 32767 and the table's
      sum correction adds the missing 1 to the DIAGONAL tap (initInterTab2D's index arithmetic for ksize = 2)
      dst = (sum_i w_i * tap_i + 2^14) >> 15 ; a tap outside the image contributes 0 ; wholly outside -> 0"""
    h, w = src.shape
    assert mapx.shape == mapy.shape
    sx = np.rint((mapx.astype(F) * F(32)).astype(F)).astype(np.int64)
    sy = np.rint((mapy.astype(F) * F(32)).astype(F)).astype(np.int64)
    ix = np.clip(sx >> 5, -32768, 32767)
    iy = np.clip(sy >> 5, -32768, 32767)
    fx, fy = sx & 31, sy & 31
    w00 = (32 - fx) * (32 - fy) * 32; w01 = fx * (32 - fy) * 32; w10 = (32 - fx) * fy * 32; w11 = fx * fy * 32
    z = (fx | fy) == 0
    w00 = np.where(z, 32767, w00); w11 = np.where(z, 1, w11)
    s = src.astype(np.int64)

    def tap(xx, yy):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(ok, s[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], 0)
    v = (tap(ix, iy) * w00 + tap(ix + 1, iy) * w01 + tap(ix, iy + 1) * w10 + tap(ix + 1, iy + 1) * w11 + (1 << 14)) >> 15
    outside = (ix >= w) | (ix + 1 < 0) | (iy >= h) | (iy + 1 < 0)
    return np.where(outside, 0, np.clip(v, 0, 255)).astype(np.uint8)


# ------------------------------------------------------------------ model-side NMS
def simple_nms(scores: np.ndarray, radius: int) -> np.ndarray:
    """SuperPoint `simple_nms` (public SuperGluePretrainedNetwork models/superpoint.py); believed
    to be baked into superpoint_v1_sim_int32.onnx / plnet_s0.onnx (UNVERIFIED-UPSTREAM; the C++
    does no NMS: src/plnet.cpp:309-355).  radius 0 disables it."""
    if radius <= 0:
        return scores.copy()

    def max_pool(x):
        h, w = x.shape
        k = 2 * radius + 1
        p = np.full((h + 2 * radius, w + 2 * radius), -np.inf, dtype=x.dtype)
        p[radius:radius + h, radius:radius + w] = x
        # separable sliding max
        t = p[:, 0:w].copy()
        for i in range(1, k):
            np.maximum(t, p[:, i:i + w], out=t)
        o = t[0:h].copy()
        for i in range(1, k):
            np.maximum(o, t[i:i + h], out=o)
        return o

    zeros = np.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.astype(scores.dtype)) > 0
        supp_scores = np.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return np.where(max_mask, scores, zeros)


# ------------------------------------------------------------------ keypoint decode
def detect_point(heat: np.ndarray, threshold: float, border: int, top_k: int):
    """PLNet::detect_point src/plnet.cpp:309-355 ≡ SuperPoint::detect_point src/super_point.cpp:174-217.
    Upper border bound is INCLUSIVE (x > w-border rejects).  If more than top_k candidates: sorted by
    score descending (std::sort, unstable → we fix ties by ascending raster index);
    otherwise raster order, unsorted."""
    h, w = heat.shape
    flat = heat.reshape(-1)
    idx = np.nonzero(~(flat < F(threshold)))[0]
    y = idx // w
    x = idx - y * w
    keep = ~((x < border) | (x > w - border) | (y < border) | (y > h - border))
    idx, x, y = idx[keep], x[keep], y[keep]
    s = flat[idx]
    if idx.size > top_k:
        order = np.argsort(-s.astype(np.float64), kind="stable")[:top_k]
        idx, x, y, s = idx[order], x[order], y[order], s[order]
    return s.astype(F), x.astype(F), y.astype(F)


def _clip(v, mx):
    return np.maximum(0, np.minimum(v, mx - 1))


def extract_descriptors(desc_chw: np.ndarray, xs: np.ndarray, ys: np.ndarray, s: int = 8) -> np.ndarray:
    """PLNet::extract_descriptors src/plnet.cpp:369-417 ≡ src/super_point.cpp:224-272.
    desc_chw: [256, h, w] float32.  Returns [N, 256] (row n = Eigen column n rows 3..258)."""
    c, h, w = desc_chw.shape
    sx = F(2.0 / (w * s - s // 2 - 0.5))
    bx = F((1 - s) / (w * s - s // 2 - 0.5) - 1)
    sy = F(2.0 / (h * s - s // 2 - 0.5))
    by = F((1 - s) / (h * s - s // 2 - 0.5) - 1)
    kx = (xs.astype(F) * sx + bx).astype(F)
    ky = (ys.astype(F) * sy + by).astype(F)
    kx = ((kx + F(1)) * F(0.5)).astype(F)
    ky = ((ky + F(1)) * F(0.5)).astype(F)
    ix = (kx * F(w - 1)).astype(F)
    iy = (ky * F(h - 1)).astype(F)
    ix_nw = _clip(np.floor(ix).astype(np.int64), w)
    iy_nw = _clip(np.floor(iy).astype(np.int64), h)
    ix_ne = _clip(ix_nw + 1, w)
    iy_ne = _clip(iy_nw, h)
    ix_sw = _clip(ix_nw, w)
    iy_sw = _clip(iy_nw + 1, h)
    ix_se = _clip(ix_nw + 1, w)
    iy_se = _clip(iy_nw + 1, h)
    nw = ((ix_se.astype(F) - ix) * (iy_se.astype(F) - iy)).astype(F)
    ne = ((ix - ix_sw.astype(F)) * (iy_sw.astype(F) - iy)).astype(F)
    sw_ = ((ix_ne.astype(F) - ix) * (iy - iy_ne.astype(F))).astype(F)
    se = ((ix - ix_nw.astype(F)) * (iy - iy_nw.astype(F))).astype(F)
    d = desc_chw
    v = (d[:, iy_nw, ix_nw] * nw[None]).astype(F)
    v = (v + (d[:, iy_ne, ix_ne] * ne[None]).astype(F)).astype(F)
    v = (v + (d[:, iy_sw, ix_sw] * sw_[None]).astype(F)).astype(F)
    v = (v + (d[:, iy_se, ix_se] * se[None]).astype(F)).astype(F)      # [256, N]
    nrm = np.sqrt(_eigen_sse2_sum((v * v).astype(F))).astype(F)
    # Eigen >= 3.3 colwise().normalize(): `if (squaredNorm() > 0) derived() /= sqrt(...)` — a zero column stays zero
    v = np.where(nrm[None] > 0, (v / np.where(nrm > 0, nrm, F(1))[None]).astype(F), v).astype(F)
    return np.ascontiguousarray(v.T)


def _eigen_sse2_sum(sq: np.ndarray) -> np.ndarray:
    """Column sums of sq [n, N] float32 in the order Eigen 3.3 reduces a contiguous dynamic-size float column on SSE2 (Redux.h,
    LinearVectorizedTraversal: two 4-lane accumulators over pairs of packets, res0 += res1, a trailing packet, predux = (a0 + a2) + (a1 + a3),
    scalar tail) — what `colwise().normalize()` (src/plnet.cpp:415) does in the reference's build (no -march: SSE2 packets).  The same order
    is restated in shim/stubs/Eigen/Core, which oracle/_ref is compiled against; pinned against that build by tests/test_ref_pin_cpu.py."""
    n = sq.shape[0]
    if n < 4:
        out = np.zeros(sq.shape[1:], F)
        for k in range(n):
            out = (out + sq[k]).astype(F)
        return out
    a2, a1 = n // 8 * 8, n // 4 * 4
    r0 = sq[0:4].astype(F).copy()
    if a1 > 4:
        r1 = sq[4:8].astype(F).copy()
        for k in range(8, a2, 8):
            r0 = (r0 + sq[k:k + 4]).astype(F)
            r1 = (r1 + sq[k + 4:k + 8]).astype(F)
        r0 = (r0 + r1).astype(F)
        if a1 > a2:
            r0 = (r0 + sq[a2:a2 + 4]).astype(F)
    res = ((r0[0] + r0[2]).astype(F) + (r0[1] + r0[3]).astype(F)).astype(F)
    for k in range(a1, n):
        res = (res + sq[k]).astype(F)
    return res


def keypoints_decoder(heat, desc_chw, threshold, border, top_k, w_scale=F(1), h_scale=F(1)):
    """PLNet::keypoints_decoder src/plnet.cpp:419-423 + rescale :574-575 (≡ super_point.cpp:274-282).
    Returns feat [N, 259] float32; row n = [score, x*w_scale, y*h_scale, d0..d255]
    (byte-identical to column n of the reference's Eigen::Matrix<float,259,Dynamic>)."""
    s, x, y = detect_point(heat, threshold, border, top_k)
    d = extract_descriptors(desc_chw, x, y, 8)
    feat = np.empty((s.size, 259), dtype=F)
    feat[:, 0] = s
    feat[:, 1] = (x * F(w_scale)).astype(F)
    feat[:, 2] = (y * F(h_scale)).astype(F)
    feat[:, 3:] = d
    return feat


# ------------------------------------------------------------------ PLNet line path
def wireframe_matcher(iskeep, idx_min, idx_max, jn: int = 300):
    """PLNet::wireframe_matcher src/plnet.cpp:272-307.  Inputs flattened [3*128*128] float32.
    Returns (is_keep_index [M1], inverse [M1], unique_pairs [M2,2] as (max,min))."""
    keep = np.nonzero(iskeep.reshape(-1) > 0)[0]
    a = idx_min.reshape(-1)[keep].astype(np.int64)     # (int) cast truncates; values are whole
    b = idx_max.reshape(-1)[keep].astype(np.int64)
    code = a * jn + b
    _, first_pos, inv = np.unique(code, return_index=True, return_inverse=True)
    order = np.argsort(first_pos, kind="stable")       # unique ids in first-seen order
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    inverse = rank[inv]
    pairs = np.empty((order.size, 2), dtype=np.int64)
    ucode = code[first_pos[order]]
    pairs[:, 0] = ucode % jn       # (j, i) = (max, min)  plnet.cpp:301
    pairs[:, 1] = ucode // jn
    return keep.astype(np.int64), inverse.astype(np.int64), pairs


def line_filter(lines_adjusted, scores_line, border, line_threshold, line_length_threshold,
                rw: int = 512, rh: int = 512):
    """process_output src/plnet.cpp:519-558.  Returns (lines [L,4] float64 in 512-space, junction_map bool [rh,rw])."""
    jmap = np.zeros((rh, rw), dtype=bool)
    border = max(border, 0)
    thr2 = F(F(line_length_threshold) * F(line_length_threshold))
    out = []
    for i in range(lines_adjusted.shape[0]):
        if scores_line[i] < 0.5:
            continue
        x1 = F(lines_adjusted[i, 0] * F(4)); y1 = F(lines_adjusted[i, 1] * F(4))
        x2 = F(lines_adjusted[i, 2] * F(4)); y2 = F(lines_adjusted[i, 3] * F(4))
        xi1 = int(np.float64(x1) + 0.1); yi1 = int(np.float64(y1) + 0.1)
        xi2 = int(np.float64(x2) + 0.1); yi2 = int(np.float64(y2) + 0.1)
        p1 = (xi1 > border) and (xi1 < rw - border) and (yi1 > border) and (yi1 < rh - border)
        p2 = (xi2 > border) and (xi2 < rw - border) and (yi2 > border) and (yi2 < rh - border)
        jmap[yi1, xi1] = p1
        jmap[yi2, xi2] = p2
        if scores_line[i] < F(line_threshold):
            continue
        l2 = F(F((x2 - x1) * (x2 - x1)) + F((y2 - y1) * (y2 - y1)))
        if l2 < thr2:
            continue
        out.append((float(x1), float(y1), float(x2), float(y2)))
    return np.array(out, dtype=np.float64).reshape(-1, 4), jmap


def junction_detector(heat, desc_chw, jmap, border, w_scale=F(1), h_scale=F(1)):
    """PLNet::junction_detector src/plnet.cpp:425-448 (+ rescale :569-570): raster scan, upper bound EXCLUSIVE."""
    h, w = heat.shape
    border = max(border, 0)
    m = np.zeros_like(jmap)
    m[border:h - border, border:w - border] = jmap[border:h - border, border:w - border]
    ys, xs = np.nonzero(m)
    s = heat[ys, xs].astype(F)
    d = extract_descriptors(desc_chw, xs.astype(F), ys.astype(F), 8)
    feat = np.empty((s.size, 259), dtype=F)
    feat[:, 0] = s
    feat[:, 1] = (xs.astype(F) * F(w_scale)).astype(F)
    feat[:, 2] = (ys.astype(F) * F(h_scale)).astype(F)
    feat[:, 3:] = d
    return feat


def rescale_lines(lines512: np.ndarray, w_scale, h_scale) -> np.ndarray:
    """src/plnet.cpp:577-582: Vector4d *= float scale (double * float → double)."""
    out = lines512.astype(np.float64).copy()
    out[:, 0] *= float(F(w_scale)); out[:, 2] *= float(F(w_scale))
    out[:, 1] *= float(F(h_scale)); out[:, 3] *= float(F(h_scale))
    return out


# ------------------------------------------------------------------ matcher glue
def normalize_keypoints(feat: np.ndarray, width: int, height: int, scale: float) -> np.ndarray:
    """PointMatcher::NormalizeKeypoints src/point_matcher.cc:39-48.  feat [N,259]; integer width/2."""
    out = feat.copy()
    l_inv = F(1.0 / max(width, height) * float(F(scale)))
    out[:, 1] = ((feat[:, 1] - F(width // 2)) * l_inv).astype(F)
    out[:, 2] = ((feat[:, 2] - F(height // 2)) * l_inv).astype(F)
    return out


_FLT_MAX = np.finfo(np.float32).max


def _libm_expf():
    import ctypes
    import ctypes.util
    lib = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    lib.expf.restype = ctypes.c_float
    lib.expf.argtypes = [ctypes.c_float]
    return lib.expf


_LIBM_EXPF = None


def _expf(x: np.ndarray) -> np.ndarray:
    """`std::exp(float)` (src/light_glue.cpp:248, src/super_glue.cpp:299) = glibc's expf — called, not restated: this file runs on the host whose libm the
    compiled reference (oracle/_ref) links, so the oracle's exponential IS the reference's.  (numpy's float32 exp is a 1-ulp SIMD routine that decides
    `exp(s) > threshold` differently one ulp beside log(threshold); the double exponential rounded once — what this function returned in rounds 3-4 — is the
    correctly rounded value, which glibc's 0.502-ulp routine misses on 0.063 % of the inputs: tools/expf_glibc_check.c.  The device restates glibc's algorithm
    operation by operation, airslam_amd/csrc/common.h expf_like_glibc, verified against this libm on every float.)"""
    global _LIBM_EXPF
    if _LIBM_EXPF is None:
        _LIBM_EXPF = _libm_expf()
    a = np.asarray(x, F)                                  # (np.ascontiguousarray would turn a scalar into a 1-element vector)
    out = np.empty(a.shape, F)
    fi, fo = np.ascontiguousarray(a).reshape(-1), out.reshape(-1)
    for i in range(fi.size):                                # (at most ~1e3 values per pair: the row maxima / the kept matches)
        fo[i] = _LIBM_EXPF(float(fi[i]))
    return out[()] if out.ndim == 0 else out


def _first_max_above_floor(m: np.ndarray, axis: int):
    """The reference's arg-max loops: `max = -FLT_MAX; if (v > max) {max = v; arg = j}` — first maximum wins, and an entry
    that does not EXCEED -FLT_MAX (-inf, -FLT_MAX itself, NaN) never becomes the maximum.  Returns (arg, max, found)."""
    above = m > -_FLT_MAX                                  # NaN compares false, like the C++
    mm = np.where(above, m, -np.inf)
    arg = np.argmax(mm, axis=axis)                         # np.argmax returns the first maximum
    found = above.any(axis=axis)
    val = np.take_along_axis(mm, np.expand_dims(arg, axis), axis).squeeze(axis)
    return np.where(found, arg, 0), val, found


def filter_matches(scores: np.ndarray, threshold: float = 0.1):
    """filter_matches src/light_glue.cpp:214-266: strict '>' from -FLT_MAX (first max wins), mutual check,
    exp(max) > threshold, ascending row order.  A row/column with nothing above -FLT_MAX keeps the value-initialised
    pair (0, 0.0f) that `row_max.resize()` gave it (:217,232) — i.e. it points at column/row 0 with score 0 -> exp = 1.
    Returns (idx [K,2] int32, score [K] float32)."""
    n0, n1 = scores.shape
    if n0 == 0 or n1 == 0:
        return np.zeros((0, 2), np.int32), np.zeros((0,), F)
    s = scores.astype(F)
    rcol, rval, rfound = _first_max_above_floor(s, 1)
    rval = np.where(rfound, rval, F(0)).astype(F)
    crow, _, _ = _first_max_above_floor(s, 0)
    rows = np.arange(n0)
    mutual = crow[rcol] == rows
    e = _expf(rval)
    ok = mutual & (e > F(threshold))
    idx = np.stack([rows[ok], rcol[ok]], axis=1).astype(np.int32)
    return idx, e[ok].astype(F)


def superglue_decode(scores: np.ndarray, threshold: float = 0.2):
    """decode src/super_glue.cpp:339-367 on scores [h,w] (h=N0+1, w=N1+1): uses rows 0..h-2, cols 0..w-2;
    max_matrix (:258-286): strict '<' from -FLT_MAX, index 0 / value -FLT_MAX when nothing exceeds it."""
    h, w = scores.shape
    inner = scores[:h - 1, :w - 1].astype(F)
    if inner.size == 0:
        return (np.zeros(h - 1, np.int32), np.zeros(w - 1, np.int32),
                np.zeros(h - 1, np.float64), np.zeros(w - 1, np.float64))
    idx0, max0, f0 = _first_max_above_floor(inner, 1)
    max0 = np.where(f0, max0, -_FLT_MAX).astype(F)
    idx1, _, _ = _first_max_above_floor(inner, 0)
    mutual0 = idx1[idx0] == np.arange(h - 1)
    mutual1 = idx0[idx1] == np.arange(w - 1)
    with np.errstate(under="ignore"):
        ms0 = np.where(mutual0, _expf(max0), F(0)).astype(F)
    ms1 = np.where(mutual1, ms0[idx1], F(0)).astype(F)
    valid0 = mutual0 & (ms0 > F(threshold))
    valid1 = mutual1 & valid0[idx1]
    i0 = np.where(valid0, idx0, -1).astype(np.int32)
    i1 = np.where(valid1, idx1, -1).astype(np.int32)
    return i0, i1, ms0.astype(np.float64), ms1.astype(np.float64)


def superglue_matches(i0, i1, ms0, ms1):
    """PointMatcher::MatchingPoints superglue branch src/point_matcher.cc:82-91 → list of (q, t, dist)."""
    out = []
    for i in range(len(i0)):
        if 0 <= i0[i] < len(i1) and i1[i0[i]] == i:
            out.append((i, int(i0[i]), 1.0 - (ms0[i] + ms1[i0[i]]) / 2.0))
    return out


def log_optimal_transport(scores: np.ndarray, alpha: float = 2.3457, iters: int = 100) -> np.ndarray:
    """Dead-code CPU Sinkhorn src/super_glue.cpp:369-435 (float32, sequential sums replaced by
    float32 logsumexp-free direct exp sums exactly as the C++ does: no max subtraction)."""
    m, n = scores.shape
    c = np.full((m + 1, n + 1), F(alpha), dtype=F)
    c[:m, :n] = scores
    norm = F(-np.log(F(m + n)))
    log_mu = np.full(m + 1, norm, dtype=F); log_mu[m] = F(np.log(F(n)) + norm)
    log_nu = np.full(n + 1, norm, dtype=F); log_nu[n] = F(np.log(F(m)) + norm)
    u = np.zeros(m + 1, dtype=F); v = np.zeros(n + 1, dtype=F)
    for _ in range(iters):
        u = (log_mu - np.log(np.sum(np.exp(c + v[None, :]), axis=1, dtype=F))).astype(F)
        v = (log_nu - np.log(np.sum(np.exp(c + u[:, None]), axis=0, dtype=F))).astype(F)
    return (c + u[:, None] + v[None, :] - norm).astype(F)


# ---------------------------------------------------------------------------------------------------------------------
# point <-> line association
def assign_points_to_lines(lines: np.ndarray, points: np.ndarray):
    """AssignPointsToLines, src/line_processor.cc:68-120, statement by statement in float64 (the reference casts the
    float32 keypoints to double, :70) with the point-line distance narrowed to float32 where the reference does (:106).
    lines [L,4] (x1,y1,x2,y2) float64, points [259,N] or [N,259]-style rows given as [N,259] float32.
    -> list of L dicts {point index j: distance} in ascending j (std::map<int,double> order)."""
    lines = np.asarray(lines, dtype=np.float64).reshape(-1, 4)
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 259)
    x = pts[:, 1].astype(np.float64)
    y = pts[:, 2].astype(np.float64)
    x1, y1, x2, y2 = lines[:, 0], lines[:, 1], lines[:, 2], lines[:, 3]
    A = y2 - y1                                   # :81
    B = x1 - x2                                   # :82
    C = x2 * y1 - x1 * y2                         # :83
    D = np.sqrt(A * A + B * B)                    # :84
    relation = []
    for i in range(lines.shape[0]):
        on_line = {}
        lx1, ly1, lx2, ly2 = x1[i], y1[i], x2[i], y2[i]
        min_lx, max_lx = (lx2, lx1) if lx1 > lx2 else (lx1, lx2)          # :100-103
        min_ly, max_ly = (ly2, ly1) if ly1 > ly2 else (ly1, ly2)
        for j in range(pts.shape[0]):
            px, py = x[j], y[j]
            if px < min_lx - 3 or px > max_lx + 3 or py < min_ly - 3 or py > max_ly + 3:   # :104
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                pl = np.float32(np.abs(A[i] * px + B[i] * py + C[i]) / D[i])                # :107 (float)
            if pl > 3:                                                                      # :108 (NaN compares false)
                continue
            side1 = (lx1 - px) * (lx1 - px) + (ly1 - py) * (ly1 - py)                       # :110
            side2 = (lx2 - px) * (lx2 - px) + (ly2 - py) * (ly2 - py)                       # :111
            line_side = D[i] * D[i]                                                         # :112
            if side1 <= 9 or side2 <= 9 or (side1 < line_side + side2 and side2 < line_side + side1):   # :113
                on_line[j] = float(pl)                                                      # :114
        relation.append(on_line)
    return relation


def match_lines(points_on_line0, points_on_line1, point_matches, point_num0: int, point_num1: int):
    """MatchLines, src/line_processor.cc:122-172, statement by statement.  points_on_line{0,1}: lists of dicts {point index:
    distance} (the std::map<int,double> relation of AssignPointsToLines); point_matches: iterable of (queryIdx, trainIdx).
    -> list of length len(points_on_line0): matched line index of frame 1, or -1."""
    n0, n1 = len(points_on_line0), len(points_on_line1)
    line_matches = [-1] * n0                                                     # :127-131
    if point_num0 == 0 or point_num1 == 0 or n0 == 0 or n1 == 0:                # :132
        return line_matches
    assigned0 = [[] for _ in range(point_num0)]                                  # :134-147
    assigned1 = [[] for _ in range(point_num1)]
    for i, rel in enumerate(points_on_line0):
        for k in rel:
            assigned0[k].append(i)
    for i, rel in enumerate(points_on_line1):
        for k in rel:
            assigned1[k].append(i)
    mat = np.zeros((n0, n1), np.int32)                                           # :150
    for q, t in point_matches:                                                   # :151-160
        for l0 in assigned0[q]:
            for l1 in assigned1[t]:
                mat[l0, l1] += 1
    row_loc = [int(np.argmax(mat[i])) for i in range(n0)]                        # :166-168 (maxCoeff: first maximum)
    for j in range(n1):                                                          # :169-179
        col = mat[:, j]
        i = int(np.argmax(col))                                                  # first maximum
        v = int(col[i])
        if v < 2 or row_loc[i] != j:
            continue
        score = np.float32(v * v) / np.float32(min(len(points_on_line0[i]), len(points_on_line1[j])))   # :174 float / size_t
        if float(score) < 0.8:
            continue
        line_matches[i] = j
    return line_matches
