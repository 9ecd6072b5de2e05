"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

ctypes binding of oracle/_ref/libairslam_ref.so = the REFERENCE's own front-end sources compiled unchanged
(oracle/Makefile, oracle/ref_driver.cpp): FeatureDetector::Detect x 6, PointMatcher::MatchingPoints / NormalizeKeypoints,
PLNet::infer, SuperPoint::infer, SuperPointLightGlue::infer, SuperGlue::infer and everything they call on the host, plus
filter_matches, decode, log_optimal_transport, AssignPointsToLines, MatchLines.  The TensorRT engines are a callback: `engines`
maps a model kind to a Python callable `fn(inputs: dict[str, ndarray]) -> dict[str, ndarray]`.

available() is False when the library has not been built (no /root/reference on this machine and no prebuilt copy).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libairslam_ref.so")
_ENGINE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                         C.POINTER(C.c_int), C.POINTER(C.c_void_p))
_lib = None
_keep = {}


def available() -> bool:
    return os.path.exists(LIB)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB)
        L.airslam_ref_detector_create.restype = C.c_void_p
        L.airslam_ref_detector_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float]
        L.airslam_ref_detector_create_from_yaml.restype = C.c_void_p
        L.airslam_ref_detector_create_from_yaml.argtypes = [C.c_char_p, C.c_char_p] + [C.POINTER(C.c_int)] * 2 + [C.POINTER(C.c_float), C.POINTER(C.c_int)] + [C.POINTER(C.c_float)] * 2
        L.airslam_ref_detector_destroy.argtypes = [C.c_void_p]
        L.airslam_ref_matcher_create.restype = C.c_void_p
        L.airslam_ref_matcher_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.airslam_ref_matcher_destroy.argtypes = [C.c_void_p]
        L.airslam_ref_set_engine.argtypes = [_ENGINE_FN, C.c_void_p]
        L.airslam_ref_sources.restype = C.c_char_p
        L.airslam_ref_point_line_distance.restype = C.c_float
        _lib = L
    return _lib


def set_engines(engines: dict):
    """Install the engine callback.  engines: {"plnet_s0" | "plnet_s1" | "superpoint" | "lightglue" | "superglue": fn}."""
    calls = []

    def cb(_user, model, nb, names, is_input, ndims, dims, bufs):
        try:
            kind = model.decode()
            ins, outs = {}, {}
            for i in range(nb):
                shape = tuple(dims[8 * i + k] for k in range(ndims[i]))
                n = int(np.prod(shape)) if shape else 1
                arr = np.ctypeslib.as_array(C.cast(bufs[i], C.POINTER(C.c_float)), shape=(n,)).reshape(shape) if n else np.zeros(shape, np.float32)
                (ins if is_input[i] else outs)[names[i].decode()] = arr
            calls.append((kind, {k: v.copy() for k, v in ins.items()}))
            res = engines[kind](ins)
            for k, dst in outs.items():
                src = np.asarray(res[k], np.float32)
                assert src.size == dst.size, f"{kind}.{k}: engine returned {src.shape}, binding is {dst.shape}"
                dst[...] = src.reshape(dst.shape)
            return 0
        except Exception as e:                       # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            print("ref_lib engine callback failed:", e)
            return 1

    fn = _ENGINE_FN(cb)
    _keep["engine"] = fn
    lib().airslam_ref_set_engine(fn, None)
    return calls


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class FeatureDetector:
    """The reference's FeatureDetector (src/feature_detector.cc) on the engines installed with set_engines()."""

    def __init__(self, model_dir: str, use_superpoint=0, max_keypoints=400, keypoint_threshold=0.004, remove_borders=4, line_threshold=0.75,
                 line_length_threshold=50.0, yaml: str | None = None):
        os.makedirs(model_dir, exist_ok=True)
        if yaml:
            iv = [C.c_int() for _ in range(3)]
            fv = [C.c_float() for _ in range(3)]
            self.h = lib().airslam_ref_detector_create_from_yaml(yaml.encode(), model_dir.encode(), C.byref(iv[0]), C.byref(iv[1]), C.byref(fv[0]),
                                                                 C.byref(iv[2]), C.byref(fv[1]), C.byref(fv[2]))
            self.cfg = dict(use_superpoint=iv[0].value, max_keypoints=iv[1].value, keypoint_threshold=fv[0].value, remove_borders=iv[2].value,
                            line_threshold=fv[1].value, line_length_threshold=fv[2].value)
        else:
            self.h = lib().airslam_ref_detector_create(model_dir.encode(), use_superpoint, max_keypoints, keypoint_threshold, remove_borders,
                                                       line_threshold, line_length_threshold)
            self.cfg = dict(use_superpoint=use_superpoint, max_keypoints=max_keypoints, keypoint_threshold=keypoint_threshold,
                            remove_borders=remove_borders, line_threshold=line_threshold, line_length_threshold=line_length_threshold)

    def close(self):
        if self.h:
            lib().airslam_ref_detector_destroy(self.h)
            self.h = None

    def detect(self, overload: int, left: np.ndarray, right: np.ndarray | None = None, lines_in: np.ndarray | None = None, cap=4096, cap_lines=50000,
               cap_j=8192):
        """Detect overload 0..5 (source order, src/feature_detector.cc:36,52,62,71,83,97).  Returns dict(ok, feat_l [n][259], feat_r, lines_l [m][4]
        (lines_in in front), lines_r, junc [k][259])."""
        h, w = left.shape
        stride = left.strides[0]
        fl = np.zeros((cap, 259), np.float32); fr = np.zeros((cap, 259), np.float32); jn = np.zeros((cap_j, 259), np.float32)
        ll = np.zeros((cap_lines, 4), np.float64); lr = np.zeros((cap_lines, 4), np.float64)
        nin = 0
        if lines_in is not None and len(lines_in):
            nin = len(lines_in)
            ll[:nin] = lines_in
        n = [C.c_int() for _ in range(5)]
        rp = right.ctypes.data_as(C.POINTER(C.c_uint8)) if right is not None else None
        if right is not None:
            assert right.shape == left.shape and right.strides[0] == stride
        ok = lib().airslam_ref_detect(C.c_void_p(self.h), overload, left.ctypes.data_as(C.POINTER(C.c_uint8)), rp, h, w, stride,
                                      _fp(fl), C.byref(n[0]), _fp(fr), C.byref(n[1]), cap, _dp(ll), nin, C.byref(n[2]), _dp(lr), C.byref(n[3]),
                                      cap_lines, _fp(jn), C.byref(n[4]), cap_j)
        assert n[0].value <= cap and n[1].value <= cap and n[2].value <= cap_lines and n[3].value <= cap_lines and n[4].value <= cap_j
        return dict(ok=bool(ok), feat_l=fl[:n[0].value].copy(), feat_r=fr[:n[1].value].copy(), lines_l=ll[:n[2].value].copy(),
                    lines_r=lr[:n[3].value].copy(), junc=jn[:n[4].value].copy())


class PointMatcher:
    """The reference's PointMatcher (src/point_matcher.cc); matcher 0 = LightGlue, 1 = SuperGlue."""

    def __init__(self, model_dir: str, matcher: int, image_width: int, image_height: int):
        os.makedirs(model_dir, exist_ok=True)
        self.h = lib().airslam_ref_matcher_create(model_dir.encode(), matcher, image_width, image_height)

    def close(self):
        if self.h:
            lib().airslam_ref_matcher_destroy(self.h)
            self.h = None

    def normalize_keypoints(self, feat: np.ndarray, width: int, height: int, scale: float) -> np.ndarray:
        feat = np.ascontiguousarray(feat, np.float32)
        out = np.zeros_like(feat)
        lib().airslam_ref_normalize_keypoints(C.c_void_p(self.h), _fp(feat), len(feat), width, height, C.c_float(scale), _fp(out))
        return out

    def matching_points(self, f0: np.ndarray, f1: np.ndarray, outlier_rejection=False):
        """f0, f1: [n][259] rows (= columns of the reference's 259 x N matrices).  Returns (count, [(query, train, distance)])."""
        f0 = np.ascontiguousarray(f0, np.float32); f1 = np.ascontiguousarray(f1, np.float32)
        cap = max(len(f0), len(f1), 1)
        q = np.zeros(cap, np.int32); t = np.zeros(cap, np.int32); d = np.zeros(cap, np.float32)
        r = lib().airslam_ref_matching_points(C.c_void_p(self.h), _fp(f0), len(f0), _fp(f1), len(f1), _ip(q), _ip(t), _fp(d), cap, int(outlier_rejection))
        return r, q[:r].copy(), t[:r].copy(), d[:r].copy()


def filter_matches(scores: np.ndarray, threshold: float = 0.1):
    s = np.ascontiguousarray(scores, np.float32)
    n0, n1 = s.shape
    cap = max(min(n0, n1), 1)
    idx = np.zeros((max(n0, 1), 2), np.int32); sc = np.zeros(max(n0, 1), np.float32)
    k = lib().airslam_ref_filter_matches(_fp(s), n0, n1, C.c_float(threshold), _ip(idx), _fp(sc))
    del cap
    return idx[:k].copy(), sc[:k].copy()


def superglue_decode(scores: np.ndarray):
    """decode (src/super_glue.cpp:339-367) on the full [h][w] matrix incl. dustbins -> indices0 [h-1], indices1 [w-1], mscores0, mscores1 (float,
    as the reference's std::vector<float>; process_output widens them to double)."""
    s = np.ascontiguousarray(scores, np.float32)
    h, w = s.shape
    i0 = np.zeros(h - 1, np.int32); i1 = np.zeros(w - 1, np.int32); m0 = np.zeros(h - 1, np.float32); m1 = np.zeros(w - 1, np.float32)
    lib().airslam_ref_sg_decode(_fp(s), h, w, _ip(i0), _ip(i1), _fp(m0), _fp(m1))
    return i0, i1, m0, m1


def log_optimal_transport(scores: np.ndarray, alpha: float = 2.3457, iters: int = 100) -> np.ndarray:
    s = np.ascontiguousarray(scores, np.float32)
    m, n = s.shape
    z = np.zeros((m + 1, n + 1), np.float32)
    lib().airslam_ref_log_optimal_transport(_fp(s), m, n, C.c_float(alpha), iters, _fp(z))
    return z


def assign_points_to_lines(lines: np.ndarray, feat: np.ndarray):
    """AssignPointsToLines (src/line_processor.cc:68-120): lines [nl][4] double, feat [n][259] -> (offsets [nl+1], point idx, distance)."""
    lines = np.ascontiguousarray(lines, np.float64); feat = np.ascontiguousarray(feat, np.float32)
    nl, n = len(lines), len(feat)
    cap = max(nl * n, 1)
    off = np.zeros(nl + 1, np.int32); pi = np.zeros(cap, np.int32); pd = np.zeros(cap, np.float64)
    k = lib().airslam_ref_assign_points_to_lines(_dp(lines), nl, _fp(feat), n, _ip(off), _ip(pi), _dp(pd), cap)
    return off, pi[:k].copy(), pd[:k].copy()


def match_lines(off0, pidx0, off1, pidx1, query, train, point_num0: int, point_num1: int) -> np.ndarray:
    off0 = np.ascontiguousarray(off0, np.int32); pidx0 = np.ascontiguousarray(pidx0, np.int32)
    off1 = np.ascontiguousarray(off1, np.int32); pidx1 = np.ascontiguousarray(pidx1, np.int32)
    query = np.ascontiguousarray(query, np.int32); train = np.ascontiguousarray(train, np.int32)
    nl0, nl1 = len(off0) - 1, len(off1) - 1
    out = np.zeros(max(nl0, 1), np.int32)
    lib().airslam_ref_match_lines(_ip(off0), _ip(pidx0), nl0, _ip(off1), _ip(pidx1), nl1, _ip(query), _ip(train), len(query), point_num0, point_num1, _ip(out))
    return out[:nl0].copy()


def bow_frame_to_bow(voc: dict, feat: np.ndarray):
    """Database::FrameToBow's per-feature part (src/bow/database.cc:57-89) on the vendored DBoW2 compiled unchanged (oracle/ref_bow.cpp): feat [N][259]
    -> (word_of_features uint32 [N] (UINT_MAX: stopped word), weight_of_features f64 [N], BowVector as (ids uint32 [K], values f64 [K]))."""
    feat = np.ascontiguousarray(feat, np.float32).reshape(-1, 259)
    n = len(feat)
    desc = np.ascontiguousarray(voc["desc"], np.float32); fc = np.ascontiguousarray(voc["first_child"], np.int32)
    nc = np.ascontiguousarray(voc["n_children"], np.int32); wi = np.ascontiguousarray(voc["word_id"], np.int32)
    w = np.ascontiguousarray(voc["weight"], np.float64)
    words = np.zeros(max(n, 1), np.uint32); wts = np.zeros(max(n, 1), np.float64); ids = np.zeros(max(n, 1), np.uint32); vals = np.zeros(max(n, 1), np.float64)
    fn = lib().airslam_ref_bow_frame_to_bow
    fn.restype = C.c_int
    k = fn(_fp(desc), _ip(fc), _ip(nc), _ip(wi), _dp(w), len(desc), int(voc.get("k", 10)), int(voc.get("L", 4)), _fp(feat), n,
           words.ctypes.data_as(C.c_void_p), _dp(wts), ids.ctypes.data_as(C.c_void_p), _dp(vals))
    return words[:n].copy(), wts[:n].copy(), ids[:k].copy(), vals[:k].copy()


def sources() -> str:
    return lib().airslam_ref_sources().decode()
