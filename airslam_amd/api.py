"""Host-side mirror of the reference's front-end interface on top of the libairfe.so C ABI.

`FeatureDetector.Detect` and `PointMatcher.MatchingPoints` keep the names, argument meaning and error behaviour
of include/feature_detector.h:8-31 and include/point_matcher.h:8-24 (bool / match-count returns, printed
messages, early-outs); feature matrices are numpy [259, N] float32 in the reference's Eigen orientation
(column = keypoint), stored Fortran-contiguous so they are byte-identical to Eigen's column-major buffer.

torch is used only as plumbing for the device-resident batch entry points (`Context.*_dev`).
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile
from typing import Dict, Optional

import numpy as np

from . import _lib, weights as W

FEAT = 259


def _pack_arg(x, tmpfiles) -> Optional[bytes]:
    if x is None:
        return None
    if isinstance(x, (str, bytes, os.PathLike)):
        return os.fsencode(x)
    f = tempfile.NamedTemporaryFile(suffix=".airfe", delete=False)
    f.close()
    W.save_pack(f.name, x)
    tmpfiles.append(f.name)
    return f.name.encode()


class AirfeError(RuntimeError):
    pass


class Context:
    """One airfe_ctx: one device, one stream, one calling thread."""

    def __init__(self, superpoint=None, lightglue=None, superglue=None, plnet_s1=None, tuning=None, **cfg):
        """cfg: airfe_cfg fields; tuning: dict of airfe_tuning fields (kernel-selection overrides for A/B runs and tests; the library reads no environment)."""
        self._l = _lib.lib()
        c = _lib.Cfg()
        self._l.airfe_default_cfg(C.byref(c))
        t = None
        if tuning:
            t = _lib.Tuning()
            self._l.airfe_default_tuning(C.byref(t))
            for k, v in tuning.items():
                if k == "reserved" or not hasattr(t, k):
                    raise TypeError(f"unknown airfe_tuning field {k!r}")
                setattr(t, k, int(v))
            c.tuning = C.pointer(t)
        for k, v in cfg.items():
            if not hasattr(c, k):
                raise TypeError(f"unknown airfe_cfg field {k!r}")
            setattr(c, k, v)
        tmp = []
        try:
            c.superpoint_pack = _pack_arg(superpoint, tmp)
            c.lightglue_pack = _pack_arg(lightglue, tmp)
            c.superglue_pack = _pack_arg(superglue, tmp)
            c.plnet_s1_pack = _pack_arg(plnet_s1, tmp)
            h = C.c_void_p()
            rc = self._l.airfe_create(C.byref(c), C.byref(h))
            if rc != 0:
                raise AirfeError((self._l.airfe_last_error(None) or b"airfe_create failed").decode())
        finally:
            for t in tmp:
                os.unlink(t)
        self._h = h
        c.tuning = None                # (read by airfe_create only)
        self.cfg = c
        self.max_keypoints = c.max_keypoints
        self.np_rows = (c.max_keypoints + 15) // 16 * 16

    def close(self):
        if getattr(self, "_h", None):
            self._l.airfe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise AirfeError(f"{what}: {(self._l.airfe_last_error(self._h) or b'').decode()}")

    # ---------------------------------------------------------------- host, batch-1 (≙ reference infer())
    def detect_points(self, gray: np.ndarray) -> np.ndarray:
        """-> [n, 259] float32 rows (score, x, y, desc).  Raises on an empty image."""
        gray = np.asarray(gray)
        if gray.ndim != 2 or gray.dtype != np.uint8:
            raise TypeError("expected a 2-D uint8 image")
        if gray.size == 0:
            raise AirfeError("empty image")
        if gray.strides[1] != 1 or gray.strides[0] < gray.shape[1]:      # negative / overlapping row strides: hand over a copy
            gray = np.ascontiguousarray(gray)
        cap = self.np_rows
        feat = np.empty((cap, FEAT), dtype=np.float32)
        n = C.c_int(0)
        self._chk(self._l.airfe_detect_points(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                              feat.ctypes.data, cap, C.byref(n)), "airfe_detect_points")
        return feat[:n.value].copy()

    def bow_load(self, voc: dict):
        """voc: dict(desc [n,256] f32, first_child [n] i32, n_children [n] i32, word_id [n] i32, weight [n] f64) — weights.synthetic_vocabulary."""
        d = np.ascontiguousarray(voc["desc"], np.float32); fc = np.ascontiguousarray(voc["first_child"], np.int32)
        nc = np.ascontiguousarray(voc["n_children"], np.int32); wi = np.ascontiguousarray(voc["word_id"], np.int32)
        w = np.ascontiguousarray(voc["weight"], np.float64)
        self._chk(self._l.airfe_bow_load(self._h, d.ctypes.data, fc.ctypes.data, nc.ctypes.data, wi.ctypes.data, w.ctypes.data, d.shape[0]),
                  "airfe_bow_load")

    def bow_transform(self, feat_rows: np.ndarray):
        """≙ the per-feature loop of Database::FrameToBow -> (word_of_features [N] uint32 (UINT_MAX = stopped), weights [N] float64)."""
        f = np.ascontiguousarray(feat_rows, np.float32).reshape(-1, FEAT)
        wid = np.empty((f.shape[0],), np.uint32); w = np.empty((f.shape[0],), np.float64)
        self._chk(self._l.airfe_bow_transform(self._h, f.ctypes.data, f.shape[0], wid.ctypes.data, w.ctypes.data), "airfe_bow_transform")
        return wid, w

    def set_rectify_maps(self, side: int, mapx: np.ndarray, mapy: np.ndarray):
        """≙ Camera's cv::initUndistortRectifyMap outputs (_mapl1/_mapl2 = side 0, _mapr1/_mapr2 = side 1): float32 [h, w] maps."""
        mx = np.ascontiguousarray(mapx, np.float32); my = np.ascontiguousarray(mapy, np.float32)
        if mx.shape != my.shape or mx.ndim != 2:
            raise TypeError("maps must be two 2-D arrays of one shape")
        self._chk(self._l.airfe_set_rectify_maps(self._h, side, mx.ctypes.data, my.ctypes.data, mx.shape[0], mx.shape[1]), "airfe_set_rectify_maps")

    def rectify_detect(self, side: int, raw: np.ndarray, detect: bool = True):
        """≙ Camera::UndistortImage + Detect: raw uint8 image -> (rectified uint8 [h, w], features [n, 259] or None)."""
        raw = np.asarray(raw)
        if raw.ndim != 2 or raw.dtype != np.uint8 or raw.size == 0:
            raise AirfeError("empty image")
        if raw.strides[1] != 1 or raw.strides[0] < raw.shape[1]:
            raw = np.ascontiguousarray(raw)
        rect = np.empty(raw.shape, np.uint8)
        cap = self.np_rows
        feat = np.empty((cap, FEAT), np.float32) if detect else None
        n = C.c_int(0)
        self._chk(self._l.airfe_rectify_detect_points(self._h, side, raw.ctypes.data, raw.shape[0], raw.shape[1], raw.strides[0],
                                                      rect.ctypes.data, feat.ctypes.data if detect else None, cap, C.byref(n)),
                  "airfe_rectify_detect_points")
        return rect, (feat[:n.value].copy() if detect else None)

    def match_lightglue(self, f0: np.ndarray, f1: np.ndarray):
        """f0/f1: [n, 258] rows (normalised x, y, desc) -> (idx [k,2] int32, score [k] float32)."""
        f0 = np.ascontiguousarray(f0, dtype=np.float32)
        f1 = np.ascontiguousarray(f1, dtype=np.float32)
        cap = self.np_rows
        idx = np.empty((cap, 2), dtype=np.int32)
        sc = np.empty((cap,), dtype=np.float32)
        n = C.c_int(0)
        self._chk(self._l.airfe_match_lightglue(self._h, f0.ctypes.data, f0.shape[0], f1.ctypes.data, f1.shape[0],
                                                idx.ctypes.data, sc.ctypes.data, cap, C.byref(n)), "airfe_match_lightglue")
        return idx[:n.value].copy(), sc[:n.value].copy()

    def lightglue_scores(self, f0: np.ndarray, f1: np.ndarray) -> np.ndarray:
        f0 = np.ascontiguousarray(f0, dtype=np.float32)
        f1 = np.ascontiguousarray(f1, dtype=np.float32)
        out = np.empty((f0.shape[0], f1.shape[0]), dtype=np.float32)
        self._chk(self._l.airfe_debug_lightglue_scores(self._h, f0.ctypes.data, f0.shape[0], f1.ctypes.data, f1.shape[0],
                                                       out.ctypes.data), "airfe_debug_lightglue_scores")
        return out

    @staticmethod
    def _stage0(s0):
        """dict of numpy arrays (synth.plnet_stage0_lines layout) -> (_lib.Stage0, keep-alive list)"""
        st = _lib.Stage0()
        keep = []
        for name, _ in _lib.Stage0._fields_:
            a = np.ascontiguousarray(s0[name], dtype=np.float32)
            keep.append(a)
            setattr(st, name, a.ctypes.data)
        return st, keep

    def detect_plnet(self, gray: np.ndarray, stage0=None, want_junctions: bool = False, cap_lines: int = 45056,
                     cap_junc: int = 2048):
        """≙ PLNet::infer -> (feat [n,259], lines [L,4] float64, junctions [K,259])."""
        gray = np.asarray(gray)
        if gray.ndim != 2 or gray.dtype != np.uint8 or gray.size == 0:
            raise AirfeError("empty image")
        if gray.strides[1] != 1 or gray.strides[0] < gray.shape[1]:
            gray = np.ascontiguousarray(gray)
        cap = self.np_rows
        feat = np.empty((cap, FEAT), np.float32)
        lines = np.empty((cap_lines, 4), np.float64)
        junc = np.empty((cap_junc, FEAT), np.float32)
        n, nl, nj = C.c_int(0), C.c_int(0), C.c_int(0)
        st, keep = (self._stage0(stage0) if stage0 is not None else (None, None))
        self._chk(self._l.airfe_detect_plnet(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                             C.byref(st) if st is not None else None, feat.ctypes.data, cap, C.byref(n),
                                             lines.ctypes.data, cap_lines, C.byref(nl), junc.ctypes.data, cap_junc,
                                             C.byref(nj), int(want_junctions)), "airfe_detect_plnet")
        return feat[:n.value].copy(), lines[:nl.value].copy(), junc[:nj.value].copy()

    def stereo_keyframe(self, left: np.ndarray, right: np.ndarray, match: bool = True, want_junctions: bool = True, cap_lines: int = 4096,
                        cap_junc: int = 2048, track: bool = False, ref_feat=None):
        """ONE stereo keyframe in one call (airfe_stereo_keyframe ≙ map_builder.cc:85-86): -> dict(featL, featR [n,259], linesL, linesR [L,4] float64,
        juncL [K,259], idx [m,2] int32, score [m]) — idx / score absent with match=False.  track=True (airfe_stereo_keyframe_tracked): also the temporal
        match of map_builder.cc:96 against the last keyframe's features (`ref_feat` [n,259]: uploaded when given, else the ones on the device) in the SAME
        LightGlue forward -> track_idx [t,2] (reference, left), track_score [t]."""
        if ref_feat is not None and not track:
            raise AirfeError("stereo_keyframe: ref_feat is the temporal match's reference: pass track=True")
        if track and not match:
            raise AirfeError("stereo_keyframe: track=True needs match=True (the temporal pair rides in the stereo match's forward)")
        imgs = []
        for g in (left, right):
            g = np.asarray(g)
            if g.ndim != 2 or g.dtype != np.uint8 or g.size == 0:
                raise AirfeError("empty image")
            imgs.append(g)
        if imgs[0].shape != imgs[1].shape:
            raise AirfeError("stereo_keyframe: left and right images differ in size")
        if any(g.strides[1] != 1 or g.strides[0] < g.shape[1] for g in imgs) or imgs[0].strides[0] != imgs[1].strides[0]:
            imgs = [np.ascontiguousarray(g) for g in imgs]
        # fresh output arrays, filled by the library directly (like the reference's caller-owned Eigen matrices): the slices returned below own them
        cap = self.np_rows
        fl, fr = np.empty((cap, FEAT), np.float32), np.empty((cap, FEAT), np.float32)
        ll, lr = np.empty((cap_lines, 4), np.float64), np.empty((cap_lines, 4), np.float64)
        jl = np.empty((cap_junc if want_junctions else 0, FEAT), np.float32)
        idx, sc = np.empty((cap if match else 0, 2), np.int32), np.empty((cap if match else 0,), np.float32)
        n = (C.c_int * 6)()
        p = lambda i: C.cast(C.byref(n, 4 * i), C.POINTER(C.c_int))
        args = (self._h, imgs[0].ctypes.data, imgs[1].ctypes.data, imgs[0].shape[0], imgs[0].shape[1], imgs[0].strides[0],
                fl.ctypes.data, fr.ctypes.data, cap, p(0), p(1), ll.ctypes.data, lr.ctypes.data, cap_lines, p(2), p(3),
                jl.ctypes.data if want_junctions else None, cap_junc, p(4), idx.ctypes.data if match else None, sc.ctypes.data, cap, p(5))
        if track:
            ref = None if ref_feat is None else np.ascontiguousarray(ref_feat, dtype=np.float32).reshape(-1, FEAT)
            tidx, tsc, nt = np.empty((cap, 2), np.int32), np.empty((cap,), np.float32), C.c_int(0)
            self._chk(self._l.airfe_stereo_keyframe_tracked(*args, None if ref is None else ref.ctypes.data, 0 if ref is None else len(ref), tidx.ctypes.data,
                                                            tsc.ctypes.data, C.byref(nt)), "airfe_stereo_keyframe_tracked")
        else:
            self._chk(self._l.airfe_stereo_keyframe(*args), "airfe_stereo_keyframe")
        out = dict(featL=fl[:n[0]], featR=fr[:n[1]], linesL=ll[:n[2]], linesR=lr[:n[3]], juncL=jl[:n[4]])
        if match:
            out["idx"], out["score"] = idx[:n[5]], sc[:n[5]]
        if track:
            out["track_idx"], out["track_score"] = tidx[:nt.value], tsc[:nt.value]
        return out

    def track_frame(self, gray: np.ndarray, ref_feat=None):
        """ONE tracked frame in one call (airfe_track_frame ≙ map_builder.cc:94-101): points of `gray` + LightGlue against the last keyframe's features
        (`ref_feat` [n,259]: uploaded when given, kept on the device when None) -> (feat [n,259], idx [m,2] (reference, new), score [m])."""
        gray = np.asarray(gray)
        if gray.ndim != 2 or gray.dtype != np.uint8 or gray.size == 0:
            raise AirfeError("empty image")
        if gray.strides[1] != 1 or gray.strides[0] < gray.shape[1]:
            gray = np.ascontiguousarray(gray)
        cap = self.np_rows
        feat, idx, sc = np.empty((cap, FEAT), np.float32), np.empty((cap, 2), np.int32), np.empty((cap,), np.float32)
        n, nm = C.c_int(0), C.c_int(0)
        ref = None if ref_feat is None else np.ascontiguousarray(ref_feat, dtype=np.float32).reshape(-1, FEAT)
        self._chk(self._l.airfe_track_frame(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                            None if ref is None else ref.ctypes.data, 0 if ref is None else len(ref), feat.ctypes.data, cap, C.byref(n),
                                            idx.ctypes.data, sc.ctypes.data, cap, C.byref(nm)), "airfe_track_frame")
        return feat[:n.value], idx[:nm.value], sc[:nm.value]

    def promote_frame(self, right: np.ndarray):
        """The promotion of map_builder.cc:104-108 for the frame of the last track_frame call (airfe_promote_frame): Detect(right) + MatchingPoints(left, right)
        with the left rows still on the device -> (featR [n,259], idx [m,2] (left, right), score [m])."""
        right = np.asarray(right)
        if right.ndim != 2 or right.dtype != np.uint8 or right.size == 0:
            raise AirfeError("empty image")
        if right.strides[1] != 1 or right.strides[0] < right.shape[1]:
            right = np.ascontiguousarray(right)
        cap = self.np_rows
        feat, idx, sc = np.empty((cap, FEAT), np.float32), np.empty((cap, 2), np.int32), np.empty((cap,), np.float32)
        n, nm = C.c_int(0), C.c_int(0)
        self._chk(self._l.airfe_promote_frame(self._h, right.ctypes.data, right.shape[0], right.shape[1], right.strides[0], feat.ctypes.data, cap, C.byref(n),
                                              idx.ctypes.data, sc.ctypes.data, cap, C.byref(nm)), "airfe_promote_frame")
        return feat[:n.value], idx[:nm.value], sc[:nm.value]

    def adopt_reference(self):
        """`_last_keyframe_feature = frame` for a promoted frame: the last track_frame's rows become the reference, on the device (airfe_adopt_reference)."""
        self._chk(self._l.airfe_adopt_reference(self._h), "airfe_adopt_reference")

    def debug_plnet_stage0(self):
        """The on-device stage-0 line branch of the last detected image: dict in synth.plnet_stage0_lines' layout + jloc / joff."""
        n = 3 * 128 * 128
        out = dict(juncs_pred=np.empty((300, 2), np.float32), lines_pred=np.empty((n, 4), np.float32),
                   iskeep=np.empty((1, 3, 128, 128), np.float32), idx_junc_to_end_min=np.empty((1, 3, 128, 128), np.float32),
                   idx_junc_to_end_max=np.empty((1, 3, 128, 128), np.float32), loi_features=np.empty((1, 128, 128, 128), np.float32),
                   loi_features_thin=np.empty((1, 4, 128, 128), np.float32), loi_features_aux=np.empty((1, 4, 128, 128), np.float32),
                   jloc=np.empty((128, 128), np.float32), joff=np.empty((2, 128, 128), np.float32))
        self._chk(self._l.airfe_debug_plnet_stage0(self._h, *[out[k].ctypes.data for k in (
            "juncs_pred", "lines_pred", "iskeep", "idx_junc_to_end_min", "idx_junc_to_end_max", "loi_features",
            "loi_features_thin", "loi_features_aux", "jloc", "joff")]), "airfe_debug_plnet_stage0")
        return out

    def debug_plnet_j2l(self, fast: bool):
        """(iskeep, idx_min, idx_max) [3*128*128] of the last detected image: as the line path computes them (fast) or in full."""
        out = [np.zeros((3 * 128 * 128,), np.float32) for _ in range(3)]
        self._chk(self._l.airfe_debug_plnet_j2l(self._h, 1 if fast else 0, *(o.ctypes.data for o in out)), "airfe_debug_plnet_j2l")
        return tuple(out)

    def debug_plnet_s1(self, stage0, cap: int = 45056):
        st, keep = self._stage0(stage0)
        la = np.empty((cap, 4), np.float32)
        sc = np.empty((cap,), np.float32)
        m2 = C.c_int(0)
        self._chk(self._l.airfe_debug_plnet_s1(self._h, C.byref(st), la.ctypes.data, sc.ctypes.data, cap, C.byref(m2)),
                  "airfe_debug_plnet_s1")
        return la[:m2.value].copy(), sc[:m2.value].copy()

    def debug_plnet_s1_last(self, cap: int = 45056):
        """(lines_adjusted [m2, 4], scores_line [m2]) of image 0 of the last PLNet call, as the device path's stage-1 kernel left them"""
        la = np.empty((cap, 4), np.float32)
        sc = np.empty((cap,), np.float32)
        m2 = C.c_int(0)
        self._chk(self._l.airfe_debug_plnet_s1_last(self._h, la.ctypes.data, sc.ctypes.data, cap, C.byref(m2)), "airfe_debug_plnet_s1_last")
        return la[:m2.value].copy(), sc[:m2.value].copy()

    def match_superglue(self, f0: np.ndarray, f1: np.ndarray):
        """f0/f1: [n, 259] rows (score, normalised x, y, desc) -> (indices0, indices1, mscores0, mscores1)."""
        f0 = np.ascontiguousarray(f0, dtype=np.float32)
        f1 = np.ascontiguousarray(f1, dtype=np.float32)
        i0 = np.empty((f0.shape[0],), np.int32); i1 = np.empty((f1.shape[0],), np.int32)
        m0 = np.empty((f0.shape[0],), np.float64); m1 = np.empty((f1.shape[0],), np.float64)
        self._chk(self._l.airfe_match_superglue(self._h, f0.ctypes.data, f0.shape[0], f1.ctypes.data, f1.shape[0],
                                                i0.ctypes.data, i1.ctypes.data, m0.ctypes.data, m1.ctypes.data),
                  "airfe_match_superglue")
        return i0, i1, m0, m1

    def assign_points_to_lines(self, lines: np.ndarray, feat: np.ndarray):
        """AssignPointsToLines (src/line_processor.cc:68-120).  lines [L,4] float64, feat [N,259] float32 rows ->
        list of L dicts {point index: distance} (ascending index, like the reference's std::map<int,double>)."""
        lines = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 4)
        feat = np.ascontiguousarray(feat, dtype=np.float32).reshape(-1, 259)
        L, N = lines.shape[0], feat.shape[0]
        cap = max(L * N, 1)
        row_ptr = np.zeros((L + 1,), np.int32)
        idx = np.empty((cap,), np.int32); dist = np.empty((cap,), np.float64)
        total = C.c_int(0)
        self._chk(self._l.airfe_assign_points_to_lines(self._h, lines.ctypes.data, L, feat.ctypes.data, N, row_ptr.ctypes.data,
                                                       idx.ctypes.data, dist.ctypes.data, cap, C.byref(total)),
                  "airfe_assign_points_to_lines")
        return [dict(zip(idx[row_ptr[i]:row_ptr[i + 1]].tolist(), dist[row_ptr[i]:row_ptr[i + 1]].tolist())) for i in range(L)]

    @staticmethod
    def _relation_csr(relation):
        row_ptr = np.zeros((len(relation) + 1,), np.int32)
        for i, rel in enumerate(relation):
            row_ptr[i + 1] = row_ptr[i] + len(rel)
        idx = np.array([k for rel in relation for k in sorted(rel)], np.int32).reshape(-1)
        return row_ptr, np.ascontiguousarray(idx)

    def match_lines(self, points_on_line0, points_on_line1, point_matches, point_num0: int, point_num1: int):
        """MatchLines (src/line_processor.cc:122-172).  points_on_line{0,1}: the relations returned by assign_points_to_lines;
        point_matches: [(queryIdx, trainIdx), ...] -> list of len(points_on_line0) matched line indices of frame 1 (-1: none)."""
        rp0, pi0 = self._relation_csr(points_on_line0)
        rp1, pi1 = self._relation_csr(points_on_line1)
        m = np.ascontiguousarray(np.array([(int(a), int(b)) for a, b in point_matches], np.int32).reshape(-1, 2))
        out = np.full((max(len(points_on_line0), 1),), -1, np.int32)
        self._chk(self._l.airfe_match_lines(self._h, rp0.ctypes.data, pi0.ctypes.data if pi0.size else None, len(points_on_line0),
                                            int(point_num0), rp1.ctypes.data, pi1.ctypes.data if pi1.size else None,
                                            len(points_on_line1), int(point_num1), m.ctypes.data if m.size else None, m.shape[0],
                                            out.ctypes.data), "airfe_match_lines")
        return out[:len(points_on_line0)].tolist()

    def superglue_scores(self, f0: np.ndarray, f1: np.ndarray) -> np.ndarray:
        f0 = np.ascontiguousarray(f0, dtype=np.float32)
        f1 = np.ascontiguousarray(f1, dtype=np.float32)
        out = np.empty((f0.shape[0] + 1, f1.shape[0] + 1), dtype=np.float32)
        self._chk(self._l.airfe_debug_superglue_scores(self._h, f0.ctypes.data, f0.shape[0], f1.ctypes.data, f1.shape[0],
                                                       out.ctypes.data), "airfe_debug_superglue_scores")
        return out

    def debug_lg_filter(self, scores: np.ndarray):
        """filter_matches alone on a host score matrix [n0, n1] -> (idx [k,2], score [k])."""
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        cap = self.np_rows
        idx = np.empty((cap, 2), np.int32); sc = np.empty((cap,), np.float32)
        n = C.c_int(0)
        self._chk(self._l.airfe_debug_lg_filter(self._h, scores.ctypes.data, scores.shape[0], scores.shape[1], idx.ctypes.data,
                                                sc.ctypes.data, cap, C.byref(n)), "airfe_debug_lg_filter")
        return idx[:n.value].copy(), sc[:n.value].copy()

    def debug_sg_decode(self, z: np.ndarray):
        """SuperGlue decode alone on a host score matrix [n0+1, n1+1] -> (indices0, indices1, mscores0, mscores1)."""
        z = np.ascontiguousarray(z, dtype=np.float32)
        n0, n1 = z.shape[0] - 1, z.shape[1] - 1
        i0 = np.empty((n0,), np.int32); i1 = np.empty((n1,), np.int32)
        m0 = np.empty((n0,), np.float64); m1 = np.empty((n1,), np.float64)
        self._chk(self._l.airfe_debug_sg_decode(self._h, z.ctypes.data, n0, n1, i0.ctypes.data, i1.ctypes.data, m0.ctypes.data,
                                                m1.ctypes.data), "airfe_debug_sg_decode")
        return i0, i1, m0, m1

    def detector_maps(self, b: int = 1):
        heat = np.empty((b, 512, 512), np.float32)
        nms = np.empty((b, 512, 512), np.float32)
        desc = np.empty((b, 64, 64, 256), np.float32)
        self._chk(self._l.airfe_debug_detector_maps(self._h, b, heat.ctypes.data, nms.ctypes.data, desc.ctypes.data),
                  "airfe_debug_detector_maps")
        return heat, nms, desc

    def debug_preprocess(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        out = np.empty((512, 512), np.float32)
        self._chk(self._l.airfe_debug_preprocess(self._h, gray.ctypes.data, gray.shape[0], gray.shape[1], gray.strides[0],
                                                 out.ctypes.data), "airfe_debug_preprocess")
        return out

    def debug_conv3x3(self, x, w, b, pool=False):
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        bb, cin, hh, ww = x.shape
        cout = w.shape[0]
        ho, wo = (hh // 2, ww // 2) if pool else (hh, ww)
        y = np.empty((bb, cout, ho, wo), np.float32)
        self._chk(self._l.airfe_debug_conv3x3(self._h, x.ctypes.data, bb, cin, hh, ww, w.ctypes.data, b.ctypes.data, cout,
                                              int(pool), y.ctypes.data), "airfe_debug_conv3x3")
        return y

    def debug_gemm(self, x, w, b, relu=False):
        x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32); b = np.ascontiguousarray(b, np.float32)
        m, k = x.shape
        n = w.shape[0]
        y = np.empty((m, n), np.float32)
        self._chk(self._l.airfe_debug_gemm(self._h, x.ctypes.data, m, k, w.ctypes.data, b.ctypes.data, n, int(relu),
                                           y.ctypes.data), "airfe_debug_gemm")
        return y

    def debug_attention(self, q, k, v, lens, cross=False):
        """The matcher's flash attention alone (include/airfe_debug.h): q, k, v [S, H, n, 64] fp32 (scale and log2 e already inside q / k), lens [S] -> out [S, n, H * 64]."""
        q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32); v = np.ascontiguousarray(v, np.float32)
        lens = np.ascontiguousarray(lens, np.int32)
        s, h, n, d = q.shape
        assert d == 64 and k.shape == q.shape and v.shape == q.shape and lens.shape == (s,)
        out = np.empty((s, n, h * 64), np.float32)
        self._chk(self._l.airfe_debug_attention(self._h, q.ctypes.data, k.ctypes.data, v.ctypes.data, lens.ctypes.data, s, h, n, int(cross), out.ctypes.data),
                  "airfe_debug_attention")
        return out

    # ---------------------------------------------------------------- device-resident batch (torch plumbing)
    def detect_batch_dev(self, gray_t, feat_t, n_t, stream=None):
        b, h, w = gray_t.shape
        self._chk(self._l.airfe_detect_points_batch_dev(self._h, gray_t.data_ptr(), b, h, w, gray_t.stride(1),
                                                        gray_t.stride(0), feat_t.data_ptr(), feat_t.shape[1],
                                                        n_t.data_ptr(), self._stream(stream)), "airfe_detect_points_batch_dev")

    def match_lightglue_batch_dev(self, f0_t, n0_t, f1_t, n1_t, idx_t, score_t, nm_t, stream=None):
        self._chk(self._l.airfe_match_lightglue_batch_dev(self._h, f0_t.data_ptr(), n0_t.data_ptr(), f1_t.data_ptr(),
                                                          n1_t.data_ptr(), f0_t.shape[0], f0_t.shape[1], idx_t.data_ptr(),
                                                          score_t.data_ptr(), idx_t.shape[1], nm_t.data_ptr(), self._stream(stream)),
                  "airfe_match_lightglue_batch_dev")

    def match_superglue_batch_dev(self, f0_t, n0_t, f1_t, n1_t, idx0_t, idx1_t, ms0_t, ms1_t, stream=None):
        self._chk(self._l.airfe_match_superglue_batch_dev(self._h, f0_t.data_ptr(), n0_t.data_ptr(), f1_t.data_ptr(), n1_t.data_ptr(),
                                                          f0_t.shape[0], f0_t.shape[1], idx0_t.data_ptr(), idx1_t.data_ptr(),
                                                          ms0_t.data_ptr(), ms1_t.data_ptr(), self._stream(stream)),
                  "airfe_match_superglue_batch_dev")

    def stereo_batch_dev(self, left_t, right_t, featL, featR, nL, nR, idx_t, score_t, nm_t, stream=None):
        b, h, w = left_t.shape
        self._chk(self._l.airfe_stereo_batch_dev(self._h, left_t.data_ptr(), right_t.data_ptr(), b, h, w, left_t.stride(1),
                                                 left_t.stride(0), featL.data_ptr(), featR.data_ptr(), featL.shape[1],
                                                 nL.data_ptr(), nR.data_ptr(), idx_t.data_ptr(), score_t.data_ptr(),
                                                 idx_t.shape[1], nm_t.data_ptr(), self._stream(stream)), "airfe_stereo_batch_dev")

    def detect_plnet_batch_dev(self, gray_t, feat_t, n_t, lines_t, nlines_t, junc_t=None, njunc_t=None, found_t=None, stream=None):
        """PLNet over a device batch: gray [B][h][w] u8, feat [B][cap][259] f32, n [B] i32, lines [B][capL][4] f64, nlines [B] i32,
        junc [J][capJ][259] f32 + njunc [J] i32 for the first J images (None: no junctions), found [B + J] i32 (None: not reported)."""
        b, h, w = gray_t.shape
        nj = 0 if junc_t is None else junc_t.shape[0]
        self._chk(self._l.airfe_detect_plnet_batch_dev(
            self._h, gray_t.data_ptr(), b, h, w, gray_t.stride(1), gray_t.stride(0), feat_t.data_ptr(), feat_t.shape[1], n_t.data_ptr(),
            lines_t.data_ptr(), lines_t.shape[1], nlines_t.data_ptr(), junc_t.data_ptr() if nj else None, junc_t.shape[1] if nj else 0,
            njunc_t.data_ptr() if nj else None, nj, found_t.data_ptr() if found_t is not None else None, self._stream(stream)),
            "airfe_detect_plnet_batch_dev")

    def stereo_plnet_batch_dev(self, left_t, right_t, featL, featR, nL, nR, lines_t, nlines_t, juncL, njuncL, idx_t, score_t, nm_t,
                               found_t=None, stream=None):
        """B stereo pairs with the PLNet detector: lines [2B][capL][4] / nlines [2B] (left images first), junctions of the left images
        juncL [B][capJ][259] / njuncL [B], LightGlue matches as stereo_batch_dev; found [3B] i32 (None: not reported)."""
        b, h, w = left_t.shape
        self._chk(self._l.airfe_stereo_plnet_batch_dev(
            self._h, left_t.data_ptr(), right_t.data_ptr(), b, h, w, left_t.stride(1), left_t.stride(0), featL.data_ptr(), featR.data_ptr(),
            featL.shape[1], nL.data_ptr(), nR.data_ptr(), lines_t.data_ptr(), lines_t.shape[1], nlines_t.data_ptr(), juncL.data_ptr(),
            juncL.shape[1], njuncL.data_ptr(), found_t.data_ptr() if found_t is not None else None, idx_t.data_ptr(), score_t.data_ptr(),
            idx_t.shape[1], nm_t.data_ptr(), self._stream(stream)), "airfe_stereo_plnet_batch_dev")

    def assign_points_to_lines_batch_dev(self, lines_t, nlines_t, feat_t, n_t, row_ptr_t, pt_idx_t, pt_dist_t, total_t=None, stream=None):
        """AssignPointsToLines (src/line_processor.cc:68-120) over B device-resident frames: lines [B][capL][4] f64 + nlines [B], feat [B][cap][259] + n [B]
        -> CSR per frame row_ptr [B][capL + 1] i32, pt_idx [B][capE] i32, pt_dist [B][capE] f64, total [B] i32 (entries found; > capE = overflow)."""
        b, capl = lines_t.shape[0], lines_t.shape[1]
        self._chk(self._l.airfe_assign_points_to_lines_batch_dev(
            self._h, lines_t.data_ptr(), nlines_t.data_ptr(), capl, feat_t.data_ptr(), n_t.data_ptr(), feat_t.shape[1], b, row_ptr_t.data_ptr(),
            pt_idx_t.data_ptr(), pt_dist_t.data_ptr(), pt_idx_t.shape[1], total_t.data_ptr() if total_t is not None else None, self._stream(stream)),
            "airfe_assign_points_to_lines_batch_dev")

    def match_lines_batch_dev(self, row_ptr0_t, pt_idx0_t, nlines0_t, n0_t, row_ptr1_t, pt_idx1_t, nlines1_t, n1_t, matches_t, nmatch_t, line_matches_t,
                              stereo_filter=None, feat0_t=None, feat1_t=None, stream=None):
        """MatchLines (src/line_processor.cc:122-180) over B frame pairs from the relations above and the matcher's lists matches [B][mcap][2] / nmatch [B]
        -> line_matches [B][capL] i32.  stereo_filter = (min_x_diff, max_x_diff, max_y_diff): the band of Frame::AddRightFeatures (src/frame.cc:147-160)."""
        b, capl = line_matches_t.shape
        f3 = (C.c_double * 3)(*stereo_filter) if stereo_filter is not None else None
        self._chk(self._l.airfe_match_lines_batch_dev(
            self._h, row_ptr0_t.data_ptr(), pt_idx0_t.data_ptr(), nlines0_t.data_ptr(), n0_t.data_ptr(), row_ptr1_t.data_ptr(), pt_idx1_t.data_ptr(),
            nlines1_t.data_ptr(), n1_t.data_ptr(), capl, pt_idx0_t.shape[1], matches_t.data_ptr(), nmatch_t.data_ptr(), matches_t.shape[1], b,
            C.cast(f3, C.c_void_p) if f3 is not None else None, feat0_t.data_ptr() if feat0_t is not None else None,
            feat1_t.data_ptr() if feat1_t is not None else None, feat0_t.shape[1] if feat0_t is not None else 0, line_matches_t.data_ptr(),
            self._stream(stream)), "airfe_match_lines_batch_dev")

    def rectify_batch_dev(self, side, raw_t, rect_t, stream=None):
        """cv::remap of Camera::UndistortImage (camera.cc:161-182) over B device-resident raw images [B][h][w] u8 -> rect_t (same shape)."""
        b, h, w = raw_t.shape
        self._chk(self._l.airfe_rectify_batch_dev(self._h, side, raw_t.data_ptr(), b, h, w, raw_t.stride(1), raw_t.stride(0), rect_t.data_ptr(),
                                                  rect_t.stride(1), rect_t.stride(0), self._stream(stream)), "airfe_rectify_batch_dev")

    def bow_transform_dev(self, feat_t, word_t, weight_t, stream=None):
        """TemplatedVocabulary::transform per feature row of feat_t [..., 259] (device) -> word_t u32 / weight_t f32, one per row."""
        n = feat_t.numel() // FEAT
        self._chk(self._l.airfe_bow_transform_dev(self._h, feat_t.data_ptr(), n, word_t.data_ptr(), weight_t.data_ptr(), self._stream(stream)),
                  "airfe_bow_transform_dev")

    def copy_rows_plan(self, jobs):
        """jobs: [(src tensor / pointer, dst tensor / pointer, count tensor (int32, one element) / pointer / None, row_bytes, cap rows)] -> a reusable plan for
        copy_rows_dev (the five host arrays of airfe_copy_rows_dev, built once: the buffers of a pipeline do not move)"""
        ptr = lambda x: 0 if x is None else (int(x) if isinstance(x, int) else x.data_ptr())        # (dst None: a plan for pack_rows_dev only)
        n = len(jobs)
        return dict(n=n, src=np.array([ptr(j[0]) for j in jobs], np.uint64), dst=np.array([ptr(j[1]) for j in jobs], np.uint64),
                    cnt=np.array([ptr(j[2]) for j in jobs], np.uint64), rb=np.array([j[3] for j in jobs], np.uint32), cap=np.array([j[4] for j in jobs], np.uint32),
                    keep=jobs)

    def copy_rows_dev(self, plan, stream=None):
        """airfe_copy_rows_dev: the valid rows of every job of `plan` in one launch on `stream` (asynchronous)"""
        self._chk(self._l.airfe_copy_rows_dev(self._h, plan["n"], plan["src"].ctypes.data, plan["dst"].ctypes.data, plan["cnt"].ctypes.data, plan["rb"].ctypes.data,
                                              plan["cap"].ctypes.data, self._stream(stream)), "airfe_copy_rows_dev")

    def pack_rows_dev(self, plan, packed_t, offsets_t, stream=None):
        """airfe_pack_rows_dev: the valid rows of every job of `plan` (its dst column is ignored) back to back into `packed_t` (device uint8), job j at
        offsets_t[j], offsets_t[n] = bytes used (device int64 [n + 1]); two launches on `stream` (asynchronous)"""
        assert offsets_t.numel() >= plan["n"] + 1
        self._chk(self._l.airfe_pack_rows_dev(self._h, plan["n"], plan["src"].ctypes.data, plan["cnt"].ctypes.data, plan["rb"].ctypes.data, plan["cap"].ctypes.data,
                                              packed_t.data_ptr(), offsets_t.data_ptr(), self._stream(stream)), "airfe_pack_rows_dev")

    def _stream(self, stream):
        # the ctx runs on its own non-blocking stream: order it after whatever torch queued on ITS streams
        if stream is None:
            import torch
            torch.cuda.synchronize()
        return stream

    def profile(self, on=True, stages=None):
        """Per-stage hipEvent timers: on=False off, on=True every stage, stages=[names] only those (cheaper)."""
        mask = 0
        if stages is not None:
            names = [self._l.airfe_profile_stage_name(i).decode() for i in range(self._l.airfe_profile_stages())]
            for s in stages:
                mask |= 1 << names.index(s)
        elif on:
            mask = -1
        self._chk(self._l.airfe_profile_enable(self._h, mask), "airfe_profile_enable")

    def profile_read(self):
        n = self._l.airfe_profile_stages()
        ms = (C.c_double * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)(); la = (C.c_int * n)()
        self._chk(self._l.airfe_profile_read(self._h, ms, fl, by, la), "airfe_profile_read")
        return {self._l.airfe_profile_stage_name(i).decode(): dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=la[i])
                for i in range(n)}

    def sync(self):
        self._chk(self._l.airfe_sync(self._h), "airfe_sync")

    def debug_fail_next_launch(self, stage):
        """stage: a profiling stage name (profile_read's keys) or None to disarm — see include/airfe_debug.h"""
        names = [self._l.airfe_profile_stage_name(i).decode() for i in range(self._l.airfe_profile_stages())]
        self._chk(self._l.airfe_debug_fail_next_launch(self._h, -1 if stage is None else names.index(stage)), "airfe_debug_fail_next_launch")

    # ---- fault hunting: checksums of the matcher's state behind every launch (airfe_debug_trace*)
    def trace(self, on=True):
        self._chk(self._l.airfe_debug_trace(self._h, 1 if on else 0), "airfe_debug_trace")

    def trace_stop(self, slot=-1):
        self._chk(self._l.airfe_debug_trace_stop(self._h, slot), "airfe_debug_trace_stop")

    def trace_buffer(self, slot, dtype):
        """The whole buffer slot `slot` of the last matcher call covers, as a flat numpy array of `dtype`."""
        _, _, units, uw = self.trace_slots()[slot]
        out = np.zeros(units * uw * 4 // np.dtype(dtype).itemsize, dtype)
        self._chk(self._l.airfe_debug_trace_buffer(self._h, slot, out.ctypes.data, out.nbytes), "airfe_debug_trace_buffer")
        return out

    def trace_slots(self):
        """[(name, first unit, units, words per unit)] of the last matcher call."""
        out = []
        for i in range(self._l.airfe_debug_trace_slots(self._h)):
            name = C.create_string_buffer(64)
            off, units, uw = C.c_uint(), C.c_uint(), C.c_uint()
            self._chk(self._l.airfe_debug_trace_slot(self._h, i, name, 64, C.byref(off), C.byref(units), C.byref(uw)), "airfe_debug_trace_slot")
            out.append((name.value.decode(), off.value, units.value, uw.value))
        return out

    def trace_read(self, table=False, stream=None):
        """(digests [slots] u64, unit table u64 or None) of the last matcher call; synchronises the stream."""
        slots = self.trace_slots()
        dig = np.zeros(len(slots), np.uint64)
        tab = np.zeros(slots[-1][1] + slots[-1][2], np.uint64) if (table and slots) else None
        self._chk(self._l.airfe_debug_trace_read(self._h, stream, dig.ctypes.data, tab.ctypes.data if tab is not None else None), "airfe_debug_trace_read")
        return dig, tab


# ------------------------------------------------------------------------------------ reference-shaped façade
class FeatureDetector:
    """Mirror of FeatureDetector (include/feature_detector.h:8-31, src/feature_detector.cc)."""

    def __init__(self, ctx: Context):
        self._ctx = ctx

    def Detect(self, image: np.ndarray):
        """Detect(cv::Mat&, Eigen::Matrix<float,259,Dynamic>&) -> (ok, features [259, N])."""
        try:
            f = self._ctx.detect_points(image)
        except (AirfeError, TypeError):
            print("Failed when extracting point features !")     # feature_detector.cc:46-48
            return False, np.zeros((FEAT, 0), np.float32, order="F")
        return True, np.asfortranarray(f.T)

    def DetectLines(self, image: np.ndarray, stage0, lines: list, junction_detection: bool = False):
        """Detect(image, features, lines[, junctions]) (feature_detector.cc:52-69) -> (ok, features, junctions).
        `lines` is APPENDED to, never cleared — exactly like the reference (plnet.cpp:544)."""
        try:
            f, l, j = self._ctx.detect_plnet(image, stage0, want_junctions=junction_detection)
        except (AirfeError, TypeError):
            print("Failed when extracting point features !")
            return False, np.zeros((FEAT, 0), np.float32, order="F"), np.zeros((FEAT, 0), np.float32, order="F")
        lines.extend(tuple(r) for r in l)
        return True, np.asfortranarray(f.T), np.asfortranarray(j.T)

    def DetectKeyframe(self, left: np.ndarray, right: np.ndarray, left_lines: list, right_lines: list):
        """Detect(image_left, image_right, left_features, right_features, left_lines, right_lines, junctions) (feature_detector.cc:97-108) as ONE
        device pass over both images (Context.stereo_keyframe without the match) -> (ok, left_features, right_features, junctions)."""
        try:
            k = self._ctx.stereo_keyframe(left, right, match=False)
        except (AirfeError, TypeError):
            print("Failed when extracting point features !")
            z = np.zeros((FEAT, 0), np.float32, order="F")
            return False, z, z.copy(), z.copy()
        left_lines.extend(tuple(r) for r in k["linesL"])
        right_lines.extend(tuple(r) for r in k["linesR"])
        return True, np.asfortranarray(k["featL"].T), np.asfortranarray(k["featR"].T), np.asfortranarray(k["juncL"].T)

    def DetectStereo(self, left: np.ndarray, right: np.ndarray):
        okl, fl = self.Detect(left)
        okr, fr = self.Detect(right)
        ok = okl & okr                                            # feature_detector.cc:74-80
        if not ok:
            print("Failed when extracting point features !")
        return ok, fl, fr


class PointMatcher:
    """Mirror of PointMatcher (include/point_matcher.h:8-24, src/point_matcher.cc).  The F-matrix RANSAC of
    MatchingPoints (cv::findFundamentalMat, :95-104) stays in the reference's own code and is not part of this path."""

    def __init__(self, ctx: Context, image_width: int, image_height: int, matcher: int = 0):
        self._ctx = ctx
        self.image_width, self.image_height, self.matcher = image_width, image_height, matcher

    @staticmethod
    def NormalizeKeypoints(features: np.ndarray, width: int, height: int, scale: float) -> np.ndarray:
        """point_matcher.cc:39-48 on a [259, N] matrix."""
        out = np.array(features, dtype=np.float32, order="F", copy=True)
        l_inv = np.float32(1.0 / max(width, height) * float(np.float32(scale)))
        out[1] = (features[1] - np.float32(width // 2)) * l_inv
        out[2] = (features[2] - np.float32(height // 2)) * l_inv
        return out

    def MatchingPoints(self, features0: np.ndarray, features1: np.ndarray):
        """-> (count, matches) with matches = list of (queryIdx, trainIdx, distance) ≙ cv::DMatch."""
        if features0.shape[1] < 1 or features1.shape[1] < 1:
            return 0, []                                          # point_matcher.cc:53-55
        scale = 0.7 if self.matcher else 0.5
        n0 = self.NormalizeKeypoints(features0, self.image_width, self.image_height, scale)
        n1 = self.NormalizeKeypoints(features1, self.image_width, self.image_height, scale)
        if self.matcher == 0:
            idx, sc = self._ctx.match_lightglue(np.ascontiguousarray(n0[1:].T), np.ascontiguousarray(n1[1:].T))
            matches = [(int(i), int(j), float(np.float32(1.0) - s)) for (i, j), s in zip(idx, sc)]
        else:
            i0, i1, m0, m1 = self._ctx.match_superglue(np.ascontiguousarray(n0.T), np.ascontiguousarray(n1.T))
            matches = []
            for i in range(len(i0)):                              # point_matcher.cc:82-91
                if 0 <= i0[i] < len(i1) and i1[i0[i]] == i:
                    matches.append((i, int(i0[i]), float(np.float32(1.0 - (m0[i] + m1[i0[i]]) / 2.0))))
        return len(matches), matches
