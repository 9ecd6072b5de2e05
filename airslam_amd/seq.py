"""The feature thread's per-frame loop over the airfe entry points: BASELINE.json configs[3] (one stereo SEQUENCE per rank, frames depend on each other
through the last keyframe) as a workload.

What the reference runs per frame is `MapBuilder::ExtractFeatureThread`, src/map_builder.cc:55-147, with every shipped configuration's
`use_superpoint: 1` (configs/visual_odometry/*.yaml:2; FeatureDetector then holds a SuperPoint AND a PLNet, src/feature_detector.cc:7-34):

    keyframe candidate (`!_init || _insert_next_keyframe`, :83-92):
        Detect(left, right, features, lines, junctions)           -> PLNet::infer on both images            (feature_detector.cc:97-108)
        MatchingPoints(left, right, stereo_matches)                                                          (:86)
    normal frame (:93-97):
        Detect(left, features)                                     -> SuperPoint::infer, points only         (feature_detector.cc:36-41)
    every frame once initialised (:99-121):
        MatchingPoints(features_last_keyframe, left, matches)      the temporal match                        (:100-101)
        AddKeyframeCheck(last_keyframe, frame, matches)            0 = make this frame a keyframe, 1 = the next one, 2 = neither   (:102, :429-466)
        a NORMAL frame with result 0 is promoted: Detect(right) + MatchingPoints(left, right)                (:104-108)
    `_last_keyframe_feature = frame` for every frame that is not a normal one (:139-141)

Three drivers of the same loop, with identical per-frame outputs (tests/test_gpu_seq.py):
  * `SequenceFrontEnd`  — ONE sequence through the batch-1 host entries, one call per branch: airfe_stereo_keyframe[_tracked] (PLNet x2 + stereo match
    [+ temporal match in the same LightGlue forward]), airfe_track_frame (SuperPoint + temporal match, the reference rows resident on the device),
    airfe_promote_frame (+ airfe_adopt_reference).  The regime AirSLAM itself runs in: latency per frame.
  * `BatchedSequences`  — S independent sequences in lock-step through the device-resident *_batch_dev entries: per time-step the sequences are grouped by
    branch (keyframe candidates -> one PLNet stereo batch, the rest -> one SuperPoint batch; all temporal matches -> one LightGlue batch; promotions -> one
    more detector + matcher batch), one host synchronisation per decision point.  The regime of a server replaying many sequences: frames/s.
  * `NativeSequences`   — the same lock-step schedule driven from C++ (include/airfe_seq.h, csrc/airfe_seq.hip: the reference's caller is C++ too): the host side
    of a time-step is two C calls, every gather / scatter / result copy of the step is one copy-job launch, and only the valid rows cross PCIe.
    `NativePipeline` runs two such drivers (two groups of sequences, each with its own contexts) half a step apart, so that the device works on one group
    while the host decides for the other.

The keyframe POLICY (AddKeyframeCheck, Frame::AddRightFeatures' stereo count) is the caller's, not the path's: it is restated here (file:line on every
function) only so that the loop can run without the SLAM back end.  Not restated: the F-matrix RANSAC behind MatchingPoints(..., true)
(src/point_matcher.cc:95-104, cv::findFundamentalMat) — the temporal matches are the matcher's own list; IMU branches (UseIMU() is false in the VO configs).

Two contexts, like the reference's two detector objects: `kf` = PLNet (+ stage 1) + LightGlue, `nf` = SuperPoint + LightGlue.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import api

NORMAL, KEYFRAME, INIT = 0, 1, 2          # FrameType, include/map_builder.h:40-44


@dataclass
class KeyframeConfig:
    """configs/visual_odometry/vo_euroc.yaml:16-22 (include/read_configs.h:128-147) + the camera's stereo band (src/camera.cc:50-51, :25: bf / depth
    thresholds and max_y_diff; the numbers of an EuRoC-like rig)."""
    min_init_stereo_feature: int = 90
    min_num_match: int = 30
    max_num_match: int = 80
    tracking_point_rate: float = 0.65
    tracking_parallax_rate: float = 0.1
    min_x_diff: float = 1.0
    max_x_diff: float = 200.0
    max_y_diff: float = 5.0
    image_width: int = 752
    image_height: int = 480


def add_keyframe_check(cfg: KeyframeConfig, ref_feat: np.ndarray, cur_feat: np.ndarray, idx: np.ndarray) -> int:
    """MapBuilder::AddKeyframeCheck (src/map_builder.cc:429-466), UseIMU() == false.  ref_feat / cur_feat [n, 259] rows, idx [m, 2] = (queryIdx, trainIdx)."""
    m = int(len(idx))
    if m < cfg.min_num_match:
        return 0                                                                                   # :431
    thr = np.float32(cfg.tracking_point_rate)
    if (np.float32(m) / np.float32(len(ref_feat)) < thr or np.float32(m) / np.float32(len(cur_feat)) < thr or m < cfg.max_num_match):
        return 1                                                                                   # :443-445
    par = ref_feat[idx[:, 0], 1:3].astype(np.float32) - cur_feat[idx[:, 1], 1:3].astype(np.float32)     # Matrix2Xf parallax, :447-457
    g = par.T.astype(np.float32) @ par.astype(np.float32)                                          # (parallax * parallax.transpose()).sum(): ALL entries of the 2 x 2 product
    average_parallax = float(np.float32(g.sum())) / m                                              # :458
    image_size = float(cfg.image_height * cfg.image_width)
    if average_parallax > image_size * cfg.tracking_parallax_rate * cfg.tracking_parallax_rate:    # :461
        return 1
    return 2


def good_stereo_points(cfg: KeyframeConfig, feat_left: np.ndarray, feat_right: np.ndarray, idx: np.ndarray) -> int:
    """The count Frame::AddRightFeatures returns (src/frame.cc:141-172): stereo matches inside the camera's band whose signed parallax is inside it too."""
    if len(idx) == 0:
        return 0
    xl, xr = feat_left[idx[:, 0], 1].astype(np.float32), feat_right[idx[:, 1], 1].astype(np.float32)
    yl, yr = feat_left[idx[:, 0], 2].astype(np.float32), feat_right[idx[:, 1], 2].astype(np.float32)
    dx = np.abs(xl - xr).astype(np.float64)                  # std::abs(float - float) -> double, :150-151
    dy = np.abs(yl - yr).astype(np.float64)
    keep = (dx > cfg.min_x_diff) & (dx < cfg.max_x_diff) & (dy <= cfg.max_y_diff)                  # :153
    parallax = (xl - xr).astype(np.float64)                  # :165
    return int((keep & (parallax < cfg.max_x_diff) & (parallax > cfg.min_x_diff)).sum())           # :167-171


@dataclass
class FrameResult:
    """What one iteration of the loop hands to the tracking thread (TrackingData, map_builder.cc:132-137) + what it decided on the way."""
    frame_type: int = NORMAL
    candidate: bool = False                   # took the keyframe branch (:83)
    promoted: bool = False                    # took the promotion branch (:104-108)
    dropped: bool = False                     # "Not enough stereo points to initialize!" (:122-125): the frame is not handed on
    enough_match: int = -1                    # AddKeyframeCheck's result (-1: not initialised yet)
    good_stereo_point: int = 0
    features_left: np.ndarray = field(default_factory=lambda: np.zeros((0, 259), np.float32))
    features_right: Optional[np.ndarray] = None
    lines_left: Optional[np.ndarray] = None
    lines_right: Optional[np.ndarray] = None
    junctions: Optional[np.ndarray] = None
    stereo_idx: Optional[np.ndarray] = None
    stereo_score: Optional[np.ndarray] = None
    matches_idx: Optional[np.ndarray] = None  # temporal: (last keyframe index, this frame's index)
    matches_score: Optional[np.ndarray] = None

    ARRAYS = ("features_left", "features_right", "lines_left", "lines_right", "junctions", "stereo_idx", "stereo_score", "matches_idx", "matches_score")

    def same_as(self, o: "FrameResult") -> List[str]:
        """names of the fields that differ (byte comparison of every array)"""
        bad = [k for k in ("frame_type", "candidate", "promoted", "dropped", "enough_match", "good_stereo_point") if getattr(self, k) != getattr(o, k)]
        for k in self.ARRAYS:
            a, b = getattr(self, k), getattr(o, k)
            if (a is None) != (b is None) or (a is not None and (a.shape != b.shape or a.tobytes() != b.tobytes())):
                bad.append(k)
        return bad


class _LoopState:
    """`_init`, `_insert_next_keyframe`, `_last_keyframe_feature` of MapBuilder (include/map_builder.h) for one sequence"""

    def __init__(self):
        self.init = False
        self.insert_next = False
        self.ref: Optional[np.ndarray] = None            # the last keyframe's features, host copy (AddKeyframeCheck reads their x, y)

    def candidate(self) -> bool:
        return (not self.init) or self.insert_next                                                  # :83

    def decide(self, cfg: KeyframeConfig, r: FrameResult, promote) -> None:
        """map_builder.cc:99-141 for one frame whose detections and temporal matches are in `r`; `promote()` runs the promotion branch on demand and fills
        r.features_right / r.stereo_* / r.good_stereo_point."""
        frame_type = (KEYFRAME if self.init else INIT) if r.candidate else NORMAL                  # :88, :96
        if self.init:
            r.enough_match = add_keyframe_check(cfg, self.ref, r.features_left, r.matches_idx)     # :102
            if r.enough_match == 0:
                if frame_type == NORMAL:
                    promote()                                                                       # :105-109
                    r.promoted = True
                if r.good_stereo_point < 10:                                                        # :111-117
                    self.insert_next = True
                    frame_type = NORMAL
                else:
                    frame_type = KEYFRAME
                    self.insert_next = False
            else:
                self.insert_next = (r.enough_match == 1) and (frame_type == NORMAL)                 # :119
        else:
            if r.good_stereo_point < cfg.min_init_stereo_feature:                                   # :122-125
                r.dropped = True
                r.frame_type = frame_type
                return
            self.init = True                                                                        # :127-128
        r.frame_type = frame_type
        if frame_type != NORMAL:
            self.ref = r.features_left                                                              # :139-141


class SequenceFrontEnd:
    """ONE sequence, one frame per call, batch-1 host entries (the module docstring has the map)."""

    def __init__(self, kf: api.Context, nf: api.Context, cfg: Optional[KeyframeConfig] = None):
        self.kf, self.nf, self.cfg = kf, nf, cfg or KeyframeConfig()
        self.state = _LoopState()
        self._ref_on = {id(kf): False, id(nf): False}     # which context holds the current reference rows on its device block
        self._nf_has_frame = False

    def _ref_arg(self, ctx):
        """the last keyframe's rows for a call on `ctx`: None when they are resident there already"""
        if self._ref_on[id(ctx)]:
            return None
        self._ref_on[id(ctx)] = True
        return self.state.ref

    def step(self, left: np.ndarray, right: np.ndarray) -> FrameResult:
        st, cfg = self.state, self.cfg
        r = FrameResult(candidate=st.candidate())
        if r.candidate:
            if st.init:        # keyframe + its temporal match in one call (map_builder.cc:85-86 and :100-101)
                k = self.kf.stereo_keyframe(left, right, track=True, ref_feat=self._ref_arg(self.kf))
                r.matches_idx, r.matches_score = k["track_idx"], k["track_score"]
            else:
                k = self.kf.stereo_keyframe(left, right)
            r.features_left, r.features_right = k["featL"], k["featR"]
            r.lines_left, r.lines_right, r.junctions = k["linesL"], k["linesR"], k["juncL"]
            r.stereo_idx, r.stereo_score = k["idx"], k["score"]
            r.good_stereo_point = good_stereo_points(cfg, r.features_left, r.features_right, r.stereo_idx)
            self._nf_has_frame = False
        else:
            r.features_left, r.matches_idx, r.matches_score = self.nf.track_frame(left, ref_feat=self._ref_arg(self.nf))
            self._nf_has_frame = True

        def promote():
            r.features_right, r.stereo_idx, r.stereo_score = self.nf.promote_frame(right)
            r.good_stereo_point = good_stereo_points(cfg, r.features_left, r.features_right, r.stereo_idx)
        before = st.ref
        st.decide(cfg, r, promote)
        if st.ref is not before:                          # a new last keyframe
            self._ref_on = {id(self.kf): False, id(self.nf): False}
            if self._nf_has_frame:                        # a promoted frame: its rows are on nf's device already (airfe_adopt_reference)
                self.nf.adopt_reference()
                self._ref_on[id(self.nf)] = True
        return r


class BatchedSequences:
    """S sequences in lock-step through the device-resident batch entries.  `step(L, R)` takes the S left / right images of one time-step as device
    tensors [S, h, w] uint8 and returns S FrameResults whose arrays are the bytes SequenceFrontEnd returns for each sequence on its own."""

    def __init__(self, kf: api.Context, nf: api.Context, S: int, cfg: Optional[KeyframeConfig] = None, cap_lines: int = 1024, cap_junc: int = 1024,
                 device=None, copy_results: bool = True):
        """copy_results = False: the arrays of a step's FrameResults are VIEWS of pinned staging memory, valid until the end of the NEXT step (two staging sets
        alternate) — a consumer that hands them on within a frame time saves S x 0.4 MB of host copies per step (1.6 ms at S = 16)."""
        import torch
        self.t = torch
        self.kf, self.nf, self.S, self.cfg = kf, nf, S, cfg or KeyframeConfig()
        self.states = [_LoopState() for _ in range(S)]
        K = nf.max_keypoints
        assert kf.max_keypoints == K, "both contexts must be created with the same max_keypoints"
        self.K, self.CL, self.CJ = K, cap_lines, cap_junc
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
        i32 = torch.int32
        # per-sequence state on the device: the last keyframe's rows and this frame's left rows
        self.ref, self.ref_n = z(S, K, 259), z(S, dt=i32)
        self.cur, self.cur_n = z(S, K, 259), z(S, dt=i32)
        # branch staging (contiguous sub-batches)
        self.kl, self.kr, self.knl, self.knr = z(S, K, 259), z(S, K, 259), z(S, dt=i32), z(S, dt=i32)
        self.klines, self.knlines = z(2 * S, cap_lines, 4, dt=torch.float64), z(2 * S, dt=i32)
        self.kjunc, self.knjunc, self.kfound = z(S, cap_junc, 259), z(S, dt=i32), z(3 * S, dt=i32)
        self.kidx, self.ksc, self.knm = z(S, K, 2, dt=i32), z(S, K), z(S, dt=i32)
        self.nfeat, self.nn = z(S, K, 259), z(S, dt=i32)
        self.tref, self.tref_n, self.tcur, self.tcur_n = z(S, K, 259), z(S, dt=i32), z(S, K, 259), z(S, dt=i32)
        self.tidx, self.tsc, self.tnm = z(S, K, 2, dt=i32), z(S, K), z(S, dt=i32)
        self.pr, self.pnr = z(S, K, 259), z(S, dt=i32)
        self.pidx, self.psc, self.pnm = z(S, K, 2, dt=i32), z(S, K), z(S, dt=i32)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream_k = torch.cuda.Stream(device=dev)      # the keyframe candidates' PLNet batch runs here, beside the normal frames' SuperPoint batch on `stream`
        self.syncs = 0
        self.t_queue = self.t_wait = self.t_host = 0.0      # where a time-step's wall time goes: queueing device work, waiting for it, the host side of the loop
        # pinned host twins of everything the host side of the loop reads (two sets in turn): queued as asynchronous copies, read after ONE stream synchronisation
        names = ("cur", "cur_n", "kr", "knr", "klines", "knlines", "kjunc", "knjunc", "kidx", "ksc", "knm", "kfound", "tidx", "tsc", "tnm", "pr", "pnr", "pidx", "psc", "pnm")
        self._pins = [{k: torch.empty(getattr(self, k).shape, dtype=getattr(self, k).dtype).pin_memory() for k in names} for _ in range(2)]
        self._pin = self._pins[0]
        self._flip = 0
        self.copy_results = copy_results
        self._own = (lambda a: a.copy()) if copy_results else (lambda a: a)
        # the index lists of a step (which sequences take which branch) go up in ONE small copy from a pinned block
        self._idx_h = torch.empty((6 * S,), dtype=torch.int64).pin_memory()
        self._idx_d = torch.empty((6 * S,), dtype=torch.int64, device=dev)

    def _upload_idx(self, lists, base=0):
        """index lists -> device int64 slices (one asynchronous H2D on `stream`; the pinned block is free again: every step ends behind a synchronisation)"""
        off, out = base, []
        for ids in lists:
            n = len(ids)
            if n:
                self._idx_h[off:off + n] = self.t.as_tensor(ids, dtype=self.t.int64)
            out.append((off, n))
            off += n
        if off > base:
            self._idx_d[base:off].copy_(self._idx_h[base:off], non_blocking=True)
        return [self._idx_d[o:o + n] for o, n in out]

    def _home(self, name, n):
        """queue rows [0, n) of device tensor `name` into its pinned twin (asynchronous); the numpy view is valid after the stream synchronisation"""
        self._pin[name][:n].copy_(getattr(self, name)[:n], non_blocking=True)
        return self._pin[name][:n].numpy()

    def step(self, L, R) -> List[FrameResult]:
        t, S, cfg, sh = self.t, self.S, self.cfg, self.stream.cuda_stream
        out = [FrameResult(candidate=s.candidate()) for s in self.states]
        kset = [i for i in range(S) if out[i].candidate]
        nset = [i for i in range(S) if not out[i].candidate]
        tset = [i for i in range(S) if self.states[i].init]
        import time as _time
        t_a = _time.perf_counter()
        self._flip ^= 1
        self._pin = self._pins[self._flip]
        with t.cuda.stream(self.stream):
            ks, ns, ts = self._upload_idx([kset, nset, tset])
        if kset:              # keyframe candidates: PLNet on both images + the stereo match, one batch (map_builder.cc:85-86) — on its own stream, beside the
            with t.cuda.stream(self.stream_k):             # normal frames' batch (the two contexts share nothing; they join before the temporal match)
                self.stream_k.wait_stream(self.stream)     # (the previous step's reference update ran on `stream`)
                nk = len(kset)
                Lk, Rk = L.index_select(0, ks), R.index_select(0, ks)
                self.kf.stereo_plnet_batch_dev(Lk, Rk, self.kl[:nk], self.kr[:nk], self.knl[:nk], self.knr[:nk], self.klines[:2 * nk], self.knlines[:2 * nk],
                                               self.kjunc[:nk], self.knjunc[:nk], self.kidx[:nk], self.ksc[:nk], self.knm[:nk], self.kfound[:3 * nk],
                                               stream=self.stream_k.cuda_stream)
                self.cur.index_copy_(0, ks, self.kl[:nk]); self.cur_n.index_copy_(0, ks, self.knl[:nk])
        with t.cuda.stream(self.stream):
            if nset:          # normal frames: SuperPoint on the left image, one batch (:94)
                nn_ = len(nset)
                self.nf.detect_batch_dev(L.index_select(0, ns), self.nfeat[:nn_], self.nn[:nn_], stream=sh)
                self.cur.index_copy_(0, ns, self.nfeat[:nn_]); self.cur_n.index_copy_(0, ns, self.nn[:nn_])
            if kset:
                self.stream.wait_stream(self.stream_k)
            if tset:          # the temporal match of every initialised sequence, one LightGlue batch (:100-101)
                nt = len(tset)
                self.tref[:nt] = self.ref.index_select(0, ts); self.tref_n[:nt] = self.ref_n.index_select(0, ts)
                self.tcur[:nt] = self.cur.index_select(0, ts); self.tcur_n[:nt] = self.cur_n.index_select(0, ts)
                self.nf.match_lightglue_batch_dev(self.tref[:nt], self.tref_n[:nt], self.tcur[:nt], self.tcur_n[:nt], self.tidx[:nt], self.tsc[:nt], self.tnm[:nt],
                                                  stream=sh)
            # everything the host side of the loop reads, in one wait
            h_cur, h_cur_n = self._home("cur", S), self._home("cur_n", S)
            if kset:
                nk = len(kset)
                h_k = [self._home(k, n) for k, n in (("kr", nk), ("knr", nk), ("klines", 2 * nk), ("knlines", 2 * nk), ("kjunc", nk), ("knjunc", nk), ("kidx", nk),
                                                     ("ksc", nk), ("knm", nk), ("kfound", 3 * nk))]
            if tset:
                nt = len(tset)
                h_t = [self._home(k, nt) for k in ("tidx", "tsc", "tnm")]
        t_b = _time.perf_counter()
        self.stream.synchronize()
        t_c = _time.perf_counter()
        self.syncs += 1
        # the asynchronous entries report through their contexts (an fp16 overflow of a detector: never keypoints of a poisoned score map): ask both
        self.kf.sync(); self.nf.sync()
        for i in range(S):
            out[i].features_left = self._own(h_cur[i, :h_cur_n[i]])
        if kset:
            kr, knr, kln, knl, kj, knj, kidx, ksc, knm, kfound = h_k
            nk = len(kset)
            if (kfound[:2 * nk] > self.CL).any() or (kfound[2 * nk:] > self.CJ).any():
                raise api.AirfeError("BatchedSequences: line / junction capacity overflow (cap_lines, cap_junc)")
            for j, i in enumerate(kset):
                r = out[i]
                r.features_right = self._own(kr[j, :knr[j]])
                r.lines_left, r.lines_right = self._own(kln[j, :knl[j]]), self._own(kln[nk + j, :knl[nk + j]])
                r.junctions = self._own(kj[j, :knj[j]])
                m = int(knm[j]) if (len(r.features_left) and len(r.features_right)) else 0           # point_matcher.cc:53-55
                r.stereo_idx, r.stereo_score = self._own(kidx[j, :m]), self._own(ksc[j, :m])
                r.good_stereo_point = good_stereo_points(cfg, r.features_left, r.features_right, r.stereo_idx)
        if tset:
            tidx, tsc, tnm = h_t
            for j, i in enumerate(tset):
                m = int(tnm[j]) if (len(self.states[i].ref) and len(out[i].features_left)) else 0
                out[i].matches_idx, out[i].matches_score = self._own(tidx[j, :m]), self._own(tsc[j, :m])
        # decisions; promotions are collected first (they need a second device pass), then replayed
        pset = []
        for i in tset:
            if not out[i].candidate and add_keyframe_check(cfg, self.states[i].ref, out[i].features_left, out[i].matches_idx) == 0:
                pset.append(i)
        promo = {}
        if pset:              # promotions: SuperPoint on the right image + the stereo match, one batch each (:104-108)
            npz = len(pset)
            with t.cuda.stream(self.stream):
                (ps,) = self._upload_idx([pset], base=3 * S)
                self.nf.detect_batch_dev(R.index_select(0, ps), self.pr[:npz], self.pnr[:npz], stream=sh)
                self.tcur[:npz] = self.cur.index_select(0, ps); self.tcur_n[:npz] = self.cur_n.index_select(0, ps)
                self.nf.match_lightglue_batch_dev(self.tcur[:npz], self.tcur_n[:npz], self.pr[:npz], self.pnr[:npz], self.pidx[:npz], self.psc[:npz], self.pnm[:npz],
                                                  stream=sh)
                h_p = [self._home(k, npz) for k in ("pr", "pnr", "pidx", "psc", "pnm")]
            t_p0 = _time.perf_counter()
            self.stream.synchronize()
            self.t_wait += _time.perf_counter() - t_p0
            t_c += _time.perf_counter() - t_p0             # (the promotion's wait is not host time)
            self.syncs += 1
            self.nf.sync()
            pr, pnr, pidx, psc, pnm = h_p
            for j, i in enumerate(pset):
                fr = self._own(pr[j, :pnr[j]])
                m = int(pnm[j]) if (len(out[i].features_left) and len(fr)) else 0
                promo[i] = (fr, self._own(pidx[j, :m]), self._own(psc[j, :m]))
        newkf = []
        for i in range(S):
            r, stt = out[i], self.states[i]

            def promote(i=i, r=r):
                r.features_right, r.stereo_idx, r.stereo_score = promo[i]
                r.good_stereo_point = good_stereo_points(cfg, r.features_left, r.features_right, r.stereo_idx)
            before = stt.ref
            stt.decide(cfg, r, promote)
            if stt.ref is not before:
                newkf.append(i)
                if not self.copy_results:
                    stt.ref = np.array(stt.ref)     # (the reference outlives the staging set its rows came back in)
        if newkf:             # `_last_keyframe_feature = frame`: on the device, the new keyframes' rows become the reference rows
            with t.cuda.stream(self.stream):
                (ks,) = self._upload_idx([newkf], base=4 * S)
                self.ref.index_copy_(0, ks, self.cur.index_select(0, ks)); self.ref_n.index_copy_(0, ks, self.cur_n.index_select(0, ks))
        self.t_queue += t_b - t_a
        self.t_wait += t_c - t_b
        self.t_host += _time.perf_counter() - t_c
        return out


class NativeSequences:
    """S sequences in lock-step through the C++ driver (include/airfe_seq.h).  `step(L, R)` = BatchedSequences.step with the same results (views of the
    driver's pinned staging memory, valid until the end of the next step unless copy_results); `begin` / `end_raw` are the two halves for pipelined use,
    `raw` the ctypes array of airfe_seq_frame records of the last step (what a C++ caller would read)."""

    def __init__(self, kf: api.Context, nf: api.Context, S: int, cfg: Optional[KeyframeConfig] = None, cap_lines: int = 1024, cap_junc: int = 1024,
                 device=None, copy_results: bool = True, temporal_buffers: bool = False):
        import ctypes as C
        from . import _lib
        self._C, self._l = C, _lib.lib()
        self.kf, self.nf, self.S, self.cfg = kf, nf, S, cfg or KeyframeConfig()
        self.K = nf.max_keypoints
        c = self.cfg
        pol = _lib.SeqPolicy(c.min_init_stereo_feature, c.min_num_match, c.max_num_match, c.tracking_point_rate, c.tracking_parallax_rate, c.min_x_diff, c.max_x_diff,
                             c.max_y_diff, c.image_width, c.image_height)
        self.tidx = self.tsc = self.tnm = None
        ptrs = (None, None, None)
        if temporal_buffers:         # the temporal match lists stay on the device in caller-owned tensors (MatchGatherer forwards them)
            import torch
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            self.tidx = torch.zeros((S, self.K, 2), dtype=torch.int32, device=dev)
            self.tsc = torch.zeros((S, self.K), dtype=torch.float32, device=dev)
            self.tnm = torch.zeros((S,), dtype=torch.int32, device=dev)
            ptrs = (self.tidx.data_ptr(), self.tsc.data_ptr(), self.tnm.data_ptr())
        h = C.c_void_p()
        if self._l.airfe_seq_create(kf._h, nf._h, S, C.byref(pol), cap_lines, cap_junc, ptrs[0], ptrs[1], ptrs[2], C.byref(h)):
            raise api.AirfeError("airfe_seq_create: " + (self._l.airfe_seq_last_error(None) or b"").decode())
        self._h = h
        self.raw = (_lib.SeqFrame * S)()
        self.copy_results = copy_results
        self.stream_ptr = self._l.airfe_seq_stream(self._h)
        self._ext_stream = None

    @property
    def stream(self):
        """the driver's stream as a torch stream (for ordering a forwarder of the temporal buffers behind it)"""
        if self._ext_stream is None:
            import torch
            self._ext_stream = torch.cuda.ExternalStream(self.stream_ptr)
        return self._ext_stream

    def close(self):
        if getattr(self, "_h", None):
            self._l.airfe_seq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise api.AirfeError(f"{what}: {(self._l.airfe_seq_last_error(self._h) or b'').decode()}")

    def begin(self, L, R):
        """L, R: [S, h, w] uint8 device tensors (contiguous rows)"""
        assert L.shape[0] == self.S and R.shape == L.shape and L.stride(2) == 1 and R.stride() == L.stride()
        self._chk(self._l.airfe_seq_begin(self._h, L.data_ptr(), R.data_ptr(), L.shape[1], L.shape[2], L.stride(1), L.stride(0)), "airfe_seq_begin")

    def end_raw(self):
        self._chk(self._l.airfe_seq_end(self._h, self.raw), "airfe_seq_end")
        return self.raw

    def step_raw(self, L, R):
        assert L.shape[0] == self.S and R.shape == L.shape and L.stride(2) == 1 and R.stride() == L.stride()
        self._chk(self._l.airfe_seq_step(self._h, L.data_ptr(), R.data_ptr(), L.shape[1], L.shape[2], L.stride(1), L.stride(0), self.raw), "airfe_seq_step")
        return self.raw

    def results(self) -> List[FrameResult]:
        """the last step's records as FrameResults"""
        C = self._C
        own = (lambda a: a.copy()) if self.copy_results else (lambda a: a)

        def arr(ptr, n, cols, ct, dt):
            if n < 0 or not ptr:
                return None
            if n == 0:
                return np.zeros((0, cols) if cols else (0,), dt)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n * max(cols, 1),)).view(dt)
            return own(a.reshape(n, cols) if cols else a)
        out = []
        for r in self.raw:
            fr = FrameResult(frame_type=r.frame_type, candidate=bool(r.candidate), promoted=bool(r.promoted), dropped=bool(r.dropped), enough_match=r.enough_match,
                             good_stereo_point=r.good_stereo_point)
            fr.features_left = arr(r.features_left, r.n_left, 259, C.c_float, np.float32)
            fr.features_right = arr(r.features_right, r.n_right, 259, C.c_float, np.float32)
            fr.lines_left = arr(r.lines_left, r.n_lines_left, 4, C.c_double, np.float64)
            fr.lines_right = arr(r.lines_right, r.n_lines_right, 4, C.c_double, np.float64)
            fr.junctions = arr(r.junctions, r.n_junctions, 259, C.c_float, np.float32)
            fr.stereo_idx = arr(r.stereo_idx, r.n_stereo, 2, C.c_int32, np.int32)
            fr.stereo_score = arr(r.stereo_score, r.n_stereo, 0, C.c_float, np.float32)
            fr.matches_idx = arr(r.matches_idx, r.n_matches, 2, C.c_int32, np.int32)
            fr.matches_score = arr(r.matches_score, r.n_matches, 0, C.c_float, np.float32)
            out.append(fr)
        return out

    def step(self, L, R) -> List[FrameResult]:
        self.step_raw(L, R)
        return self.results()

    def counts(self):
        """the integer fields of the last step's records as one [S, 13] array (schedule statistics without building FrameResults)"""
        a = np.frombuffer(self.raw, dtype=np.uint8).reshape(self.S, -1)
        return a[:, :52].copy().view(np.int32)

    def wall_split(self):
        """-> dict(queue_s, wait_s, host_s, host_syncs, steps) since the last call"""
        C = self._C
        q, w, h, n, st = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_int()
        self._chk(self._l.airfe_seq_wall_split(self._h, C.byref(q), C.byref(w), C.byref(h), C.byref(n), C.byref(st)), "airfe_seq_wall_split")
        return dict(queue_s=q.value, wait_s=w.value, host_s=h.value, host_syncs=n.value, steps=st.value)


COUNT_FIELDS = ("frame_type", "candidate", "promoted", "dropped", "enough_match", "good_stereo_point", "n_left", "n_right", "n_lines_left", "n_lines_right",
                "n_junctions", "n_stereo", "n_matches")


class NativePipeline:
    """G groups of sequences (one NativeSequences each, with its own pair of contexts), half a step apart: begin(A) begin(B) | end(A) begin(A) end(B) begin(B) | ...
    — while the host takes group A's decisions and queues its next time-step, the device works on group B's.  `step(L, R)` takes the images of ALL sequences of a
    time-step ([S_total, h, w], group g owns rows [g * S, (g + 1) * S)) and returns the records of the time-step BEFORE (one step of latency: the pipeline's price);
    `flush()` returns the last ones."""

    def __init__(self, groups: List[NativeSequences]):
        self.g = groups
        self.S = groups[0].S
        assert all(x.S == self.S for x in groups)
        self._primed = False

    def _slices(self, L, R):
        S = self.S
        return [(L[i * S:(i + 1) * S], R[i * S:(i + 1) * S]) for i in range(len(self.g))]

    def step(self, L, R):
        sl = self._slices(L, R)
        if not self._primed:
            for x, (l, r) in zip(self.g, sl):
                x.begin(l, r)
            self._primed = True
            return None
        out = []
        for x, (l, r) in zip(self.g, sl):
            out.append(x.end_raw())
            self.on_group_done(x)
            x.begin(l, r)
        return out

    def on_group_done(self, x):
        """hook: called after a group's end, before its next begin (the temporal buffers of that group are complete and not yet overwritten)"""

    def flush(self):
        if not self._primed:
            return None
        out = []
        for x in self.g:
            out.append(x.end_raw())
            self.on_group_done(x)
        self._primed = False
        return out


class MatchGatherer:
    """SURVEY.md 8(e) / BASELINE configs[3]: every K frames the ranks' temporal match lists go to rank 0 in ONE collective (airslam_amd.dist.gather_matches:
    a padded [K * S][cap * 3 + 1] int32 buffer per rank), issued on a SIDE stream behind an event so that the next frames' kernels do not wait for it.
    `add(idx, score, n)` takes one time-step's [S, cap, 2] / [S, cap] / [S] tensors (device or CPU); the K-th call starts the gather and returns a handle whose
    `.result()` is rank 0's (idx, score, n) — [world * K * S, ...] in (rank, frame, sequence) order — or None on the other ranks."""

    def __init__(self, K: int, S: int, cap: int, device, dst: int = 0, buffers: int = 1):
        """buffers: sets of (idx, score, n) filled in turn, each with the event of the gather that last read it — with one set and K = 1 every add() waits for the
        previous collective before it refills the set; with two the compute stream only ever waits for the collective before the last one."""
        import torch
        self.t, self.K, self.S, self.cap, self.dst, self.dev = torch, K, S, cap, dst, device
        self.sets = [dict(idx=torch.zeros((K * S, cap, 2), dtype=torch.int32, device=device), score=torch.zeros((K * S, cap), dtype=torch.float32, device=device),
                          n=torch.zeros((K * S,), dtype=torch.int32, device=device), last=None) for _ in range(max(1, buffers))]
        self.cur = 0
        self.fill = 0
        self.gathers = 0
        self.cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.cuda else None

    @property
    def idx(self):
        return self.sets[self.cur]["idx"]

    @property
    def score(self):
        return self.sets[self.cur]["score"]

    @property
    def n(self):
        return self.sets[self.cur]["n"]

    def add(self, idx, score, n, stream=None):
        t, S, k = self.t, self.S, self.fill
        b = self.sets[self.cur]
        ctxm = t.cuda.stream(stream) if (self.cuda and stream is not None) else _Null()
        with ctxm:
            if k == 0 and b["last"] is not None and self.cuda:      # the set is about to be refilled: the gather that last read it must be done
                (stream or t.cuda.current_stream(self.dev)).wait_event(b["last"])
            rows = slice(k * S, k * S + idx.shape[0])
            b["idx"][rows] = idx; b["score"][rows] = score; b["n"][rows] = n
            if idx.shape[0] < S:
                b["n"][k * S + idx.shape[0]:(k + 1) * S] = 0
        self.fill += 1
        if self.fill < self.K:
            return None
        self.fill = 0
        return self._gather(stream)

    def _gather(self, stream):
        from . import dist as adist
        t = self.t
        self.gathers += 1
        b = self.sets[self.cur]
        self.cur = (self.cur + 1) % len(self.sets)
        if not self.cuda:
            out = adist.gather_matches(b["idx"], b["score"], b["n"], dst=self.dst)
            return _Done(out)
        ev = t.cuda.Event()
        ev.record(stream or t.cuda.current_stream(self.dev))
        with t.cuda.stream(self.side):
            self.side.wait_event(ev)
            out = adist.gather_matches(b["idx"], b["score"], b["n"], dst=self.dst)
            b["last"] = t.cuda.Event()
            b["last"].record(self.side)
        return _Done(out, self.side)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Done:
    def __init__(self, out, stream=None):
        self._out, self._stream = out, stream

    def result(self):
        if self._stream is not None:
            self._stream.synchronize()
        return self._out
