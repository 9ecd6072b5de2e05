"""CPU checks of the sequence workload's host logic (airslam_amd/seq.py; BASELINE configs[3]): the product-side driver of MapBuilder::ExtractFeatureThread's
loop (src/map_builder.cc:83-141) against the oracle's independent restatement of it (oracle/ref_seq.py), with the device entries replaced by the oracle
networks; and the K = 8 gather cadence over two gloo ranks."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from airslam_amd import seq, synth, weights
from conftest import GOLDEN

W, H, K = 752, 480, 128


class OracleContext:
    """The five host entries SequenceFrontEnd calls, answered by the CPU oracle (test infrastructure: the product has no CPU path)."""

    def __init__(self, chain):
        self.c = chain
        self.ref = None
        self.last = None
        self.uploads = 0

    def _ref(self, ref_feat):
        if ref_feat is not None:
            self.ref = np.array(ref_feat)
            self.uploads += 1
        assert self.ref is not None, "no reference rows were ever given"
        return self.ref

    def stereo_keyframe(self, left, right, track=False, ref_feat=None):
        from oracle import ref_chain
        L = ref_chain.plnet_infer(self.c.pl, self.c.s1, left, want_junctions=True, top_k=K)
        R = ref_chain.plnet_infer(self.c.pl, self.c.s1, right, want_junctions=False, top_k=K)
        idx, sc, _ = self.c.match(L["features"], R["features"])
        out = dict(featL=L["features"], featR=R["features"], linesL=L["lines"], linesR=R["lines"], juncL=L["junctions"], idx=idx, score=sc)
        if track:
            out["track_idx"], out["track_score"], _ = self.c.match(self._ref(ref_feat), L["features"])
        return out

    def track_frame(self, gray, ref_feat=None):
        f = self.c.superpoint(gray)
        idx, sc, _ = self.c.match(self._ref(ref_feat), f)
        self.last = f
        return f, idx, sc

    def promote_frame(self, right):
        fr = self.c.superpoint(right)
        idx, sc, _ = self.c.match(self.last, fr)
        return fr, idx, sc

    def adopt_reference(self):
        self.ref = self.last


def test_driver_and_oracle_read_the_loop_the_same_way():
    """28 frames (scenes of 9): the product-side driver with oracle-backed entries takes the branches the oracle chain takes on its own and returns the same
    arrays — two independent restatements of map_builder.cc:83-141, AddKeyframeCheck and AddRightFeatures' count."""
    from oracle import ref_seq
    pol = dict(tracking_point_rate=0.2, min_num_match=12, max_num_match=25, min_init_stereo_feature=20)
    pl, sp = weights.synthetic_plnet_s0(1234), weights.synthetic_superpoint(1234)
    s1, lg = weights.load_pack(os.path.join(GOLDEN, "plnet_s1.airfe")), weights.synthetic_lightglue(1234, n_layers=2)
    own = ref_seq.Chain(pl, sp, s1, lg, W, H, K, policy=dict(pol))
    own.match_layers = 2
    backing = ref_seq.Chain(pl, sp, s1, lg, W, H, K, policy=dict(pol))
    for c in (own, backing):          # 2-layer LightGlue keeps the CPU suite short
        c.match = (lambda c_: lambda f0, f1: _match2(c_, f0, f1))(c)
    kf, nf = OracleContext(backing), OracleContext(backing)
    fe = seq.SequenceFrontEnd(kf, nf, seq.KeyframeConfig(image_width=W, image_height=H, **pol))
    types, promos = [], 0
    for t, (left, right) in enumerate(synth.stereo_sequence(28, H, W, 10, scene_len=9)):
        r = fe.step(left, right)
        o = own.step(left, right)
        assert (r.candidate, r.promoted, r.frame_type, r.dropped, r.enough_match, r.good_stereo_point) == \
               (o["candidate"], o["promoted"], o["frame_type"], o["dropped"], o["enough_match"], o["good_stereo_point"]), f"frame {t}"
        assert fe.state.insert_next == own.insert_next and fe.state.init == own.init
        np.testing.assert_array_equal(r.features_left, o["features_left"])
        for a, b in ((r.matches_idx, o.get("matches_idx")), (r.stereo_idx, o.get("stereo_idx")), (r.features_right, o.get("features_right")),
                     (r.lines_left, o.get("lines_left"))):
            assert (a is None) == (b is None)
            if a is not None:
                np.testing.assert_array_equal(a, b)
        types.append(r.frame_type)
        promos += r.promoted
    assert types[0] == seq.INIT and types.count(seq.KEYFRAME) >= 2 and types.count(seq.NORMAL) >= 12 and promos >= 1, (types, promos)
    # the reference rows cross "PCIe" once per new keyframe and context, not once per frame
    assert nf.uploads <= types.count(seq.KEYFRAME) + 1 and kf.uploads <= types.count(seq.KEYFRAME) + 1


def _match2(chain, f0, f1):
    from oracle import ref_nets, ref_post
    if f0.shape[0] < 1 or f1.shape[0] < 1:
        return np.zeros((0, 2), np.int32), np.zeros((0,), np.float32), None
    a = ref_post.normalize_keypoints(f0, chain.W, chain.H, 0.5)
    b = ref_post.normalize_keypoints(f1, chain.W, chain.H, 0.5)
    s = ref_nets.lightglue_forward(chain.lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:], n_layers=2)
    idx, sc = ref_post.filter_matches(s, 0.1)
    return np.asarray(idx, np.int32).reshape(-1, 2), np.asarray(sc, np.float32), s


def test_keyframe_check_restatements_agree_on_threshold_cases():
    from oracle import ref_seq
    rng = np.random.default_rng(3)
    cfg = seq.KeyframeConfig()
    for trial in range(200):
        n0, n1 = int(rng.integers(60, 400)), int(rng.integers(60, 400))
        a, b = np.zeros((n0, 259), np.float32), np.zeros((n1, 259), np.float32)
        a[:, 1:3] = rng.uniform(0, 700, (n0, 2)); b[:, 1:3] = rng.uniform(0, 700, (n1, 2))
        m = int(rng.choice([0, 29, 30, 79, 80, 81, int(0.65 * min(n0, n1)), int(0.65 * min(n0, n1)) + 1, min(n0, n1)]))
        m = min(m, n0, n1)
        idx = np.stack([rng.permutation(n0)[:m], rng.permutation(n1)[:m]], 1).astype(np.int32)
        shift = rng.choice([0.5, 40.0, 59.0, 61.0, 90.0])
        b[idx[:, 1], 1:3] = a[idx[:, 0], 1:3] + np.float32(shift) / np.sqrt(np.float32(2))
        assert seq.add_keyframe_check(cfg, a, b, idx) == ref_seq.add_keyframe_check(a, b, idx), (trial, n0, n1, m, shift)
        assert seq.good_stereo_points(cfg, a, b, idx) == ref_seq.add_right_features_count(a, b, idx)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from airslam_amd import dist as adist
    adist.init_from_env("gloo")
    S, cap, KF = 3, 16, 8
    g = seq.MatchGatherer(KF, S, cap, "cpu")
    got = []
    for t in range(20):                                   # 20 frames: gathers behind frames 7 and 15, frames 16-19 stay pending
        n = torch.tensor([(rank * 100 + t * 3 + s) % (cap + 1) for s in range(S)], dtype=torch.int32)
        idx = torch.zeros((S, cap, 2), dtype=torch.int32); sc = torch.zeros((S, cap))
        for s in range(S):
            idx[s, :n[s], 0] = torch.arange(int(n[s]), dtype=torch.int32) + 1000 * rank + t
            idx[s, :n[s], 1] = s
            sc[s, :n[s]] = 0.5 + 0.01 * t
        h = g.add(idx, sc, n)
        assert (h is not None) == (t % KF == KF - 1)
        if h is not None:
            got.append(h.result())
    q.put((rank, g.gathers, [None if o is None else tuple(x.numpy() for x in o) for o in got]))
    torch.distributed.destroy_process_group()


def test_gather_every_8_frames_gloo_world2():
    """BASELINE configs[3]'s exchange: the temporal match lists of K = 8 frames x S sequences go to rank 0 in ONE collective, every 8th frame."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(2))}
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == 2 and res[1][1] == 2 and all(o is None for o in res[1][2])
    S, cap, KF = 3, 16, 8
    for gi, (idx, sc, n) in enumerate(res[0][2]):
        assert idx.shape == (2 * KF * S, cap, 2) and n.shape == (2 * KF * S,)
        for rank in range(2):
            for k in range(KF):
                t = gi * KF + k
                for s in range(S):
                    row = rank * KF * S + k * S + s
                    want = (rank * 100 + t * 3 + s) % (cap + 1)
                    assert n[row] == want
                    assert (idx[row, :want, 0] == np.arange(want) + 1000 * rank + t).all() and (idx[row, :want, 1] == s).all()
                    assert np.allclose(sc[row, :want], 0.5 + 0.01 * t)


def test_native_policy_functions_equal_the_python_driver_and_the_oracle():
    """include/airfe_seq.h: airfe_seq_add_keyframe_check / airfe_seq_good_stereo_points (what the C++ lock-step driver decides with; pure host functions, no GPU)
    against airslam_amd.seq's Python forms and the oracle's independent restatement (oracle/ref_seq.py) of src/map_builder.cc:429-466 and src/frame.cc:141-172 —
    on random frames whose match counts and ratios straddle every threshold of the policy."""
    import ctypes as C
    from airslam_amd import _lib
    from oracle import ref_seq
    lib = _lib.lib()
    rng = np.random.default_rng(5)
    seen_akc, seen_good = set(), 0
    for trial in range(400):
        cfg = seq.KeyframeConfig(min_num_match=int(rng.integers(5, 40)), max_num_match=int(rng.integers(40, 90)), tracking_point_rate=float(rng.choice([0.2, 0.35, 0.65])),
                                 tracking_parallax_rate=float(rng.choice([0.005, 0.02, 0.1])))
        pol = _lib.SeqPolicy(cfg.min_init_stereo_feature, cfg.min_num_match, cfg.max_num_match, cfg.tracking_point_rate, cfg.tracking_parallax_rate, cfg.min_x_diff,
                             cfg.max_x_diff, cfg.max_y_diff, cfg.image_width, cfg.image_height)
        n0, n1 = int(rng.integers(1, 200)), int(rng.integers(1, 200))
        f0 = rng.uniform(0, 1, (n0, 259)).astype(np.float32); f1 = rng.uniform(0, 1, (n1, 259)).astype(np.float32)
        f0[:, 1] = rng.uniform(0, W, n0); f0[:, 2] = rng.uniform(0, H, n0)
        m = int(rng.integers(0, min(n0, n1) + 1))
        idx = np.stack([rng.permutation(n0)[:m], rng.permutation(n1)[:m]], 1).astype(np.int32)
        spread = float(rng.choice([0.5, 3.0, 30.0, 150.0]))             # matched keypoints move by about this many pixels: parallax below / above the threshold
        f1[:, 1] = rng.uniform(0, W, n1); f1[:, 2] = rng.uniform(0, H, n1)
        f1[idx[:, 1], 1] = f0[idx[:, 0], 1] - rng.uniform(0.2, 1.0, m).astype(np.float32) * np.float32(spread)
        f1[idx[:, 1], 2] = f0[idx[:, 0], 2] + rng.normal(0, 2.5, m).astype(np.float32)
        got = lib.airfe_seq_add_keyframe_check(C.byref(pol), f0.ctypes.data, n0, f1.ctypes.data, n1, idx.ctypes.data, m)
        want = seq.add_keyframe_check(cfg, f0, f1, idx)
        orc = ref_seq.add_keyframe_check(f0, f1, idx, cfg.min_num_match, cfg.max_num_match, cfg.tracking_point_rate, cfg.tracking_parallax_rate, W, H)
        assert got == want == orc, (trial, got, want, orc, m, n0, n1)
        seen_akc.add(got)
        g = lib.airfe_seq_good_stereo_points(C.byref(pol), f0.ctypes.data, f1.ctypes.data, idx.ctypes.data, m)
        assert g == seq.good_stereo_points(cfg, f0, f1, idx) == ref_seq.add_right_features_count(f0, f1, idx, cfg.min_x_diff, cfg.max_x_diff, cfg.max_y_diff), trial
        seen_good += g > 0 and g < m
    assert seen_akc == {0, 1, 2} and seen_good > 50
    # the band's edges: |dx| exactly on min_x_diff / max_x_diff, dy exactly on max_y_diff (the comparisons are >, <, <=)
    cfg = seq.KeyframeConfig()
    pol = _lib.SeqPolicy(cfg.min_init_stereo_feature, cfg.min_num_match, cfg.max_num_match, cfg.tracking_point_rate, cfg.tracking_parallax_rate, 1.0, 200.0, 5.0, W, H)
    fl = np.zeros((6, 259), np.float32); fr = np.zeros((6, 259), np.float32)
    fl[:, 1] = 300; fl[:, 2] = 100
    fr[:, 1] = [299, 298.5, 100, 99.5, 301.5, 250]; fr[:, 2] = [100, 105, 100, 95, 100, 105.5]
    idx = np.stack([np.arange(6), np.arange(6)], 1).astype(np.int32)
    assert lib.airfe_seq_good_stereo_points(C.byref(pol), fl.ctypes.data, fr.ctypes.data, idx.ctypes.data, 6) == seq.good_stereo_points(cfg, fl, fr, idx) == 1      # only row 1 (|dx| 1.5, dy 5): rows 0, 2, 3 sit ON the band, row 4 has a negative parallax, row 5 dy 5.5
    assert lib.airfe_seq_add_keyframe_check(None, None, 0, None, 0, None, 0) == -1
