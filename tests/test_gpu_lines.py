"""SURVEY.md 8(f) rank 2 on the GPU: airfe_assign_points_to_lines (C ABI) vs the statement-by-statement oracle of
AssignPointsToLines (src/line_processor.cc:68-120).  Double arithmetic without contraction on both sides: the gate is
exact equality of the relation (same point sets per line, same order, bit-identical distances)."""
import numpy as np
import pytest

from gpu_common import context, diag
from oracle import ref_post

pytestmark = pytest.mark.gpu


def _feat(xy):
    f = np.zeros((len(xy), 259), np.float32)
    f[:, 0] = 0.5
    f[:, 1:3] = np.asarray(xy, np.float32)
    return f


def _same(dev, ref):
    assert len(dev) == len(ref)
    for i, (d, r) in enumerate(zip(dev, ref)):
        assert list(d.keys()) == list(r.keys()), f"line {i}: point sets differ"
        dv, rv = np.array(list(d.values()), np.float64), np.array(list(r.values()), np.float64)
        bad = ~((dv == rv) | (np.isnan(dv) & np.isnan(rv)))
        assert not bad.any(), f"line {i}: distances differ at points {np.array(list(r.keys()))[bad][:4]}: {dv[bad][:4]!r} vs {rv[bad][:4]!r}"


@pytest.mark.parametrize("seed,L,N", [(0, 150, 400), (1, 1, 1), (2, 37, 1024), (3, 600, 65), (4, 5, 0), (5, 0, 10)])
def test_assign_points_to_lines_exact(seed, L, N):
    ctx, _, _ = context("sp")
    rng = np.random.default_rng(seed)
    lines = rng.uniform(0, 752, size=(L, 4))
    lines[:, [1, 3]] *= 480.0 / 752.0
    pts = rng.uniform(0, 752, size=(N, 2))
    pts[:, 1] *= 480.0 / 752.0
    if L > 3 and N > 8:
        # plant points exactly on / near segments, the 3-px rim, endpoints, and degenerate lines
        lines[1, 2:] = lines[1, :2]
        t = rng.uniform(0, 1, size=6)[:, None]
        pts[:6] = lines[0, :2] * (1 - t) + lines[0, 2:] * t
        pts[6] = lines[2, :2] + np.array([3.0, 0.0])
        pts[7] = lines[1, :2] + np.array([1.0, -1.0])
    feat = _feat(pts)
    dev = ctx.assign_points_to_lines(lines, feat)
    ref = ref_post.assign_points_to_lines(lines, feat)
    diag(f"lines_assign_{seed}", lines=L, points=N, pairs=int(sum(len(r) for r in ref)))
    _same(dev, ref)


def test_assign_on_detector_output():
    """Same call on the path's own outputs: PLNet-style line list over detected keypoints (frame.cc:125)."""
    ctx, _, _ = context("sp")
    from airslam_amd import api, synth
    left, _ = synth.stereo_pair(480, 752, 3)
    ok, f = api.FeatureDetector(ctx).Detect(left)
    assert ok and f.shape[1] > 50
    feat = np.ascontiguousarray(f.T)
    rng = np.random.default_rng(9)
    ends = feat[rng.integers(0, feat.shape[0], size=(120, 2)), 1:3].astype(np.float64)      # lines joining keypoints
    lines = ends.reshape(120, 4)
    dev = ctx.assign_points_to_lines(lines, feat)
    ref = ref_post.assign_points_to_lines(lines, feat)
    _same(dev, ref)
    assert sum(len(r) for r in ref) >= 240       # at least the two endpoints of every line


@pytest.mark.parametrize("seed,L0,L1,N0,N1,M", [(0, 150, 140, 400, 400, 300), (1, 1, 1, 5, 5, 5), (2, 37, 300, 1024, 900, 700),
                                                (3, 600, 65, 400, 380, 33), (4, 5, 7, 30, 30, 0), (5, 40, 0, 50, 50, 20)])
def test_match_lines_vs_oracle(seed, L0, L1, N0, N1, M):
    """airfe_match_lines == MatchLines (line_processor.cc:122-172): integer votes, first-maximum ties, float score gate: exact."""
    from test_oracle_post import _random_line_frames
    ctx, _, _ = context("sp")
    rel0, rel1, matches = _random_line_frames(100 + seed, L0, L1, N0, N1, M)
    dev = ctx.match_lines(rel0, rel1, matches, N0, N1)
    ref = ref_post.match_lines(rel0, rel1, matches, N0, N1)
    diag(f"lines_match_{seed}", lines0=L0, lines1=L1, matches=len(matches), matched=int(sum(1 for r in ref if r >= 0)))
    assert dev == ref


def test_match_lines_on_assign_output():
    """The reference's call order (frame.cc:125,177-190): AssignPointsToLines on both frames, then MatchLines with the point matches."""
    ctx, _, _ = context("sp")
    rng = np.random.default_rng(17)
    pts0 = rng.uniform(20, 700, size=(300, 2)); pts0[:, 1] *= 480.0 / 752.0
    shift = np.array([-14.0, 0.5])
    pts1 = pts0 + shift + rng.normal(0, 0.2, size=pts0.shape)
    ends = pts0[rng.integers(0, 300, size=(80, 2))]
    lines0 = ends.reshape(80, 4).astype(np.float64)
    lines1 = (ends + shift).reshape(80, 4).astype(np.float64)
    f0, f1 = _feat(pts0), _feat(pts1)
    rel0 = ctx.assign_points_to_lines(lines0, f0)
    rel1 = ctx.assign_points_to_lines(lines1, f1)
    matches = [(i, i) for i in range(0, 300, 1) if rng.uniform() < 0.9]
    dev = ctx.match_lines(rel0, rel1, matches, 300, 300)
    ref = ref_post.match_lines(ref_post.assign_points_to_lines(lines0, f0), ref_post.assign_points_to_lines(lines1, f1), matches, 300, 300)
    assert dev == ref
    assert sum(1 for i, j in enumerate(ref) if j == i) >= 40          # most lines find their shifted copy


def test_match_lines_rejects_malformed_relations():
    """The device indexes bit rows with pt_idx and walks row_ptr: a relation that is not a valid CSR must be refused on the host
    (ADVICE r01), not read out of bounds.  The wrapper builds valid CSR itself, so the C entry point is called directly."""
    import ctypes as C
    ctx, _, _ = context("sp")
    out = np.full((2,), -1, np.int32)
    m = np.array([[0, 0]], np.int32)

    def call(rp0, pi0, rp1, pi1, n0=4, n1=4):
        rp0, pi0, rp1, pi1 = (np.asarray(a, np.int32) for a in (rp0, pi0, rp1, pi1))
        return ctx._l.airfe_match_lines(ctx._h, rp0.ctypes.data, pi0.ctypes.data, 2, n0, rp1.ctypes.data, pi1.ctypes.data, 2, n1,
                                        m.ctypes.data, 1, out.ctypes.data)

    good = ([0, 2, 3], [0, 1, 2])
    assert call(*good, *good) == 0
    assert call([0, 2, 1], [0, 1, 2], *good) != 0               # row_ptr decreases
    assert call([1, 2, 3], [0, 1, 2], *good) != 0               # does not start at 0
    assert call(*good, [0, 2, 3], [0, 1, 7]) != 0               # point index beyond point_num1
    assert call([0, 2, 3], [0, -1, 2], *good) != 0              # negative point index
    assert b"CSR" in C.string_at(ctx._l.airfe_last_error(ctx._h))
    assert call(*good, *good) == 0                              # the context is still usable
