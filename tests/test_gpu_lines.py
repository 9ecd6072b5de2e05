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
