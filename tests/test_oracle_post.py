"""The numpy restatements of the reference's CPU routines against slow literal transcriptions of the C++ loops
(small sizes) and against their defining properties."""
import numpy as np
import pytest

from oracle import ref_post as rp


def _detect_point_loops(heat, thr, border, top_k):
    """Literal transcription of src/plnet.cpp:309-355 with the SURVEY B.1 tie rule."""
    h, w = heat.shape
    cand = []
    for i in range(h * w):
        s = heat.flat[i]
        if s < np.float32(thr):
            continue
        y = i // w
        x = i - y * w
        if x < border or x > w - border or y < border or y > h - border:
            continue
        cand.append((float(s), i, x, y))
    if len(cand) > top_k:
        cand.sort(key=lambda c: (-c[0], c[1]))
        cand = cand[:top_k]
    return (np.array([c[0] for c in cand], np.float32), np.array([c[2] for c in cand], np.float32),
            np.array([c[3] for c in cand], np.float32))


@pytest.mark.parametrize("seed,top_k", [(0, 50), (1, 5000), (2, 1)])
def test_detect_point(seed, top_k):
    rng = np.random.default_rng(seed)
    heat = (rng.random((40, 48)) ** 6).astype(np.float32)
    heat[5, 7] = heat[9, 9] = heat[20, 30] = 0.9          # exact ties
    a = rp.detect_point(heat, 0.05, 4, top_k)
    b = _detect_point_loops(heat, 0.05, 4, top_k)
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


def test_detect_point_border_inclusive_upper():
    heat = np.zeros((32, 32), np.float32)
    heat[28, 28] = 1.0      # x == w - border -> kept (upper bound inclusive, plnet.cpp:332)
    heat[29, 10] = 1.0      # y > h - border -> dropped
    heat[3, 10] = 1.0       # y < border -> dropped
    s, x, y = rp.detect_point(heat, 0.5, 4, 10)
    assert list(zip(x, y)) == [(28.0, 28.0)]


def test_extract_descriptors_matches_loops():
    rng = np.random.default_rng(3)
    d = rng.normal(size=(256, 8, 8)).astype(np.float32)
    xs = np.array([0, 5, 31.0, 63, 62, 17], np.float32)
    ys = np.array([0, 63, 2.0, 63, 1, 40], np.float32)
    out = rp.extract_descriptors(d, xs, ys, 8)
    # literal loop (double precision reference of the same formula) for a loose cross-check
    h = w = 8
    s = 8
    for j in range(len(xs)):
        sx = 2.0 / (w * s - s // 2 - 0.5); bx = (1 - s) / (w * s - s // 2 - 0.5) - 1
        ix = ((xs[j] * sx + bx) + 1) * 0.5 * (w - 1)
        iy = ((ys[j] * sx + bx) + 1) * 0.5 * (h - 1)
        x0 = min(max(int(np.floor(ix)), 0), w - 1); y0 = min(max(int(np.floor(iy)), 0), h - 1)
        x1 = min(x0 + 1, w - 1); y1 = min(y0 + 1, h - 1)
        v = (d[:, y0, x0] * (x1 - ix) * (y1 - iy) + d[:, y0, x1] * (ix - x0) * (y1 - iy)
             + d[:, y1, x0] * (x1 - ix) * (iy - y0) + d[:, y1, x1] * (ix - x0) * (iy - y0))
        n = np.linalg.norm(v)
        if n > 0:
            np.testing.assert_allclose(out[j], v / n, atol=2e-5)
    # far edge (ix == w-1): clamped neighbours collapse every weight to 0 -> a zero column, which Eigen (>= 3.3, what the
    # reference's ROS/Ubuntu 20.04 toolchain ships) leaves untouched: MatrixBase::normalize() divides only `if (z > 0)`.
    # With remove_borders >= 1 at 512x512 this never happens in the pipeline.
    edge = np.array([False, True, False, True, False, False])
    assert (out[edge] == 0).all()
    assert np.allclose(np.linalg.norm(out[~edge], axis=1), 1, atol=1e-5)


def test_resize_identity_and_range():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(512, 512), dtype=np.uint8)
    np.testing.assert_array_equal(rp.resize_linear_u8(img, 512, 512), img)
    src = rng.integers(0, 256, size=(480, 752), dtype=np.uint8)
    out = rp.resize_linear_u8(src, 512, 512)
    assert out.shape == (512, 512) and out.dtype == np.uint8
    const = np.full((480, 752), 137, np.uint8)
    np.testing.assert_array_equal(rp.resize_linear_u8(const, 512, 512), np.full((512, 512), 137, np.uint8))
    # within +-1 grey level of float bilinear interpolation with half-pixel centres
    yy = np.clip((np.arange(512) + 0.5) * 480 / 512 - 0.5, 0, 479); xx = np.clip((np.arange(512) + 0.5) * 752 / 512 - 0.5, 0, 751)
    y0 = np.floor(yy).astype(int); x0 = np.floor(xx).astype(int)
    y1 = np.minimum(y0 + 1, 479); x1 = np.minimum(x0 + 1, 751)
    fy = (yy - y0)[:, None]; fx = (xx - x0)[None, :]
    s = src.astype(np.float64)
    ref = (s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx) * (1 - fy) + (s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx) * fy
    assert np.abs(out.astype(np.float64) - ref).max() <= 1.0


def test_wireframe_matcher_matches_loops():
    rng = np.random.default_rng(5)
    n = 3 * 16 * 16
    iskeep = (rng.random(n) < 0.2).astype(np.float32)
    a = rng.integers(0, 12, n); b = rng.integers(0, 12, n)
    mn, mx = np.minimum(a, b).astype(np.float32), np.maximum(a, b).astype(np.float32)
    keep, inv, pairs = rp.wireframe_matcher(iskeep, mn, mx)
    # literal transcription of src/plnet.cpp:272-307
    k2 = [i for i in range(n) if iskeep[i] > 0]
    table = {}
    inv2 = []
    for i in k2:
        key = (int(mn[i]), int(mx[i]))
        if key not in table:
            table[key] = len(table) + 1
        inv2.append(table[key] - 1)
    pairs2 = [None] * len(table)
    for (x, y), v in table.items():
        pairs2[v - 1] = (y, x)
    np.testing.assert_array_equal(keep, k2)
    np.testing.assert_array_equal(inv, inv2)
    np.testing.assert_array_equal(pairs, np.array(pairs2))


def test_filter_matches_matches_loops():
    rng = np.random.default_rng(6)
    sc = np.log(rng.random((37, 29)).astype(np.float32) ** 3 + 1e-9).astype(np.float32)
    sc[3, 4] = sc[3, 9] = 0.0          # tie in a row: first column wins
    idx, val = rp.filter_matches(sc, 0.1)
    # literal transcription of src/light_glue.cpp:214-266
    rmax = []
    for r in range(sc.shape[0]):
        best, bc = -np.inf, 0
        for c in range(sc.shape[1]):
            if sc[r, c] > best:
                best, bc = sc[r, c], c
        rmax.append((bc, best))
    cmax = []
    for c in range(sc.shape[1]):
        best, br = -np.inf, 0
        for r in range(sc.shape[0]):
            if sc[r, c] > best:
                best, br = sc[r, c], r
        cmax.append(br)
    exp = [(r, rmax[r][0], np.exp(np.float32(rmax[r][1]))) for r in range(sc.shape[0])
           if cmax[rmax[r][0]] == r and np.exp(np.float32(rmax[r][1])) > 0.1]
    assert [tuple(i) for i in idx] == [(e[0], e[1]) for e in exp]
    np.testing.assert_allclose(val, [e[2] for e in exp], rtol=1e-6)
    assert np.all(np.diff(idx[:, 0]) > 0)


def test_superglue_decode_properties():
    rng = np.random.default_rng(7)
    z = np.log(rng.random((21, 18)).astype(np.float32) + 1e-6).astype(np.float32)
    i0, i1, m0, m1 = rp.superglue_decode(z)
    assert i0.shape == (20,) and i1.shape == (17,)
    for i, j in enumerate(i0):
        if j >= 0:
            assert i1[j] == i and m0[i] > 0.2
    assert np.all((i0 >= -1) & (i0 < 17))


def test_sinkhorn_marginals():
    rng = np.random.default_rng(8)
    s = rng.normal(size=(12, 9)).astype(np.float32)
    z = rp.log_optimal_transport(s, 1.0, 100)
    p = np.exp(z.astype(np.float64))
    # rows of the coupling sum to the prescribed marginals (x (m+n) after the -norm shift)
    np.testing.assert_allclose(p[:12].sum(1), 1.0, atol=1e-3)
    np.testing.assert_allclose(p[:, :9].sum(0), 1.0, atol=1e-3)


def test_normalize_keypoints_integer_halves():
    f = np.zeros((3, 259), np.float32)
    f[:, 1] = [0, 376, 751]; f[:, 2] = [0, 240, 479]
    o = rp.normalize_keypoints(f, 752, 480, 0.5)
    linv = np.float32(1.0 / 752 * 0.5)
    np.testing.assert_array_equal(o[:, 1], (f[:, 1] - 376) * linv)
    np.testing.assert_array_equal(o[:, 2], (f[:, 2] - 240) * linv)


# ---- AssignPointsToLines (src/line_processor.cc:68-120)
def _feat_rows(xy):
    f = np.zeros((len(xy), 259), np.float32)
    f[:, 1:3] = np.asarray(xy, np.float32)
    return f


def test_assign_points_to_lines_hand_cases():
    lines = np.array([[10.0, 10.0, 50.0, 10.0],      # horizontal segment
                      [20.0, 5.0, 20.0, 5.0]])       # degenerate (zero length): D = 0, distance NaN -> only the <= 9 tests decide
    pts = _feat_rows([(30.0, 12.0),    # 2 px above the segment, projection inside      -> on line 0, dist 2
                      (30.0, 13.5),    # 3.5 px away                                   -> off
                      (52.5, 10.0),    # beyond the end by 2.5 px: side2 = 6.25 <= 9   -> on line 0, dist 0
                      (54.0, 10.0),    # outside the +-3 box                           -> off
                      (21.0, 6.0),     # 1.41 px from the degenerate line's point      -> on line 1 (side1 = 2 <= 9), dist NaN
                      (10.0, 7.0)])    # exactly 3 px: pl_distance > 3 is false        -> on line 0, dist 3
    rel = rp.assign_points_to_lines(lines, pts)
    assert list(rel[0].keys()) == [0, 2, 5]
    assert rel[0][0] == 2.0 and rel[0][2] == 0.0 and rel[0][5] == 3.0
    assert list(rel[1].keys()) == [4] and np.isnan(rel[1][4])


def test_assign_points_to_lines_vectorised_cross_check():
    """The statement-by-statement loops agree with an independent vectorised formulation on random data."""
    rng = np.random.default_rng(11)
    lines = rng.uniform(0, 200, size=(40, 4))
    pts = _feat_rows(rng.uniform(0, 200, size=(300, 2)))
    rel = rp.assign_points_to_lines(lines, pts)
    x, y = pts[:, 1].astype(np.float64), pts[:, 2].astype(np.float64)
    for i, (x1, y1, x2, y2) in enumerate(lines):
        A, B, C = y2 - y1, x1 - x2, x2 * y1 - x1 * y2
        D = np.sqrt(A * A + B * B)
        pl = (np.abs(A * x + B * y + C) / D).astype(np.float32)
        box = (x >= min(x1, x2) - 3) & (x <= max(x1, x2) + 3) & (y >= min(y1, y2) - 3) & (y <= max(y1, y2) + 3)
        s1 = (x1 - x) ** 2 + (y1 - y) ** 2
        s2 = (x2 - x) ** 2 + (y2 - y) ** 2
        ok = box & ~(pl > 3) & ((s1 <= 9) | (s2 <= 9) | ((s1 < D * D + s2) & (s2 < D * D + s1)))
        assert list(rel[i].keys()) == np.nonzero(ok)[0].tolist()
        assert np.array_equal(np.array(list(rel[i].values()), np.float32), pl[ok])


def _random_line_frames(seed, L0, L1, N0, N1, M):
    """Two frames' point-line relations that share structure (so that votes >= 2 exist) + point matches."""
    rng = np.random.default_rng(seed)
    rel0 = [dict() for _ in range(L0)]
    rel1 = [dict() for _ in range(L1)]
    perm = rng.permutation(N1)                    # true correspondence point i (frame 0) <-> perm[i] (frame 1), where it exists
    for i in range(L0):
        for k in rng.choice(N0, size=int(rng.integers(0, min(9, N0) + 1)), replace=False) if N0 else []:
            rel0[i][int(k)] = float(rng.uniform(0, 3))
    for j in range(L1):
        src = rel0[j % L0] if L0 and rng.uniform() < 0.7 else {}
        for k in src:
            if k < N1 and rng.uniform() < 0.8:
                rel1[j][int(perm[k % N1])] = float(rng.uniform(0, 3))
        for k in rng.choice(N1, size=int(rng.integers(0, min(4, N1) + 1)), replace=False) if N1 else []:
            rel1[j][int(k)] = float(rng.uniform(0, 3))
    matches = []
    if N0 and N1:
        for q in rng.choice(N0, size=min(M, N0), replace=False):
            t = int(perm[q % N1]) if rng.uniform() < 0.85 else int(rng.integers(0, N1))
            matches.append((int(q), t))
    return rel0, rel1, matches


def test_match_lines_hand_cases():
    """MatchLines (line_processor.cc:122-172): votes, mutual first-maximum, the >= 2 and score >= 0.8 gates."""
    rel0 = [{0: 1.0, 1: 1.0, 2: 1.0}, {3: 0.5}, {4: 0.1, 5: 0.2}]
    rel1 = [{7: 1.0}, {10: 1.0, 11: 1.0, 12: 1.0}, {14: 0.3, 15: 0.3, 16: 0.1, 17: 0.2, 18: 0.0, 19: 0.0}]
    matches = [(0, 10), (1, 11), (2, 12), (3, 7), (4, 14), (5, 15)]
    # line 0 <-> line 1: 3 votes, score 9/3 = 3; line 1 <-> 0: one vote only (< 2); line 2 <-> 2: 2 votes, score 4/2 = 2 -> accepted
    assert rp.match_lines(rel0, rel1, matches, 6, 20) == [1, -1, 2]
    # score gate: 2 votes over lines of 6 points each: 4/6 < 0.8
    rel0b = [{0: 0., 1: 0., 2: 0., 3: 0., 4: 0., 5: 0.}]
    rel1b = [{0: 0., 1: 0., 2: 0., 3: 0., 4: 0., 5: 0.}]
    assert rp.match_lines(rel0b, rel1b, [(0, 0), (1, 1)], 6, 6) == [-1]
    assert rp.match_lines(rel0b, rel1b, [(0, 0), (1, 1), (2, 2), (3, 3), (4, 4)], 6, 6) == [0]      # 25/6
    # early-outs (:132) and a tie: two frame-1 lines with equal votes -> the first one wins the row, the second is not mutual
    assert rp.match_lines(rel0, rel1, matches, 0, 20) == [-1, -1, -1]
    assert rp.match_lines([], rel1, matches, 6, 20) == []
    tie1 = [{10: 0., 11: 0.}, {10: 0., 11: 0.}]
    assert rp.match_lines([{0: 0., 1: 0.}], tie1, [(0, 10), (1, 11)], 2, 12) == [0]


def test_match_lines_matrix_cross_check():
    """The voting matrix of the statement-by-statement port equals the indicator-matrix product A0^T . P . A1 on random frames."""
    for seed in range(6):
        L0, L1, N0, N1 = 23 + seed, 19 + 2 * seed, 60, 70
        rel0, rel1, matches = _random_line_frames(seed, L0, L1, N0, N1, 45)
        a0 = np.zeros((L0, N0), np.int64); a1 = np.zeros((L1, N1), np.int64); pm = np.zeros((N0, N1), np.int64)
        for i, r in enumerate(rel0):
            a0[i, list(r)] = 1
        for j, r in enumerate(rel1):
            a1[j, list(r)] = 1
        for q, t in matches:
            pm[q, t] += 1
        mat = a0 @ pm @ a1.T
        want = [-1] * L0
        row_loc = mat.argmax(1)
        for j in range(L1):
            i = int(mat[:, j].argmax()); v = int(mat[i, j])
            if v >= 2 and row_loc[i] == j and np.float32(v * v) / np.float32(min(len(rel0[i]), len(rel1[j]))) >= 0.8:
                want[i] = j
        assert rp.match_lines(rel0, rel1, matches, N0, N1) == want
        assert any(w >= 0 for w in want)


def test_glibc_expf_restatement_equals_the_hosts_libm(tmp_path):
    """`std::exp(float)` of the reference's host code (src/light_glue.cpp:248, src/super_glue.cpp:299) is glibc's expf; the device restates its algorithm
    (airslam_amd/csrc/common.h expf_like_glibc), and tools/expf_glibc_check.c is that restatement in C, compared with the host's libm.  Every 257th float of both
    signs here (16.6 million values, 0 mismatches expected; the whole range: the tool without a stride, ~90 s) — on a host whose libm selects another build of expf
    (no FMA) this test says so instead of letting the GPU pin drift.  Also: the oracle's _expf IS libm's."""
    import ctypes
    import os
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "expf_check")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "expf_glibc_check.c")
    subprocess.run([gcc, "-O2", "-mfma", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    for sign in ("0", "1"):
        out = subprocess.run([exe, sign, "257"], check=True, capture_output=True, text=True).stdout
        assert "mismatches=0" in out and "n=8323328" in out, out
    x = np.float32([-2.3025851, -0.6931472, -1e-8, 0.0, -17.5, -103.0, 88.0])
    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    want = np.float32([libm.expf(float(v)) for v in x])
    assert np.array_equal(np.asarray(rp._expf(x), np.float32).view(np.uint32), want.view(np.uint32))
