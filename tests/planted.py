"""Planted-correspondence inputs for the matcher parity tests (CPU-safe: numpy only).

Half of the keypoints of image 1 are noisy copies of keypoints of image 0 (descriptor + 0.05 N(0, I) renormalised, x shifted
by a 12 px disparity), the rest are unrelated; with the structured synthetic weights of airslam_amd.weights the fp32 oracle
matches ~all of the planted half (203 / 400, 514 / 1024), so filter_matches / decode, the [B][cap] match buffers and the
multi-GPU gather carry hundreds of entries instead of the 0-11 a Kaiming final_proj produced."""
import numpy as np


def features(n, seed, w=752, h=480):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 256)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xy = np.stack([rng.uniform(4, w - 4, n), rng.uniform(4, h - 4, n)], 1).astype(np.float32)
    f = np.zeros((n, 259), np.float32)
    f[:, 0] = rng.uniform(0.01, 1, n)
    f[:, 1:3] = xy
    f[:, 3:] = d
    return f


def planted_pair(n0, n1, seed, w=752, h=480):
    """-> (f0, f1) feature rows [n, 259] in original pixels; rows 0 .. min(n0, n1)//2 - 1 correspond."""
    f0 = features(n0, seed, w, h)
    f1 = features(n1, seed + 1, w, h)
    k = min(n0, n1) // 2
    rng = np.random.default_rng(seed + 2)
    f1[:k, 3:] = f0[:k, 3:] + 0.05 * rng.normal(size=(k, 256)).astype(np.float32)
    f1[:k, 3:] /= np.linalg.norm(f1[:k, 3:], axis=1, keepdims=True)
    f1[:k, 1] = f0[:k, 1] - 12
    f1[:k, 2] = f0[:k, 2]
    return f0, f1


def normalised(f, w=752, h=480, scale=0.5):
    """PointMatcher::NormalizeKeypoints (src/point_matcher.cc:39-48) on [n, 259] rows."""
    out = f.copy()
    l_inv = np.float32(1.0 / max(w, h) * float(np.float32(scale)))
    out[:, 1] = ((f[:, 1] - np.float32(w // 2)) * l_inv).astype(np.float32)
    out[:, 2] = ((f[:, 2] - np.float32(h // 2)) * l_inv).astype(np.float32)
    return out


def fragile_rows(scores, tol, thr=0.1):
    """Rows of a log-assignment matrix whose filter_matches decision a perturbation of at most `tol` per entry can flip:
    the row maximum within `tol` of log(thr), or the runner-up of its row / of its column within 2 tol of the maximum.
    Parity of match SETS is asserted on all other rows; this set must stay (nearly) empty for the test to mean anything."""
    s = scores.astype(np.float64)
    n0, n1 = s.shape
    out = set()
    if n0 == 0 or n1 == 0:
        return out
    rcol = s.argmax(1)
    rval = s[np.arange(n0), rcol]
    lt = np.log(thr)

    def runner_up_gap(m, axis):
        if m.shape[axis] < 2:
            return np.full(m.shape[1 - axis], np.inf)
        part = np.sort(m, axis=axis)
        return (part.take(-1, axis) - part.take(-2, axis))
    rgap = runner_up_gap(s, 1)
    cgap = runner_up_gap(s, 0)
    for i in range(n0):
        if rval[i] < lt - 2 * tol and s[:, rcol[i]].argmax() != i:
            continue                                    # far below threshold AND not mutual: two flips needed
        if abs(rval[i] - lt) <= tol or (rval[i] > lt - tol and (rgap[i] <= 2 * tol or cgap[rcol[i]] <= 2 * tol)):
            out.add(i)
    return out


def decision_margins(scores, thr=0.1):
    """Per row of a log-assignment matrix: how far (in score units) the nearest entry change is that flips the row's filter_matches decision —
    min(|row max - log thr|, half the gap to the runner-up of its row, half the gap to the runner-up of its column).  A row on which the device
    and the oracle DISAGREE must have a margin below twice the measured score error: disagreements are then explained, not exempted."""
    s = np.where(np.isfinite(scores), scores, -1e30).astype(np.float64)
    n0, n1 = s.shape
    if n0 == 0 or n1 == 0:
        return np.zeros(n0)
    rcol = s.argmax(1)
    rval = s[np.arange(n0), rcol]

    def gap(m, axis):
        if m.shape[axis] < 2:
            return np.full(m.shape[1 - axis], np.inf)
        part = np.sort(m, axis=axis)
        return part.take(-1, axis) - part.take(-2, axis)
    rgap, cgap = gap(s, 1), gap(s, 0)
    return np.minimum(np.abs(rval - np.log(thr)), np.minimum(rgap, cgap[rcol]) / 2)
