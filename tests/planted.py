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


from oracle.margins import decision_margins, fragile_rows  # noqa: E402,F401  (moved: bench.py's cpu_baseline leg reports the fragile share too)
