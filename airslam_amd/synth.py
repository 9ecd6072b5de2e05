"""Seeded synthetic inputs for benchmarks and parity tests.

No dataset ships with the reference (EuRoC etc. are external), so the harness feeds
synthetic stereo pairs: a smooth texture of random 2-D Gabor blobs plus noise; the right
image is the left one shifted by a per-row disparity so that matches exist.  Also a
stand-in generator for the PLNet stage-0 *line-branch* tensors (Appendix A.1 contract),
because plnet_s0.onnx is missing from the reference checkout.
"""
from __future__ import annotations

import numpy as np


def gabor_image(h: int, w: int, seed: int, n_blobs: int = 64) -> np.ndarray:
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.full((h, w), 110.0, dtype=np.float32)
    for _ in range(n_blobs):
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        sig = rng.uniform(6, 40)
        th = rng.uniform(0, np.pi)
        lam = rng.uniform(6, 30)
        amp = rng.uniform(20, 70)
        xr = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        env = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sig * sig))
        img += amp * env * np.cos(2 * np.pi * xr / lam + rng.uniform(0, 2 * np.pi))
    img += rng.normal(0, 4.0, size=(h, w)).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def stereo_pair(h: int, w: int, seed: int):
    """(left, right) uint8 [h,w]; right = left shifted left by a smooth 8..40 px disparity."""
    left = gabor_image(h, w + 48, seed)
    rng = np.random.default_rng(seed + 7919)
    d0 = rng.uniform(8, 40)
    disp = (d0 + 6.0 * np.sin(np.arange(h) / h * 2 * np.pi + rng.uniform(0, 6))).clip(8, 40).astype(np.int64)
    right = np.empty((h, w), dtype=np.uint8)
    for r in range(h):
        right[r] = left[r, disp[r]:disp[r] + w]
    noise = rng.normal(0, 2.0, size=(h, w))
    right = np.clip(right.astype(np.float32) + noise, 0, 255).astype(np.uint8)
    return np.ascontiguousarray(left[:, :w]), right


def stereo_batch(b: int, h: int, w: int, seed: int):
    """[b,h,w] left and right uint8 stacks; cheap: one generated pair, rolled per item."""
    l0, r0 = stereo_pair(h, w, seed)
    ls = np.stack([np.roll(l0, (3 * i, 5 * i), axis=(0, 1)) for i in range(b)])
    rs = np.stack([np.roll(r0, (3 * i, 5 * i), axis=(0, 1)) for i in range(b)])
    return ls, rs


def stereo_sequence(n: int, h: int, w: int, seed: int, scene_len: int = 40):
    """n stereo frames of a camera panning over synthetic scenes (a new scene every `scene_len` frames): the stand-in for a dataset sequence
    (EuRoC MH_01 is external, src/dataset.cc:20-21 reads cam0 / cam1 image folders).  Yields (left, right) uint8 [h, w]."""
    cur, l0, r0 = -1, None, None
    for i in range(n):
        scene, j = divmod(i, scene_len)
        if scene != cur:
            l0, r0 = stereo_pair(h, w, seed + 101 * scene)
            cur = scene
        yield np.roll(l0, (2 * j, 3 * j), axis=(0, 1)), np.roll(r0, (2 * j, 3 * j), axis=(0, 1))


def stereo_sequence_arrays(args):
    """(n, h, w, seed, scene_len) -> (left [n,h,w], right [n,h,w]) uint8: stereo_sequence as two stacks (a picklable job for a process pool)."""
    n, h, w, seed, scene_len = args
    fr = list(stereo_sequence(n, h, w, seed, scene_len))
    return np.stack([f[0] for f in fr]), np.stack([f[1] for f in fr])


def plnet_stage0_lines(seed: int, n_lines: int = 400, fh: int = 128, fw: int = 128, jn: int = 300):
    """Stand-in for the line-branch outputs of plnet_s0.onnx.
    Shapes/dtypes follow the contract read off src/plnet.cpp:453-507; the VALUES are synthetic."""
    rng = np.random.default_rng(seed)
    juncs = (rng.integers(8, 4 * fw - 8, size=(jn, 2)) / 4.0).astype(np.float32)     # x4 is a whole number
    lines_pred = rng.uniform(1, fw - 1, size=(3 * fh * fw, 4)).astype(np.float32)
    iskeep = np.zeros(3 * fh * fw, dtype=np.float32)
    pos = rng.choice(3 * fh * fw, size=n_lines, replace=False)
    iskeep[pos] = 1.0
    a = rng.integers(0, jn, size=3 * fh * fw)
    b = rng.integers(0, jn, size=3 * fh * fw)
    # a small junction pool for kept proposals so that duplicates (min,max) pairs occur
    pool = rng.integers(0, jn, size=(max(n_lines // 3, 1), 2))
    pick = pool[rng.integers(0, pool.shape[0], size=n_lines)]
    a[pos], b[pos] = pick[:, 0], pick[:, 1]
    idx_min = np.minimum(a, b).astype(np.float32)
    idx_max = np.maximum(a, b).astype(np.float32)
    loi = rng.normal(0, 1, size=(128, fh, fw)).astype(np.float32)
    thin = rng.normal(0, 1, size=(4, fh, fw)).astype(np.float32)
    aux = rng.normal(0, 1, size=(4, fh, fw)).astype(np.float32)
    return dict(juncs_pred=juncs, lines_pred=lines_pred,
                iskeep=iskeep.reshape(1, 3, fh, fw), idx_junc_to_end_min=idx_min.reshape(1, 3, fh, fw),
                idx_junc_to_end_max=idx_max.reshape(1, 3, fh, fw),
                loi_features=loi[None], loi_features_thin=thin[None], loi_features_aux=aux[None])


def rectify_maps(h: int, w: int, seed: int, k1: float = -0.28, k2: float = 0.07, rot_deg: float = 1.5):
    """Stand-in for cv::initUndistortRectifyMap's CV_32FC1 output (Camera's constructor, src/camera.cc:60-75): for every pixel of
    the rectified image the float source coordinates in the raw (radially distorted, slightly rotated) image.  EuRoC-like
    coefficients; corners map outside the raw image, so the BORDER_CONSTANT path is exercised."""
    rng = np.random.default_rng(seed)
    fx = fy = 0.61 * w
    cx, cy = w / 2 + rng.uniform(-8, 8), h / 2 + rng.uniform(-6, 6)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    x = (xx - w / 2) / fx; y = (yy - h / 2) / fy
    a = np.deg2rad(rot_deg) * rng.uniform(0.5, 1.0)
    xr = x * np.cos(a) - y * np.sin(a); yr = x * np.sin(a) + y * np.cos(a)
    r2 = xr * xr + yr * yr
    d = 1 + k1 * r2 + k2 * r2 * r2
    return (xr * d * fx + cx).astype(np.float32), (yr * d * fy + cy).astype(np.float32)
