"""Camera::UndistortImage (cv::remap INTER_LINEAR, src/camera.cc:161-182) in front of the detector.
CPU part: the oracle's restatement of OpenCV's fixed-point remap on cases whose answer is known without OpenCV.
GPU part: the HIP kernel bit-exact against that restatement, and rectify -> detect without the image leaving the device."""
import numpy as np
import pytest

from airslam_amd import synth
from oracle import ref_post


def _grid(h, w):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    return xx, yy


def test_remap_oracle_known_answers():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(37, 53)).astype(np.uint8)
    xx, yy = _grid(*img.shape)
    # identity map: exact copy (the 32767 / 1 weight pair never changes a pixel)
    np.testing.assert_array_equal(ref_post.remap_linear_u8(img, xx, yy), img)
    # integer shift: copy with a zero border where the source falls outside
    out = ref_post.remap_linear_u8(img, xx + 5, yy - 3)
    np.testing.assert_array_equal(out[3:, :-5], img[:-3, 5:])
    assert (out[:3] == 0).all() and (out[:, -5:] == 0).all()
    # half-pixel shift in x: (a + b + 1) >> 1 of horizontal neighbours (weights 16384 / 16384, rounding constant 2^14)
    out = ref_post.remap_linear_u8(img, xx + 0.5, yy)
    a, b = img[:, :-1].astype(np.int32), img[:, 1:].astype(np.int32)
    np.testing.assert_array_equal(out[:, :-1], ((a + b + 1) >> 1).astype(np.uint8))
    np.testing.assert_array_equal(out[:, -1], ((img[:, -1].astype(np.int32) + 1) >> 1).astype(np.uint8))     # right tap outside -> 0
    # coordinates are quantised to 1/32 px with round-half-to-even: 0.515625 = 16.5/32 -> 16 ; 0.546875 = 17.5/32 -> 18
    o1 = ref_post.remap_linear_u8(img, xx + np.float32(0.515625), yy)
    o2 = ref_post.remap_linear_u8(img, xx + np.float32(0.5), yy)
    np.testing.assert_array_equal(o1, o2)
    o3 = ref_post.remap_linear_u8(img, xx + np.float32(0.546875), yy)
    o4 = ref_post.remap_linear_u8(img, xx + np.float32(0.5625), yy)
    np.testing.assert_array_equal(o3, o4)
    # wholly outside and far outside (saturating short cast) -> 0
    assert (ref_post.remap_linear_u8(img, xx + 1e6, yy) == 0).all() and (ref_post.remap_linear_u8(img, xx, yy - 1e7) == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed", [(480, 752, 1), (480, 640, 2), (37, 53, 3)])
def test_remap_kernel_bit_exact(h, w, seed):
    from gpu_common import context
    ctx, _, _ = context("sp", max_batch=4, enc_chunk=2)
    raw = synth.gabor_image(h, w, seed)
    mx, my = synth.rectify_maps(h, w, seed)
    ctx.set_rectify_maps(seed & 1, mx, my)
    rect, _ = ctx.rectify_detect(seed & 1, raw, detect=False)
    ref = ref_post.remap_linear_u8(raw, mx, my)
    np.testing.assert_array_equal(rect, ref)
    # the BORDER_CONSTANT path: the same map pushed 60 px off the image (taps partly / wholly outside along two edges)
    ctx.set_rectify_maps(seed & 1, mx - 60, my + 45)
    rect3, _ = ctx.rectify_detect(seed & 1, raw, detect=False)
    ref3 = ref_post.remap_linear_u8(raw, mx - 60, my + 45)
    if (h, w) == (480, 752):
        assert 0.02 < (ref3 == 0).mean() < 0.6          # a real zero border AND a real interior
    np.testing.assert_array_equal(rect3, ref3)
    ctx.set_rectify_maps(seed & 1, mx, my)
    # a strided view (cv::Mat ROI) gives the same picture
    big = np.zeros((h, w + 24), np.uint8); big[:, 8:8 + w] = raw
    rect2, _ = ctx.rectify_detect(seed & 1, big[:, 8:8 + w], detect=False)
    np.testing.assert_array_equal(rect2, ref)


@pytest.mark.gpu
def test_rectify_then_detect_equals_detect_on_the_rectified_image():
    from gpu_common import context
    ctx, _, _ = context("sp", max_batch=4, enc_chunk=2)
    raw = synth.gabor_image(480, 752, 9)
    mx, my = synth.rectify_maps(480, 752, 9)
    ctx.set_rectify_maps(0, mx, my)
    rect, feat = ctx.rectify_detect(0, raw)
    np.testing.assert_array_equal(rect, ref_post.remap_linear_u8(raw, mx, my))
    np.testing.assert_array_equal(feat, ctx.detect_points(rect))    # same detector, the image just never left the device
    assert feat.shape[0] > 50
    with pytest.raises(Exception):
        ctx.rectify_detect(1, raw)                                   # no maps for the right side yet
