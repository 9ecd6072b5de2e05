"""fp16 activation range (VERDICT r04 #4a): weights.fold_activation_scales rescales the detector between layers by exact powers of two — the function the network
computes does not change (fp32 oracle: the same bits), a healthy network is left untouched, a hostile one is brought back inside the fp16 range."""
import numpy as np

from airslam_amd import synth, weights
from oracle import ref_nets, ref_post


def hostile(seed=1234):
    """a PLNet stage-0 pack whose conv3b .. convDa activations reach 1e5 - 3e5 (fp16 maximum: 65504): six conv layers scaled up by 8 / 8 / 8 / 8 / 8 / 2"""
    w = weights.synthetic_plnet_s0(seed)
    for name, f in (("conv1b", 8.0), ("conv2a", 8.0), ("conv2b", 8.0), ("conv3a", 8.0), ("conv3b", 8.0), ("conv4a", 2.0)):
        w[name + ".weight"] = (w[name + ".weight"] * np.float32(f)).astype(np.float32)
    return w


def test_a_healthy_network_is_left_untouched():
    w = weights.synthetic_plnet_s0(1234)
    f, rep = weights.fold_activation_scales(w)
    assert all(c == 1.0 for _, c in rep.values()) and max(m for m, _ in rep.values()) < 16
    for k in w:
        np.testing.assert_array_equal(f[k], w[k])


def test_folding_keeps_the_function_and_restores_the_range():
    w = hostile()
    mx = weights.activation_maxima(w)
    assert max(mx.values()) > 65504 * 1.5, mx                      # the unfolded pack overflows fp16
    f, rep = weights.fold_activation_scales(w)
    after = weights.activation_maxima(f)
    assert max(after.values()) <= weights.ACT_TARGET and min(c for _, c in rep.values()) < 1.0
    assert all(np.log2(c) == np.round(np.log2(c)) for _, c in rep.values())
    img = synth.gabor_image(480, 752, 7)                            # (not a calibration frame)
    x, ws, hs = ref_post.process_image(img)
    h0, d0 = ref_nets.superpoint_forward(w, x[None])
    h1, d1 = ref_nets.superpoint_forward(f, x[None])
    # power-of-two factors commute with every fp32 rounding: the score map and the normalised descriptors come out bit for bit
    np.testing.assert_array_equal(h0, h1)
    np.testing.assert_array_equal(d0, d1)
    s0 = ref_nets.plnet_s0_lines(w, x)
    s1 = ref_nets.plnet_s0_lines(f, x)
    for k in ("juncs_pred", "lines_pred", "iskeep", "loi_features"):
        np.testing.assert_array_equal(s0[k], s1[k])


def test_the_hostile_detector_is_the_healthy_function_with_overflowing_activations():
    from hostile import hostile_detector
    h, w = hostile_detector(), weights.synthetic_plnet_s0(1234)
    mx = weights.activation_maxima(h)
    assert max(mx.values()) > 2 * weights.FP16_MAX
    x, _, _ = ref_post.process_image(synth.gabor_image(480, 752, 7))
    a, b = ref_nets.superpoint_forward(w, x[None]), ref_nets.superpoint_forward(h, x[None])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    f, rep = weights.fold_activation_scales(h)
    assert max(weights.activation_maxima(f).values()) <= weights.ACT_TARGET
    c = ref_nets.superpoint_forward(f, x[None])
    np.testing.assert_array_equal(a[0], c[0])
