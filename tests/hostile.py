"""Weights that stress the 2-byte storage ranges: test infrastructure, CPU-safe."""
import numpy as np

from airslam_amd import weights

# intermediate activations of the healthy synthetic detector are <= 4.2; these factors take conv3b .. convDa (and line.conv1) to 1e5 - 3e5, the SAME function
DETECTOR_FACTORS = {"conv1b": 8.0, "conv2a": 64.0, "conv2b": 512.0, "conv3a": 4096.0, "conv3b": 32768.0, "conv4a": 65536.0, "conv4b": 65536.0, "convPa": 65536.0,
                    "convDa": 65536.0, "line.conv1": 32768.0}


def hostile_detector(seed=1234):
    """the healthy synthetic PLNet stage-0 pack re-parameterised so that its 2-byte activations overflow fp16 (max ~2.7e5 > 65504) while the fp32 function is bit
    for bit the healthy one's (weights.rescale_activations: exact powers of two)"""
    return weights.rescale_activations(weights.synthetic_plnet_s0(seed), DETECTOR_FACTORS)


def hostile_lightglue(seed=1234, ln_gain=1.0, outliers=0, outlier_gain=30.0, logit_gain=1.0):
    """structured synthetic LightGlue + LayerNorm gains drawn from 1 .. ln_gain, `outliers` hidden channels of every ffn.0 scaled by outlier_gain, final projection
    scaled by logit_gain (log-assignment range grows with its square)"""
    w = weights.synthetic_lightglue(seed)
    rng = np.random.default_rng(seed + 77)
    for k in list(w):
        if k.endswith("ffn.1.weight") and ln_gain != 1.0:
            w[k] = (w[k] * rng.uniform(1.0, ln_gain, size=w[k].shape)).astype(np.float32)
        if k.endswith("ffn.0.weight") and outliers:
            rows = rng.choice(w[k].shape[0], outliers, replace=False)
            w[k][rows] *= np.float32(outlier_gain)
    a = [k for k in w if k.endswith("final_proj.weight")][0]
    w[a] = (w[a] * np.float32(logit_gain)).astype(np.float32)
    return w
