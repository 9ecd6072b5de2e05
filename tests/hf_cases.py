"""The seeded inputs of the Hugging Face pin (oracle/hf_pin.py), shared by the CPU test, the fixture generator
(tools/make_hf_fixtures.py -> tests/golden/hf_pin.npz) and the GPU twin, so that all three run the same arrays."""
import numpy as np

from airslam_amd import synth
from oracle import ref_post
from planted import normalised, planted_pair

SP_IMAGES = [(480, 752, 0), (720, 1280, 3)]                 # (h, w, seed) of synth.gabor_image; the path resizes to 512 x 512 (src/plnet.cpp:258)
LG_PAIRS = [(200, 200, 7), (317, 400, 1351), (1, 5, 8)]     # (n0, n1, seed)
SG_PAIRS = [(200, 200, 7), (300, 280, 1180), (1, 3, 6)]
DESC_STRIDE = 2                                             # the fixture keeps the dense descriptors of every second cell row / column (1 MB per image)


def sp_input(h, w, seed):
    """-> (gray uint8 [h, w], x float32 [512, 512] in [0, 1] = PLNet::process_image, src/plnet.cpp:246-270)"""
    img = synth.gabor_image(h, w, seed)
    x, _, _ = ref_post.process_image(img)
    return img, x


def lg_input(n0, n1, seed):
    f0, f1 = planted_pair(n0, n1, seed)
    a, b = normalised(f0), normalised(f1)                   # scale 0.5: src/point_matcher.cc:58
    return f0, f1, a, b


def sg_input(n0, n1, seed):
    f0, f1 = planted_pair(n0, n1, seed)
    a, b = normalised(f0, scale=0.7), normalised(f1, scale=0.7)
    return f0, f1, a, b
