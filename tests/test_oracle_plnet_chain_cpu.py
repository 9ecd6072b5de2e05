"""The oracle's PLNet chain as bench.py's CPU baseline runs it: one trunk pass per image feeding the point heads and the line branch
(`superpoint_trunk(taps=...)` + `plnet_s0_lines(f3a=...)`) must be the two-pass form bit for bit, and the baseline leg must run."""
import importlib.util
import os

import numpy as np
import torch

from airslam_amd import synth, weights
from conftest import GOLDEN
from oracle import ref_nets, ref_post

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_line_branch_on_the_point_branch_trunk_is_the_same():
    w = weights.synthetic_plnet_s0(1234)
    x, _, _ = ref_post.process_image(synth.gabor_image(480, 752, 8))
    with torch.no_grad():
        taps = {}
        f = ref_nets.superpoint_trunk(w, torch.from_numpy(x)[None, None], taps)
        heat, desc = (t.numpy() for t in ref_nets.superpoint_heads(w, f))
    heat2, desc2 = ref_nets.superpoint_forward(w, x[None])
    np.testing.assert_array_equal(heat, heat2)
    np.testing.assert_array_equal(desc, desc2)
    a = ref_nets.plnet_s0_lines(w, x)
    b = ref_nets.plnet_s0_lines(w, x, f3a=taps["conv3a"])
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["iskeep"].sum() > 1000


def test_cpu_baseline_runs_the_keyframe_step():
    from benchlib import cpu as bench_cpu          # bench.py's cpu_baseline leg (the only part of the bench that imports oracle/)
    r = bench_cpu.stereo(weights.synthetic_plnet_s0(1234), weights.synthetic_lightglue(1234), 480, 752, 1, 400, warm=0,
                           s1=weights.load_pack(os.path.join(GOLDEN, "plnet_s1.airfe")))
    assert r["unit"] == "pairs/s" and r["value"] > 0 and r["kind"] == "port" and "PLNet points + lines + junctions" in r["sample"]
