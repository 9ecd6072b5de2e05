"""Error paths of the C boundary on the GPU: a failed kernel launch is reported by the call that made it, under the NAME of the stage
it belongs to (cfg.check_launches); the context stays usable; argument combinations that used to be dropped silently are refused."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _ctx(**kw):
    return api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=weights.synthetic_lightglue(1234, n_layers=2), max_batch=2, enc_chunk=2, **kw)


@pytest.mark.parametrize("stage", ["conv1_fused", "conv3x3_cin128", "head_gemm", "simple_nms", "sample_desc"])
def test_a_failed_launch_names_its_stage(stage):
    img = synth.gabor_image(480, 752, 3)
    ctx = _ctx(check_launches=1)
    want = ctx.detect_points(img)
    ctx.debug_fail_next_launch(stage)                       # one deliberately invalid launch (4096 threads per workgroup) in front of that stage's launches
    with pytest.raises(api.AirfeError, match=rf"{stage}: kernel launch failed"):
        ctx.detect_points(img)
    np.testing.assert_array_equal(ctx.detect_points(img), want)      # the context is usable afterwards, same bits
    ctx.close()


def test_matcher_stages_are_named_too():
    from planted import normalised, planted_pair
    f0, f1 = planted_pair(200, 180, 5)
    a, b = np.ascontiguousarray(normalised(f0)[:, 1:]), np.ascontiguousarray(normalised(f1)[:, 1:])
    ctx = _ctx(check_launches=1)
    want = ctx.match_lightglue(a, b)
    for stage in ("lg_prepare", "lg_attention", "lg_gemm", "lg_assign"):
        ctx.debug_fail_next_launch(stage)
        with pytest.raises(api.AirfeError, match=rf"{stage}: kernel launch failed"):
            ctx.match_lightglue(a, b)
        got = ctx.match_lightglue(a, b)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
    ctx.close()


def test_without_check_launches_the_failure_still_surfaces_at_the_end_of_the_pipeline():
    img = synth.gabor_image(480, 752, 3)
    ctx = _ctx(check_launches=0)
    ctx.debug_fail_next_launch("conv3x3_cin128")
    with pytest.raises(api.AirfeError, match="kernel launch failed"):
        ctx.detect_points(img)
    assert ctx.detect_points(img).shape[0] > 50
    ctx.close()


def test_tracked_keyframe_refuses_an_fp32_matcher():
    """with matcher_precision = 2 the temporal pair was dropped behind the fp32 dispatch and the call returned 0 tracks without an error."""
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), lightglue=weights.synthetic_lightglue(1234, n_layers=2),
                      max_batch=2, enc_chunk=2, precision=1, matcher_precision=2)
    left, right = synth.stereo_pair(480, 752, 1000)
    k = ctx.stereo_keyframe(left, right)                     # the plain keyframe works with an fp32 matcher
    assert len(k["idx"]) > 20
    with pytest.raises(api.AirfeError, match="2-byte LightGlue forward|fp16 / bf16"):
        ctx.stereo_keyframe(left, right, track=True, ref_feat=k["featL"])
    ctx.close()


def test_bad_tuning_is_refused():
    with pytest.raises(api.AirfeError, match="lgb_tokens"):
        _ctx(tuning={"lgb_tokens": 7})
    with pytest.raises(TypeError):
        _ctx(tuning={"no_such_switch": 1})


def test_match_lines_after_an_overflowed_relation_stays_inside_its_buffers():
    """with capE too small AssignPointsToLines drops entries and leaves row_ptr unclamped; MatchLines on such a relation must clamp its walks
    (no out-of-bounds read) — frames whose relation fits are unaffected."""
    import torch
    B, CL, K = 2, 256, 400
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), lightglue=weights.synthetic_lightglue(1234),
                      max_batch=B, enc_chunk=2 * B, check_launches=1)
    ls, rs = synth.stereo_batch(B, 480, 752, 1000)
    L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    fl, fr, nl, nr = z(B, K, 259), z(B, K, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
    lines, nlines, junc, njunc = z(2 * B, CL, 4, dt=torch.float64), z(2 * B, dt=torch.int32), z(B, 1024, 259), z(B, dt=torch.int32)
    idx, sc, nm = z(B, K, 2, dt=torch.int32), z(B, K), z(B, dt=torch.int32)
    ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm)
    out = {}
    for capE in (16 * CL, 64):                                # a relation that fits, one that overflows (a frame has several hundred point-on-line entries)
        rel = [dict(rp=z(B, CL + 1, dt=torch.int32), pi=z(B, capE, dt=torch.int32), pd=z(B, capE, dt=torch.float64), tot=z(B, dt=torch.int32)) for _ in range(2)]
        lm = torch.full((B, CL), -7, dtype=torch.int32, device="cuda")
        ctx.assign_points_to_lines_batch_dev(lines[:B], nlines[:B], fl, nl, rel[0]["rp"], rel[0]["pi"], rel[0]["pd"], rel[0]["tot"])
        ctx.assign_points_to_lines_batch_dev(lines[B:], nlines[B:], fr, nr, rel[1]["rp"], rel[1]["pi"], rel[1]["pd"], rel[1]["tot"])
        ctx.match_lines_batch_dev(rel[0]["rp"], rel[0]["pi"], nlines[:B], nl, rel[1]["rp"], rel[1]["pi"], nlines[B:], nr, idx, nm, lm)
        ctx.sync()
        out[capE] = (rel[0]["tot"].cpu().numpy(), lm.cpu().numpy())
    assert (out[16 * CL][0] > 64).all() and (out[64][0] == out[16 * CL][0]).all()          # the overflow is reported through `total`
    n0 = nlines[:B].cpu().numpy()
    assert all(((out[64][1][b, :n0[b]] >= -1) & (out[64][1][b, :n0[b]] < CL)).all() for b in range(B))
    assert sum((out[16 * CL][1][b, :n0[b]] >= 0).sum() for b in range(B)) >= 2
    ctx.close()
