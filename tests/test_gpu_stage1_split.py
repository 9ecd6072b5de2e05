"""PLNet stage 1 on the 2-byte matrix pipe with fp16 (hi, lo) operand pairs (cfg.line_precision = 3, kernels_ext.hip plnet_s1h_kernel) against the f32-input MFMA form
(line_precision = 2) and against the restatement of the real plnet_s1.onnx: the SAME candidate lines, scores within 5e-6, the same kept lines."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN
from gpu_common import diag

pytestmark = pytest.mark.gpu


def _ctx(lp, B=2):
    return api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), lightglue=weights.synthetic_lightglue(1234, n_layers=2),
                       max_batch=B, enc_chunk=2 * B, line_precision=lp, check_launches=1)


@pytest.mark.parametrize("seed", [5, 8, 12, 33])
def test_split_pairs_give_the_fp32_forms_lines(seed):
    from oracle import ref_nets, ref_post
    img = synth.gabor_image(480, 752, seed)
    out = {}
    for lp in (2, 3):
        ctx = _ctx(lp)
        feat, lines, junc = ctx.detect_plnet(img, None, want_junctions=True)
        la, sc = ctx.debug_plnet_s1_last()
        s0 = ctx.debug_plnet_stage0()
        ctx.close()
        out[lp] = (lines, la, sc, junc, s0)
    (l2, la2, sc2, j2, s0), (l3, la3, sc3, j3, _) = out[2], out[3]
    np.testing.assert_array_equal(la2, la3)                         # the same candidates (stage 0 and the wireframe matcher do not depend on the switch)
    err = float(np.abs(sc2 - sc3).max())
    # both against the restatement of the real graph on the device's own stage-0 tensors
    keep, inv, pairs = ref_post.wireframe_matcher(s0["iskeep"], s0["idx_junc_to_end_min"], s0["idx_junc_to_end_max"])
    rla, rsc = ref_nets.plnet_s1_forward(weights.load_pack(os.path.join(GOLDEN, "plnet_s1.airfe")), s0["juncs_pred"], s0["lines_pred"], pairs, inv, keep,
                                         s0["loi_features"][0], s0["loi_features_thin"][0], s0["loi_features_aux"][0])
    e2, e3 = float(np.abs(sc2 - rsc).max()), float(np.abs(sc3 - rsc).max())
    flips = int(((sc2 > 0.75) != (sc3 > 0.75)).sum())
    diag(f"stage1_split_{seed}", candidates=len(sc2), lines_f32=len(l2), lines_split=len(l3), score_diff_max=err, f32_vs_onnx_restatement=e2, split_vs_onnx_restatement=e3,
         threshold_flips=flips)
    assert len(sc2) > 500 and len(l2) >= 100
    assert np.array_equal(la3, rla) and e2 < 5e-5 and e3 < 5e-5      # the gate the fp32 form has always had against the real graph's restatement
    assert err <= 5e-6
    if flips == 0:
        np.testing.assert_array_equal(l2, l3)
        np.testing.assert_array_equal(j2, j3)
    else:                                                           # a candidate within 5e-6 of 0.75: legitimate either way, must be the only difference
        assert flips <= 1 and abs(len(l2) - len(l3)) <= 1


def test_split_pairs_in_the_batched_keyframe_step():
    """8 stereo pairs through airfe_stereo_plnet_batch_dev with line_precision = 3: per image the lines of the batch-1 entry of the same context (the batch and the
    single-image launch share the kernel), and within one line per image of the fp32 form's."""
    import torch
    B = 4
    ls, rs = synth.stereo_batch(B, 480, 752, 1000)
    res = {}
    for lp in (2, 3):
        ctx = _ctx(lp, B)
        L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
        fl, fr, nl, nr = z(B, 400, 259), z(B, 400, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
        lines, nlines, junc, njunc = z(2 * B, 1024, 4, dt=torch.float64), z(2 * B, dt=torch.int32), z(B, 1024, 259), z(B, dt=torch.int32)
        idx, sc, nm = z(B, 400, 2, dt=torch.int32), z(B, 400), z(B, dt=torch.int32)
        ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm)
        ctx.sync()
        res[lp] = (lines.cpu().numpy(), nlines.cpu().numpy())
        if lp == 3:
            for b in range(B):
                _, want, _ = ctx.detect_plnet(ls[b], None, want_junctions=False)
                np.testing.assert_array_equal(res[3][0][b, :res[3][1][b]], want)
        ctx.close()
    assert res[2][1].min() >= 100
    assert np.abs(res[2][1].astype(int) - res[3][1].astype(int)).max() <= 1
    same = sum(int(res[2][1][i] == res[3][1][i] and np.array_equal(res[2][0][i, :res[2][1][i]], res[3][0][i, :res[3][1][i]])) for i in range(2 * B))
    assert same >= 2 * B - 1
