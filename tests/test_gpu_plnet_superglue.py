"""PLNet line path and SuperGlue on the GPU (C ABI) vs the oracle.  The stage-1 LOI head runs the REAL weights of
output/plnet_s1.onnx (tests/golden/plnet_s1.airfe) and is compared with the golden outputs of the real graph."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN
from gpu_common import diag
from oracle import ref_nets, ref_post

pytestmark = pytest.mark.gpu
_C = {}


def _ctx_plnet():
    if "p" not in _C:
        _C["p"] = api.Context(superpoint=weights.synthetic_superpoint(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"),
                              max_batch=2, enc_chunk=2)
    return _C["p"]


@pytest.mark.parametrize("seed,nl", [(5, 300), (6, 1500), (7, 40), (8, 1)])
def test_stage1_real_weights_vs_real_onnx_golden(seed, nl):
    ctx = _ctx_plnet()
    g = np.load(os.path.join(GOLDEN, "plnet_s1_golden.npz"))
    s0 = synth.plnet_stage0_lines(seed, n_lines=nl)
    la, sc = ctx.debug_plnet_s1(s0)
    ref_la, ref_sc = g[f"s{seed}_n{nl}_lines_adjusted"], g[f"s{seed}_n{nl}_scores_line"]
    diag(f"plnet_s1_{seed}_{nl}", m2_dev=la.shape[0], m2_ref=ref_la.shape[0],
         score_err=float(np.abs(sc - ref_sc).max()) if la.shape[0] == ref_la.shape[0] else -1.0)
    assert la.shape == ref_la.shape
    np.testing.assert_array_equal(la, ref_la)                      # wireframe dedup + junction gather: index work, exact
    # fp32 LOI pooling + 496-long fp32 dot products (fmaf order differs from the interpreter's BLAS) + softmax
    np.testing.assert_allclose(sc, ref_sc, atol=5e-5, rtol=0)


def test_no_kept_lines():
    ctx = _ctx_plnet()
    s0 = synth.plnet_stage0_lines(9, n_lines=1)
    s0["iskeep"][:] = 0
    la, sc = ctx.debug_plnet_s1(s0)
    assert la.shape == (0, 4) and sc.shape == (0,)


@pytest.mark.parametrize("seed,nl,lt,ll", [(5, 300, 0.75, 50.0), (6, 1500, 0.5, 10.0), (11, 3000, 0.2, 0.0)])
def test_plnet_infer_lines_and_junctions(seed, nl, lt, ll):
    ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"),
                      max_batch=2, enc_chunk=2, line_threshold=lt, line_length_threshold=ll)
    img = synth.gabor_image(480, 752, seed)
    s0 = synth.plnet_stage0_lines(seed, n_lines=nl)
    feat, lines, junc = ctx.detect_plnet(img, s0, want_junctions=True)
    heat, nms, desc = ctx.detector_maps(1)
    ws, hs = np.float32(752 / 512), np.float32(480 / 512)
    # oracle on the device's own stage-1 outputs (index / byte work must be exact)
    la, sc = ctx.debug_plnet_s1(s0)
    ref_lines, jmap = ref_post.line_filter(la, sc, 4, lt, ll)
    ref_lines = ref_post.rescale_lines(ref_lines, ws, hs)
    ref_junc = ref_post.junction_detector(nms[0], np.ascontiguousarray(desc[0].transpose(2, 0, 1)), jmap, 4, ws, hs)
    diag(f"plnet_lines_{seed}", n_lines=lines.shape[0], n_ref=ref_lines.shape[0], n_junc=junc.shape[0], n_junc_ref=ref_junc.shape[0])
    assert lines.shape == ref_lines.shape and lines.shape[0] > 0
    np.testing.assert_array_equal(lines, ref_lines)
    assert junc.shape == ref_junc.shape and junc.shape[0] > 0
    np.testing.assert_array_equal(junc[:, :3], ref_junc[:, :3])
    np.testing.assert_allclose(junc[:, 3:], ref_junc[:, 3:], atol=2e-6, rtol=0)
    # the point branch is the same detector
    np.testing.assert_array_equal(feat, ctx.detect_points(img))
    # lines are appended, never cleared (plnet.cpp:544)
    det = api.FeatureDetector(ctx)
    acc = [(0.0, 0.0, 1.0, 1.0)]
    ok, f, j = det.DetectLines(img, s0, acc, junction_detection=False)
    assert ok and len(acc) == 1 + lines.shape[0] and j.shape == (259, 0)
    ctx.close()


def _sg_pair(n0, n1, seed, w=752, h=480):
    from planted import normalised, planted_pair
    f0, f1 = planted_pair(n0, n1, seed, w, h)
    return f0, f1, normalised(f0, w, h, 0.7), normalised(f1, w, h, 0.7)


def _check_superglue(name, ctx, w, f0, f1, layers, iters, tol, min_valid, ref=None):
    """`ref` = the [N0+1, N1+1] matrix to compare with (tests/test_gpu_hf_pin.py passes Hugging Face's); default: the fp32 oracle's."""
    z = ctx.superglue_scores(f0, f1)
    if ref is None:
        ref = ref_nets.superglue_forward(w, f0[:, 1:3], f0[:, 0], f0[:, 3:], f1[:, 1:3], f1[:, 0], f1[:, 3:], n_layers=layers, iters=iters)
    n0, n1 = f0.shape[0], f1.shape[0]
    err = np.abs(z - ref)
    i0, i1, m0, m1 = ctx.match_superglue(f0, f1)
    d0, d1, dm0, dm1 = ref_post.superglue_decode(z, 0.2)          # decode on the DEVICE scores: exact index work
    r0, r1, rm0, rm1 = ref_post.superglue_decode(ref, 0.2)        # ... and the oracle's own decision
    # rows the oracle decides within the tolerance: mutual maximum within tol of log(0.2), or a runner-up within 2 tol
    inner = ref[:n0, :n1].astype(np.float64)
    srt_r = np.sort(inner, 1); srt_c = np.sort(inner, 0)
    rgap = srt_r[:, -1] - srt_r[:, -2] if n1 > 1 else np.full(n0, np.inf)
    cgap = srt_c[-1] - srt_c[-2] if n0 > 1 else np.full(n1, np.inf)
    am = inner.argmax(1)
    frag = {i for i in range(n0) if abs(inner[i, am[i]] - np.log(0.2)) <= tol or
            (inner[i, am[i]] > np.log(0.2) - tol and (rgap[i] <= 2 * tol or cgap[am[i]] <= 2 * tol))}
    keep = np.array([i not in frag for i in range(n0)])
    diag(name, max_err=err.max(), mean_err=err.mean(), ref_absmax=np.abs(ref).max(), n_valid=int((i0 >= 0).sum()),
         n_valid_ref=int((r0 >= 0).sum()), fragile=len(frag), nan=int(np.isnan(z).sum()))
    assert not np.isnan(z).any()
    np.testing.assert_array_equal(i0, d0)
    np.testing.assert_array_equal(i1, d1)
    np.testing.assert_allclose(m0, dm0, rtol=2e-6)
    np.testing.assert_allclose(m1, dm1, rtol=2e-6)
    v = np.nonzero(r0 >= 0)[0]
    if v.size:
        assert err[v, r0[v]].max() <= tol                          # the entries that decide the matches
    assert err.max() <= 0.05 * max(1.0, np.abs(ref).mean())
    assert int((r0 >= 0).sum()) >= min_valid, "the planted correspondences must be valid matches in the oracle"
    assert len(frag) <= max(2, n0 // 50)
    np.testing.assert_array_equal(i0[keep], r0[keep])             # identical indices0 vs the fp32 oracle
    return z


@pytest.mark.parametrize("n0,n1,layers,iters,min_valid,fused", [(300, 280, 4, 20, 100, 0), (64, 65, 18, 100, 20, 0), (1, 3, 2, 5, 0, 0),
                                                                 (400, 400, 18, 100, 150, 0), (400, 400, 18, 100, 150, 1),
                                                                 (300, 280, 4, 20, 100, 1), (1, 3, 2, 5, 0, 1)])
def test_superglue_vs_oracle(n0, n1, layers, iters, min_valid, fused):
    # fused = the propagation block (merge, mlp.0 + ReLU, mlp.3, residual) as the one-kernel form large batches use
    w = weights.synthetic_superglue(1234, n_layers=layers)
    # ... and the keypoint encoder's large layers as GEMMs (the large-batch form)
    ctx = api.Context(superglue=w, matcher=1, max_batch=2, sinkhorn_iters=iters, tuning={"fuse_lg_block": fused, "sg_kenc_gemm": fused}, check_launches=1)
    _, _, f0, f1 = _sg_pair(n0, n1, n0 * 3 + n1)
    z = _check_superglue(f"sg_{n0}_{n1}_{layers}_{'fused' if fused else 'split'}", ctx, w, f0, f1, layers, iters, 0.05, min_valid)
    # marginals: exp(Z) rows/cols sum to the prescribed masses after `iters` iterations (column step is last)
    p = np.exp(z.astype(np.float64))
    np.testing.assert_allclose(p[:, :n1].sum(0), 1.0, atol=2e-3)
    ctx.close()


@pytest.mark.parametrize("fused", [1, 0])
def test_superglue_merge_folded_into_mlp0_gives_the_bits_of_the_identity_form(fused):
    """airfe_tuning::fold_out_proj for SuperGlue: attn.merge (whose input channels the loader permutes to the head-major order of the attention output) multiplied
    into the message half of mlp.0.  Same pin as tests/test_gpu_lightglue.py: the Python fold with merge = identity, run WITH the merge GEMM, must give the bits
    of the library's fold on the original pack — which holds only if the loader applied the channel permutation to the folded columns correctly."""
    w = weights.synthetic_superglue(1234, n_layers=4)
    _, _, na, nb = _sg_pair(300, 280, 77)
    zs = []
    for pack, fold in ((w, 1), (weights.fold_out_proj(w), 0)):
        ctx = api.Context(superglue=pack, matcher=1, max_batch=2, sinkhorn_iters=20, tuning={"fuse_lg_block": fused, "sg_kenc_gemm": fused, "fold_out_proj": fold}, check_launches=1)
        zs.append(ctx.superglue_scores(na, nb).copy())
        ctx.close()
    assert np.isfinite(zs[0]).all()
    np.testing.assert_array_equal(zs[0], zs[1])


@pytest.mark.parametrize("name", ["ties_threshold_inf_row", "all_minus_inf", "all_equal_below", "all_equal_above",
                                  "random_with_floor_nan", "single", "one_row"])
def test_decode_kernels_on_hand_built_scores(name):
    """sg_rowmax / sg_colmax / sg_decode straight on score matrices with ties, -inf rows, -FLT_MAX and NaN entries."""
    from test_structured_weights_cpu import hand_built_score_matrices
    if "dec" not in _C:
        _C["dec"] = api.Context(superglue=weights.synthetic_superglue(1234, n_layers=2), matcher=1, max_batch=2)
    ctx = _C["dec"]
    s = hand_built_score_matrices()[name]
    z = np.full((s.shape[0] + 1, s.shape[1] + 1), 5.0, np.float32)     # dustbins must be ignored
    z[:-1, :-1] = s
    got = ctx.debug_sg_decode(z)
    want = ref_post.superglue_decode(z, 0.2)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_allclose(got[2], want[2], rtol=2e-6)
    np.testing.assert_allclose(got[3], want[3], rtol=2e-6)


def test_matching_points_superglue_branch():
    w = weights.synthetic_superglue(1234, n_layers=2)
    ctx = api.Context(superglue=w, matcher=1, max_batch=2, sinkhorn_iters=20)
    pm = api.PointMatcher(ctx, 752, 480, 1)
    a, b, _, _ = _sg_pair(50, 60, 1)
    cnt, matches = pm.MatchingPoints(np.asfortranarray(a.T), np.asfortranarray(b.T))
    na, nb = ref_post.normalize_keypoints(a, 752, 480, 0.7), ref_post.normalize_keypoints(b, 752, 480, 0.7)
    z = ctx.superglue_scores(na, nb)
    ref = ref_post.superglue_matches(*ref_post.superglue_decode(z, 0.2))
    assert cnt == len(ref) and cnt >= 15
    assert [(m[0], m[1]) for m in matches] == [(r[0], r[1]) for r in ref]
    ctx.close()


def test_superglue_cfg5_max_size_fp16():
    """1280x720, SuperGlue-outdoor shape, N at the engine profile maximum 1024 (super_glue.cpp:55), fp16,
    100 Sinkhorn iterations."""
    w = weights.synthetic_superglue(1234)
    ctx = api.Context(superglue=w, matcher=1, max_batch=2, sinkhorn_iters=100, max_keypoints=1024, precision=1,
                      image_width=1280, image_height=720)
    _, _, f0, f1 = _sg_pair(1024, 1000, 501, 1280, 720)
    z = _check_superglue("sg_cfg5_1024_1000_fp16", ctx, w, f0, f1, 18, 100, 0.05, 400)
    assert z.shape == (1025, 1001)
    i0, i1, m0, m1 = ctx.match_superglue(f0, f1)
    assert i0.shape == (1024,) and i1.shape == (1000,)            # lengths h-1, w-1: super_glue.cpp:357-358
    ctx.close()
