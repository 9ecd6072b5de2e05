"""LightGlue parity on the GPU (C ABI) vs the CPU oracle; filter_matches is checked bit-exact on the device's
own score matrix."""
import numpy as np
import pytest

from gpu_common import context, diag
from oracle import ref_nets, ref_post

pytestmark = pytest.mark.gpu


def _features(n, seed, w=752, h=480):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 256)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xy = np.stack([rng.uniform(4, w - 4, n), rng.uniform(4, h - 4, n)], 1).astype(np.float32)
    f = np.zeros((n, 259), np.float32)
    f[:, 0] = rng.uniform(0.01, 1, n)
    f[:, 1:3] = xy
    f[:, 3:] = d
    return f


def _pair(n0, n1, seed):
    f0 = _features(n0, seed)
    f1 = _features(n1, seed + 1)
    k = min(n0, n1) // 2                      # plant true correspondences so that matches exist
    rng = np.random.default_rng(seed + 2)
    f1[:k, 3:] = f0[:k, 3:] + 0.05 * rng.normal(size=(k, 256)).astype(np.float32)
    f1[:k, 3:] /= np.linalg.norm(f1[:k, 3:], axis=1, keepdims=True)
    f1[:k, 1] = f0[:k, 1] - 12
    f1[:k, 2] = f0[:k, 2]
    n0f = ref_post.normalize_keypoints(f0, 752, 480, 0.5)
    n1f = ref_post.normalize_keypoints(f1, 752, 480, 0.5)
    return f0, f1, np.ascontiguousarray(n0f[:, 1:]), np.ascontiguousarray(n1f[:, 1:])


# A context picks the LightGlue block form by token count (fused lg_blockf_kernel from 3200 tokens, four launches below);
# AIRFE_FUSE_LG_BLOCK forces either, so that both forms meet the oracle at every size.
FORMS = [{"AIRFE_FUSE_LG_BLOCK": "1"}, {"AIRFE_FUSE_LG_BLOCK": "0"}]


@pytest.mark.parametrize("env", FORMS, ids=["fused_block", "four_launches"])
@pytest.mark.parametrize("n0,n1", [(400, 400), (317, 400), (64, 65), (1, 5), (2, 1)])
def test_lightglue_scores_vs_oracle(n0, n1, env):
    ctx, _, lg = context("lg", env=env, max_batch=4)
    _, _, a, b = _pair(n0, n1, n0 * 3 + n1)
    s = ctx.lightglue_scores(a, b)
    ref = ref_nets.lightglue_forward(lg, a[:, :2], a[:, 2:], b[:, :2], b[:, 2:])
    err = np.abs(s - ref)
    idx, sc = ctx.match_lightglue(a, b)
    ridx, rsc = ref_post.filter_matches(ref, 0.1)
    # filter_matches on the DEVICE scores must be reproduced exactly (index work)
    didx, dsc = ref_post.filter_matches(s, 0.1)
    agree = len(set(map(tuple, idx)) & set(map(tuple, ridx))) / max(len(ridx), 1)
    diag(f"lg_scores_{n0}_{n1}_{'fused' if env['AIRFE_FUSE_LG_BLOCK'] == '1' else 'split'}", max_err=err.max(), mean_err=err.mean(), ref_absmax=np.abs(ref).max(), n_dev=len(idx),
         n_ref=len(ridx), match_agreement=agree, nan=int(np.isnan(s).sum()))
    assert not np.isnan(s).any()
    np.testing.assert_array_equal(idx, didx)
    np.testing.assert_allclose(sc, dsc, rtol=2e-6)
    assert err.max() <= 0.05 * max(1.0, np.abs(ref).mean()), "log-assignment scores drifted from the fp32 oracle"
    if len(ridx) >= 10:
        assert agree >= 0.9


def test_lightglue_layer_states_drift():
    """Where precision goes: score error with 1, 3, 9 layers (diagnostic, loose bound)."""
    from airslam_amd import weights
    from airslam_amd import api
    _, _, a, b = _pair(200, 180, 77)
    out = {}
    for L in (1, 3):
        w = weights.synthetic_lightglue(1234, n_layers=L)
        ctx = api.Context(lightglue=w, max_batch=2)
        s = ctx.lightglue_scores(a, b)
        ref = ref_nets.lightglue_forward(w, a[:, :2], a[:, 2:], b[:, :2], b[:, 2:], n_layers=L)
        out[f"L{L}_max_err"] = float(np.abs(s - ref).max())
        ctx.close()
    diag("lg_layers", **out)
    assert out["L1_max_err"] < 0.05


def test_match_batch_dev_equals_host_path():
    import torch
    from airslam_amd import api
    ctx, _, lg = context("lg", max_batch=4)
    pairs = [_pair(400, 380, 5), _pair(120, 400, 9)]
    f0 = torch.zeros((2, 400, 259)); f1 = torch.zeros((2, 400, 259))
    n0 = torch.tensor([400, 120], dtype=torch.int32); n1 = torch.tensor([380, 400], dtype=torch.int32)
    for i, (a, b, _, _) in enumerate(pairs):
        f0[i, :a.shape[0]] = torch.from_numpy(a); f1[i, :b.shape[0]] = torch.from_numpy(b)
    f0, f1, n0, n1 = f0.cuda(), f1.cuda(), n0.cuda(), n1.cuda()
    idx = torch.zeros((2, 400, 2), dtype=torch.int32, device="cuda")
    sc = torch.zeros((2, 400), dtype=torch.float32, device="cuda")
    nm = torch.zeros((2,), dtype=torch.int32, device="cuda")
    ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm)
    ctx.sync()
    pm = api.PointMatcher(ctx, 752, 480, 0)
    for i, (a, b, _, _) in enumerate(pairs):
        cnt, matches = pm.MatchingPoints(np.asfortranarray(a.T), np.asfortranarray(b.T))
        k = int(nm[i])
        assert k == cnt
        got = idx[i, :k].cpu().numpy()
        assert [tuple(g) for g in got] == [(m[0], m[1]) for m in matches]
        np.testing.assert_allclose(1.0 - sc[i, :k].cpu().numpy(), [m[2] for m in matches], atol=1e-6)


def test_matching_points_early_out():
    from airslam_amd import api
    ctx, _, _ = context("lg", max_batch=4)
    pm = api.PointMatcher(ctx, 752, 480, 0)
    assert pm.MatchingPoints(np.zeros((259, 0), np.float32), np.zeros((259, 7), np.float32)) == (0, [])


# which kernel the K = 256 projections of an 8-pair batch (6400 tokens) go through: the thresholds are lowered so that the
# large-batch kernels are exercised at a size the oracle-free comparison below finishes quickly
# (gemmr with 24 persistent workgroups instead of 256: 200 token tiles / 8 per feature group = 25 tiles per workgroup, so the
# 6- and 8-slot DMA rings wrap several times, as they do at 64 pairs on 256 workgroups)
BIG_GEMMS = {"gemm8": {"AIRFE_GEMM8_MIN_M": "4096", "AIRFE_GEMMR_MIN_M": "1000000000"},
             "gemmr": {"AIRFE_GEMM8_MIN_M": "4096", "AIRFE_GEMMR_MIN_M": "1024"},
             "gemmr_ring_wrap": {"AIRFE_GEMM8_MIN_M": "4096", "AIRFE_GEMMR_MIN_M": "1024", "AIRFE_GEMMR_WGS": "24"},
             "gemmr_two_launches": {"AIRFE_GEMM8_MIN_M": "4096", "AIRFE_GEMMR_MIN_M": "1024", "AIRFE_GEMMR_WGS": "24", "AIRFE_QKV_PAIR": "0"}}


@pytest.mark.parametrize("big", list(BIG_GEMMS))
@pytest.mark.parametrize("env", FORMS, ids=["fused_block", "four_launches"])
def test_large_batch_uses_gemm8_and_agrees_with_single_pair_path(env, big):
    """8 pairs -> M = 16 x 400 = 6400 rows: the linears go through the 8-wave LDS-DMA GEMM (kernels_gemm8.hip) or, where it
    applies (K = 256, no rotary), the register-resident streaming GEMM (kernels_gemmr.hip); a single pair (M = 896) goes
    through gemm_small_kernel.  Same K-order accumulation => same matches, with the block form held fixed (it is what
    changes the rounding points)."""
    import torch
    from airslam_amd import api
    ctx, _, lg = context("lg", env=dict(env, **BIG_GEMMS[big]), max_batch=8)
    B = 8
    pairs = [_pair(400 - 7 * i, 390 - 11 * i, 40 + i) for i in range(B)]
    f0 = torch.zeros((B, 400, 259)); f1 = torch.zeros((B, 400, 259))
    n0 = torch.tensor([p[0].shape[0] for p in pairs], dtype=torch.int32)
    n1 = torch.tensor([p[1].shape[0] for p in pairs], dtype=torch.int32)
    for i, (a, b, _, _) in enumerate(pairs):
        f0[i, :a.shape[0]] = torch.from_numpy(a); f1[i, :b.shape[0]] = torch.from_numpy(b)
    f0, f1, n0, n1 = f0.cuda(), f1.cuda(), n0.cuda(), n1.cuda()
    idx = torch.zeros((B, 400, 2), dtype=torch.int32, device="cuda")
    sc = torch.zeros((B, 400), dtype=torch.float32, device="cuda")
    nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm)
    ctx.sync()
    pm = api.PointMatcher(ctx, 752, 480, 0)
    for i, (a, b, _, _) in enumerate(pairs):
        cnt, matches = pm.MatchingPoints(np.asfortranarray(a.T), np.asfortranarray(b.T))
        k = int(nm[i])
        assert k == cnt
        assert [tuple(g) for g in idx[i, :k].cpu().numpy()] == [(m[0], m[1]) for m in matches]
        np.testing.assert_allclose(1.0 - sc[i, :k].cpu().numpy(), [m[2] for m in matches], atol=1e-5)


def test_block_form_switch_by_token_count_keeps_the_matches():
    """Switch point lowered to 4096 tokens: 1 pair (800 tokens) runs the four-launch block through gemm_small_kernel,
    8 pairs (6400 tokens) the fused kernel + gemm_kernel.  The two block forms round at different points, so scores agree
    to bf16 noise and the match sets almost entirely."""
    import torch
    from airslam_amd import api
    ctx, _, lg = context("lg", env={"AIRFE_BLOCK_MIN_M": "4096"}, max_batch=8)
    B = 8
    pairs = [_pair(400 - 7 * i, 390 - 11 * i, 140 + i) for i in range(B)]
    f0 = torch.zeros((B, 400, 259)); f1 = torch.zeros((B, 400, 259))
    n0 = torch.tensor([p[0].shape[0] for p in pairs], dtype=torch.int32)
    n1 = torch.tensor([p[1].shape[0] for p in pairs], dtype=torch.int32)
    for i, (a, b, _, _) in enumerate(pairs):
        f0[i, :a.shape[0]] = torch.from_numpy(a); f1[i, :b.shape[0]] = torch.from_numpy(b)
    f0, f1, n0, n1 = f0.cuda(), f1.cuda(), n0.cuda(), n1.cuda()
    idx = torch.zeros((B, 400, 2), dtype=torch.int32, device="cuda")
    sc = torch.zeros((B, 400), dtype=torch.float32, device="cuda")
    nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
    ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm)
    ctx.sync()
    pm = api.PointMatcher(ctx, 752, 480, 0)
    agree = []
    for i, (a, b, _, _) in enumerate(pairs):
        cnt, matches = pm.MatchingPoints(np.asfortranarray(a.T), np.asfortranarray(b.T))
        batch = {tuple(g) for g in idx[i, :int(nm[i])].cpu().numpy()}
        single = {(m[0], m[1]) for m in matches}
        agree.append(len(batch & single) / max(len(batch | single), 1))
    diag("lg_block_form_switch", min_agreement=min(agree), mean_agreement=float(np.mean(agree)))
    assert min(agree) >= 0.95


@pytest.mark.parametrize("n0,n1", [(1024, 1024), (1024, 777)])
def test_lightglue_at_the_profile_maximum(n0, n1):
    """N = 1024 is the engine's optimisation-profile maximum (light_glue.cpp:52); the arena is sized by max_keypoints."""
    from airslam_amd import api, weights
    lg = weights.synthetic_lightglue(1234)
    ctx = api.Context(lightglue=lg, max_batch=2, max_keypoints=1024)
    _, _, a, b = _pair(n0, n1, 1024 + n1)
    s = ctx.lightglue_scores(a, b)
    ref = ref_nets.lightglue_forward(lg, a[:, :2], a[:, 2:], b[:, :2], b[:, 2:])
    err = np.abs(s - ref)
    idx, sc = ctx.match_lightglue(a, b)
    didx, dsc = ref_post.filter_matches(s, 0.1)
    diag(f"lg_max_{n0}_{n1}", max_err=err.max(), mean_err=err.mean(), ref_absmax=np.abs(ref).max(), n_dev=len(idx),
         nan=int(np.isnan(s).sum()))
    assert s.shape == (n0, n1) and not np.isnan(s).any()
    np.testing.assert_array_equal(idx, didx)
    np.testing.assert_allclose(sc, dsc, rtol=2e-6)
    assert err.max() <= 0.05 * max(1.0, np.abs(ref).mean())
    ctx.close()


def test_bench_size_matcher_batch_repeats_its_distinct_pairs():
    """64 pairs per call (the bench workload: 51200 tokens, the gemmr rings wrap 13 times, 400 block tiles on 256 CUs) = 4 distinct
    planted pairs x 16: every copy must equal its original, and both must equal the 4-pair call of a small context, bit for bit."""
    import torch
    big, _, _ = context("lg", max_batch=128)
    small, _, _ = context("lg", max_batch=8)
    pairs = [_pair(400 - 9 * i, 397 - 13 * i, 300 + 7 * i) for i in range(4)]

    def run(ctx, reps):
        B = 4 * reps
        f0 = torch.zeros((B, 400, 259)); f1 = torch.zeros((B, 400, 259))
        n0 = torch.zeros((B,), dtype=torch.int32); n1 = torch.zeros((B,), dtype=torch.int32)
        for i in range(B):
            a, b, _, _ = pairs[i % 4]
            f0[i, :a.shape[0]] = torch.from_numpy(a); f1[i, :b.shape[0]] = torch.from_numpy(b)
            n0[i] = a.shape[0]; n1[i] = b.shape[0]
        f0, f1, n0, n1 = f0.cuda(), f1.cuda(), n0.cuda(), n1.cuda()
        idx = torch.zeros((B, 400, 2), dtype=torch.int32, device="cuda")
        sc = torch.zeros((B, 400), dtype=torch.float32, device="cuda")
        nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
        ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm)
        ctx.sync()
        return idx.cpu().numpy(), sc.cpu().numpy(), nm.cpu().numpy()

    ridx, rsc, rnm = run(small, 1)
    oidx, osc, onm = run(big, 16)
    diag("lg_bench_size_repeats", matches=str(rnm.tolist()))
    assert rnm.sum() >= 4                           # the planted correspondences do produce matches
    for i in range(64):
        j, m = i % 4, int(rnm[i % 4])
        assert int(onm[i]) == m
        np.testing.assert_array_equal(oidx[i, :m], ridx[j, :m])
        np.testing.assert_array_equal(osc[i, :m], rsc[j, :m])


@pytest.mark.parametrize("env", [{"AIRFE_FUSE_LG_BLOCK": "1", "AIRFE_GEMMR_MIN_M": "512"}, {"AIRFE_FUSE_LG_BLOCK": "0"}],
                         ids=["fused_block_gemmr", "four_launches"])
def test_lightglue_fp16_storage(env):
    """precision = 1 (fp16 operands, fp32 accumulate) through the same kernels: the PF16 instantiations of lg_blockf_kernel,
    gemmr_kernel / gemmr_pair_kernel and gemm_small_kernel.  fp16 keeps 3 more mantissa bits than bf16, so the scores must sit
    closer to the fp32 oracle than the bf16 bound of the tests above."""
    ctx, _, lg = context("lg", env=env, max_batch=4, precision=1)
    _, _, a, b = _pair(400, 371, 4242)
    s = ctx.lightglue_scores(a, b)
    ref = ref_nets.lightglue_forward(lg, a[:, :2], a[:, 2:], b[:, :2], b[:, 2:])
    err = np.abs(s - ref)
    idx, sc = ctx.match_lightglue(a, b)
    didx, dsc = ref_post.filter_matches(s, 0.1)
    diag(f"lg_fp16_{'fused' if env['AIRFE_FUSE_LG_BLOCK'] == '1' else 'split'}", max_err=err.max(), mean_err=err.mean(),
         ref_absmax=np.abs(ref).max(), n_dev=len(idx))
    assert not np.isnan(s).any()
    np.testing.assert_array_equal(idx, didx)
    np.testing.assert_allclose(sc, dsc, rtol=2e-6)
    assert err.max() <= 0.02 * max(1.0, np.abs(ref).mean()), "fp16 scores drifted from the fp32 oracle"
