"""The matcher's flash attention alone (kernels_attn.hip through airfe_debug_attention) against a float64 soft-max attention on the SAME 2-byte inputs: ragged lengths, the
32-key tail sub-tile, cross attention — and logits built to drive the kernel's RE-CENTRING path (its running shift is stale by design: a later tile whose partial row sums
leave the 2-byte range is recomputed with a fresh row maximum), which the LightGlue / SuperGlue parity tests never reach (0 re-centred tiles in 5 forwards of the bench
workload, profiles/r06_att_attention.txt)."""
import numpy as np
import pytest

from airslam_amd import api, weights

pytestmark = pytest.mark.gpu
_C = {}


def _ctx():
    if "c" not in _C:
        _C["c"] = api.Context(lightglue=weights.synthetic_lightglue(1234, n_layers=1), max_batch=4, max_keypoints=400)
    return _C["c"]


def _half(x):
    return x.astype(np.float16).astype(np.float64)


def _reference(q, k, v, lens, cross):
    """p = 2^(q.k - max) over the valid keys, out = sum p v / sum p; inputs rounded to fp16 like the kernel's operands (P itself is NOT rounded here: the tolerance covers it)"""
    S, H, n, _ = q.shape
    out = np.zeros((S, n, H * 64))
    q, k, v = _half(q), _half(k), _half(v)
    for s in range(S):
        skv = s ^ 1 if cross else s
        lq, lk = int(lens[s]), int(lens[skv])
        for h in range(H):
            sc = q[s, h, :lq] @ k[skv, h, :lk].T
            p = np.exp2(sc - sc.max(1, keepdims=True))
            out[s, :lq, h * 64:(h + 1) * 64] = (p @ v[skv, h, :lk]) / p.sum(1, keepdims=True)
    return out


def _check(name, q, k, v, lens, cross, tol):
    got = _ctx().debug_attention(q, k, v, lens, cross=cross)
    ref = _reference(q, k, v, lens, cross)
    assert np.isfinite(got).all(), name
    for s in range(q.shape[0]):
        err = np.abs(got[s, :lens[s]] - ref[s, :lens[s]]).max()
        assert err <= tol * max(1.0, np.abs(ref[s, :lens[s]]).max()), (name, s, err)
    return got


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("lens", [(400, 400), (400, 317), (33, 400), (1, 64), (65, 97)])
def test_attention_vs_float64_softmax(lens, cross):
    rng = np.random.default_rng(sum(lens) + int(cross))
    S, H, n = 2, 4, 400
    q = rng.standard_normal((S, H, n, 64)) * 0.6           # logits ~ N(0, 0.36 * 64 = 23): a range of +-20 in log2 units, like a trained layer's sharpest heads
    k = rng.standard_normal((S, H, n, 64)) * 0.6
    v = rng.standard_normal((S, H, n, 64))
    _check(f"attn_{lens}_{cross}", q, k, v, np.array(lens, np.int32), cross, 4e-3)


@pytest.mark.parametrize("cross", [False, True])
def test_attention_recentres_when_later_tiles_dominate(cross):
    """Keys sorted so that the row maximum GROWS from tile to tile by far more than the 2^14 a partial row sum may reach under the stale shift: every tile behind the first
    must take the re-centring path (accumulators rescaled, shift updated), and early keys must underflow to exactly the weight float64 gives them (~0)."""
    rng = np.random.default_rng(5 + int(cross))
    S, H, n = 2, 4, 400
    q = np.zeros((S, H, n, 64)); k = np.zeros((S, H, n, 64))
    q[..., 0] = 1.0                                          # logit of (query i, key j) = k[j, 0] + noise
    k[..., 0] = np.linspace(-60.0, 60.0, n)[None, None, :]   # +19 per 64-key tile in log2 units: 2^19 >> 2^14
    q[..., 1:] = rng.standard_normal((S, H, n, 63)) * 0.2
    k[..., 1:] = rng.standard_normal((S, H, n, 63)) * 0.2
    v = rng.standard_normal((S, H, n, 64))
    lens = np.array([400, 389], np.int32)
    got = _check(f"attn_recentre_{cross}", q, k, v, lens, cross, 4e-3)
    # the answer is dominated by the last few keys: a kernel that skipped the re-centring would return inf / NaN or the first tile's average
    ref = _reference(q, k, v, lens, cross)
    assert np.abs(got[0, :lens[0]] - ref[0, :lens[0]]).max() < 0.02 and np.abs(ref[0, :10]).max() > 0.1


def test_attention_recentres_on_an_isolated_spike():
    """one key in the LAST tile beats everything before it by 2^40 for half of the queries only: lanes of one wave disagree about the need to re-centre (the kernel decides per
    wave with __any), rows without the spike must come out unchanged"""
    rng = np.random.default_rng(11)
    S, H, n = 2, 4, 400
    q = rng.standard_normal((S, H, n, 64)) * 0.3
    k = rng.standard_normal((S, H, n, 64)) * 0.3
    v = rng.standard_normal((S, H, n, 64))
    q[:, :, ::2, 7] = 8.0
    k[:, :, :, 7] = 0.0
    k[:, :, 390, 7] = 5.0                                    # +40 for the even queries at key 390
    lens = np.array([400, 400], np.int32)
    got = _check("attn_spike", q, k, v, lens, False, 4e-3)
    vh = v.astype(np.float16).astype(np.float64)
    assert np.abs(got[0, 0, :64] - vh[0, 0, 390]).max() < 2e-3       # an even query returns (almost exactly) the spike key's value row
