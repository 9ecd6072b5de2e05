"""PLNet over a device-resident BATCH (airfe_detect_plnet_batch_dev / airfe_stereo_plnet_batch_dev; no reference counterpart: PLNet::infer
is one image per call).  The batched launches are the batch-1 kernels with one image per grid row, so the gate is EXACT equality with
airfe_detect_plnet image by image — which tests/test_gpu_plnet_s0.py, test_gpu_plnet_superglue.py and the golden stage-1 tests pin against
the oracle and the real plnet_s1.onnx."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN
from gpu_common import diag

pytestmark = pytest.mark.gpu
_C = {}
CAP_L, CAP_J = 2048, 1024


def _ctx(lightglue=False, **kw):
    key = (lightglue,) + tuple(sorted(kw.items()))
    if key not in _C:
        _C[key] = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"),
                              lightglue=weights.synthetic_lightglue(1234) if lightglue else None, max_batch=8, enc_chunk=4, **kw)
    return _C[key]


def _buffers(torch, B, J, cap_l=CAP_L, cap_j=CAP_J):
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    return dict(feat=z(B, 400, 259), n=z(B, dt=torch.int32), lines=z(B, cap_l, 4, dt=torch.float64), nlines=z(B, dt=torch.int32),
                junc=z(max(J, 1), cap_j, 259), njunc=z(max(J, 1), dt=torch.int32), found=z(B + J, dt=torch.int32))


def _images(B, seed0):
    return np.stack([synth.gabor_image(480, 752, seed0 + 3 * i) for i in range(B)])


@pytest.mark.parametrize("B,J", [(5, 3), (1, 1), (8, 0), (7, 7)])
def test_plnet_batch_equals_single_image_calls(B, J):
    import torch
    ctx = _ctx()
    imgs = _images(B, 8)
    o = _buffers(torch, B, J)
    ctx.detect_plnet_batch_dev(torch.from_numpy(imgs).cuda(), o["feat"], o["n"], o["lines"], o["nlines"], o["junc"] if J else None,
                               o["njunc"] if J else None, o["found"])
    ctx.sync()
    n, nl, nj, found = (o[k].cpu().numpy() for k in ("n", "nlines", "njunc", "found"))
    counts = []
    for b in range(B):
        feat, lines, junc = ctx.detect_plnet(imgs[b], None, want_junctions=b < J)
        np.testing.assert_array_equal(o["feat"][b, :n[b]].cpu().numpy(), feat)
        np.testing.assert_array_equal(o["lines"][b, :nl[b]].cpu().numpy(), lines)
        assert n[b] == feat.shape[0] and nl[b] == lines.shape[0] and found[b] == lines.shape[0]
        if b < J:
            assert nj[b] == junc.shape[0] and found[B + b] == junc.shape[0]
            np.testing.assert_array_equal(o["junc"][b, :nj[b]].cpu().numpy(), junc)
        counts.append((int(n[b]), int(nl[b]), int(nj[b]) if b < J else -1))
    diag(f"plnet_batch_{B}_{J}", counts=str(counts))
    assert min(c[1] for c in counts) >= 100, "lines must exist for this comparison to mean anything"
    if J:
        assert min(c[2] for c in counts[:J]) >= 50


@pytest.mark.parametrize("wgs", [256, 8], ids=["one_tile_per_workgroup", "ring_wraps"])
def test_loi_head_streaming_gather_gives_the_bits_of_the_tiled_kernel(wgs):
    """Round 6: the LOI head at the junctions' tap rows (K = N = 128, 1200 gathered rows per image) streams through gemmr_gather128_kernel from 8192 rows on
    (kernels_gemmr.hip); airfe_tuning::desc_gather_stream = 0 keeps the tiled gemm8 kernel.  Same fragments, same K order, bias after the sum: every line of 8
    images byte-equal — also with 8 persistent workgroups, where each streams 19 tiles through its 8-slot ring."""
    import torch
    imgs = torch.from_numpy(_images(8, 21)).cuda()
    outs = []
    for tun in ({"desc_gather_stream": 1, "gemmr_wgs": wgs}, {"desc_gather_stream": 0}):
        ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=8, enc_chunk=4, tuning=tun, check_launches=1)
        o = _buffers(torch, 8, 0)
        ctx.detect_plnet_batch_dev(imgs, o["feat"], o["n"], o["lines"], o["nlines"], None, None, o["found"])
        ctx.sync()
        outs.append({k: o[k].cpu().numpy().copy() for k in ("feat", "n", "lines", "nlines", "found")})
        ctx.close()
    assert outs[0]["nlines"].min() >= 100
    for k in outs[0]:
        np.testing.assert_array_equal(outs[0][k], outs[1][k])


def test_plnet_batch_reports_overflow_and_is_deterministic():
    import torch
    ctx = _ctx()
    B = 4
    imgs = torch.from_numpy(_images(B, 30)).cuda()
    full = _buffers(torch, B, B)
    ctx.detect_plnet_batch_dev(imgs, full["feat"], full["n"], full["lines"], full["nlines"], full["junc"], full["njunc"], full["found"])
    small = _buffers(torch, B, B, cap_l=16, cap_j=8)
    ctx.detect_plnet_batch_dev(imgs, small["feat"], small["n"], small["lines"], small["nlines"], small["junc"], small["njunc"], small["found"])
    again = _buffers(torch, B, B)
    ctx.detect_plnet_batch_dev(imgs, again["feat"], again["n"], again["lines"], again["nlines"], again["junc"], again["njunc"], again["found"])
    ctx.sync()
    for k in full:
        assert torch.equal(full[k], again[k]), k
    # clamped counts, the true counts in `found`, and the first cap entries are the head of the full lists (ascending order is kept)
    assert torch.equal(small["found"], full["found"])
    assert (small["nlines"] == 16).all() and (small["njunc"] == 8).all() and (full["nlines"] > 16).all() and (full["njunc"] > 8).all()
    assert torch.equal(small["lines"], full["lines"][:, :16])
    assert torch.equal(small["junc"], full["junc"][:, :8])


def test_stereo_plnet_batch_equals_points_path_plus_lines():
    """One detector pass over 2 B images + lines of all + junctions of the left ones + LightGlue: the points and matches are the bits
    airfe_stereo_batch_dev gives, the lines / junctions the bits of the single-image calls (feature_detector.cc:94-104)."""
    import torch
    ctx = _ctx(lightglue=True)
    B = 3
    ls, rs = synth.stereo_batch(B, 480, 752, 21)
    L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    fl, fr, nl, nr = z(B, 400, 259), z(B, 400, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
    idx, sc, nm = z(B, 400, 2, dt=torch.int32), z(B, 400), z(B, dt=torch.int32)
    lines, nlines = z(2 * B, CAP_L, 4, dt=torch.float64), z(2 * B, dt=torch.int32)
    junc, njunc, found = z(B, CAP_J, 259), z(B, dt=torch.int32), z(3 * B, dt=torch.int32)
    ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm, found)
    ctx.sync()
    fl2, fr2, nl2, nr2 = z(B, 400, 259), z(B, 400, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
    idx2, sc2, nm2 = z(B, 400, 2, dt=torch.int32), z(B, 400), z(B, dt=torch.int32)
    ctx.stereo_batch_dev(L, R, fl2, fr2, nl2, nr2, idx2, sc2, nm2)
    ctx.sync()
    for a, b in ((fl, fl2), (fr, fr2), (nl, nl2), (nr, nr2), (idx, idx2), (sc, sc2), (nm, nm2)):
        assert torch.equal(a, b)
    assert int(nm.min()) >= 40
    nlh, njh = nlines.cpu().numpy(), njunc.cpu().numpy()
    for i, img in enumerate(list(ls) + list(rs)):
        _, ln, jc = ctx.detect_plnet(img, None, want_junctions=i < B)
        np.testing.assert_array_equal(lines[i, :nlh[i]].cpu().numpy(), ln)
        assert nlh[i] == ln.shape[0] >= 100
        if i < B:
            np.testing.assert_array_equal(junc[i, :njh[i]].cpu().numpy(), jc)
            assert njh[i] == jc.shape[0] >= 50
    diag("stereo_plnet_batch", lines=str(nlh.tolist()), junctions=str(njh.tolist()), matches=str(nm.cpu().numpy().tolist()))


def test_plnet_batch_refuses_what_it_cannot_do():
    import torch
    ctx = _ctx()
    o = _buffers(torch, 9, 0)
    with pytest.raises(api.AirfeError):                                  # 9 images > max_batch = 8
        ctx.detect_plnet_batch_dev(torch.from_numpy(_images(9, 1)).cuda(), o["feat"], o["n"], o["lines"], o["nlines"])
    sp = api.Context(superpoint=weights.synthetic_superpoint(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=2, enc_chunk=2)
    o = _buffers(torch, 2, 0)
    with pytest.raises(api.AirfeError):                                  # a point-only pack has no line branch to batch
        sp.detect_plnet_batch_dev(torch.from_numpy(_images(2, 1)).cuda(), o["feat"], o["n"], o["lines"], o["nlines"])
    sp.close()
