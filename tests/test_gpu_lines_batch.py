"""AssignPointsToLines / MatchLines over device-resident batches (airfe_assign_points_to_lines_batch_dev, airfe_match_lines_batch_dev: VERDICT r03
missing #5) on the OUTPUTS of the PLNet stereo step, in place: per frame the results must be the bits of the host-pointer entries — which
tests/test_gpu_ref_pin.py / test_gpu_lines.py hold to the reference's own code — and the stereo filter of Frame::AddRightFeatures
(src/frame.cc:147-160) must equal filtering the match list first."""
import os

import numpy as np
import pytest
import torch

from airslam_amd import api, synth, weights
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [3, 8])
def test_batch_dev_line_association_equals_the_host_entries(B):
    H, W, K, CL, CJ, CE = 480, 752, 400, 512, 1024, 8192
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), lightglue=weights.synthetic_lightglue(1234),
                      plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=B, enc_chunk=min(2 * B, 16), max_keypoints=K, image_width=W,
                      image_height=H)
    dev = torch.device("cuda")
    ls, rs = synth.stereo_batch(B, H, W, 31)
    L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
    fl = torch.zeros((B, K, 259), device=dev); fr = torch.zeros((B, K, 259), device=dev)
    nl = torch.zeros((B,), dtype=torch.int32, device=dev); nr = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx = torch.zeros((B, K, 2), dtype=torch.int32, device=dev); sc = torch.zeros((B, K), device=dev)
    nm = torch.zeros((B,), dtype=torch.int32, device=dev)
    lines = torch.zeros((2 * B, CL, 4), dtype=torch.float64, device=dev); nlines = torch.zeros((2 * B,), dtype=torch.int32, device=dev)
    junc = torch.zeros((B, CJ, 259), device=dev); njunc = torch.zeros((B,), dtype=torch.int32, device=dev)
    ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm)
    ctx.sync()
    nl[B - 1] = 0; nm[B - 1] = 0                              # a frame without points / matches: the early-outs of src/line_processor.cc:132
    nlines[2 * B - 2] = 0                                     # and a right frame without lines
    rp = [torch.zeros((B, CL + 1), dtype=torch.int32, device=dev) for _ in range(2)]
    pi = [torch.full((B, CE), -7, dtype=torch.int32, device=dev) for _ in range(2)]
    pd = [torch.zeros((B, CE), dtype=torch.float64, device=dev) for _ in range(2)]
    tot = [torch.zeros((B,), dtype=torch.int32, device=dev) for _ in range(2)]
    ctx.assign_points_to_lines_batch_dev(lines[:B], nlines[:B], fl, nl, rp[0], pi[0], pd[0], tot[0])
    ctx.assign_points_to_lines_batch_dev(lines[B:], nlines[B:], fr, nr, rp[1], pi[1], pd[1], tot[1])
    lm = torch.full((B, CL), -9, dtype=torch.int32, device=dev)
    lmf = torch.full((B, CL), -9, dtype=torch.int32, device=dev)
    ctx.match_lines_batch_dev(rp[0], pi[0], nlines[:B], nl, rp[1], pi[1], nlines[B:], nr, idx, nm, lm)
    band = (2.0, 60.0, 3.0)                                   # a disparity band that drops some of the synthetic pair's matches
    ctx.match_lines_batch_dev(rp[0], pi[0], nlines[:B], nl, rp[1], pi[1], nlines[B:], nr, idx, nm, lmf, stereo_filter=band, feat0_t=fl, feat1_t=fr)
    ctx.sync()
    h = lambda t: t.cpu().numpy()
    fl_h, fr_h, nl_h, nr_h, idx_h, nm_h, lines_h, nlines_h = map(h, (fl, fr, nl, nr, idx, nm, lines, nlines))
    n_lines_matched = n_dropped = 0
    for b in range(B):
        rels = []
        for side, (f, n, off) in enumerate(((fl_h, nl_h, 0), (fr_h, nr_h, B))):
            Lb = int(nlines_h[off + b])
            rel = ctx.assign_points_to_lines(lines_h[off + b, :Lb], f[b, :n[b]])
            want_rp = np.cumsum([0] + [len(r) for r in rel]).astype(np.int32)
            got_rp = h(rp[side])[b, :Lb + 1]
            np.testing.assert_array_equal(got_rp, want_rp)
            assert int(h(tot[side])[b]) == want_rp[-1] <= CE
            np.testing.assert_array_equal(h(pi[side])[b, :want_rp[-1]], np.array([k for r in rel for k in r], np.int32))
            np.testing.assert_array_equal(h(pd[side])[b, :want_rp[-1]], np.array([r[k] for r in rel for k in r], np.float64))
            rels.append(rel)
        m = [tuple(p) for p in idx_h[b, :nm_h[b]]]
        L0 = int(nlines_h[b])
        want = ctx.match_lines(rels[0], rels[1], m, int(nl_h[b]), int(nr_h[b]))
        np.testing.assert_array_equal(h(lm)[b, :L0], np.array(want, np.int32).reshape(-1))
        assert (h(lm)[b, L0:] == -9).all()                    # nothing behind a frame's own lines is touched
        keep = []
        for q, t in m:                                        # src/frame.cc:147-160, statement by statement
            dx = float(abs(np.float32(fl_h[b, q, 1] - fr_h[b, t, 1]))); dy = float(abs(np.float32(fl_h[b, q, 2] - fr_h[b, t, 2])))
            if dx > band[0] and dx < band[1] and dy <= band[2]:
                keep.append((q, t))
        n_dropped += len(m) - len(keep)
        wantf = ctx.match_lines(rels[0], rels[1], keep, int(nl_h[b]), int(nr_h[b]))
        np.testing.assert_array_equal(h(lmf)[b, :L0], np.array(wantf, np.int32).reshape(-1))
        n_lines_matched += int((np.array(want) >= 0).sum())
    assert (h(lm)[B - 1, :int(nlines_h[B - 1])] == -1).all() and (h(lm)[B - 2, :int(nlines_h[B - 2])] == -1).all()
    assert n_lines_matched >= 5 * (B - 2) and n_dropped > 0, (n_lines_matched, n_dropped)
    ctx.close()
