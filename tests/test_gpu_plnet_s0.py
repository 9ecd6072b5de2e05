"""PLNet stage-0 LINE branch on the device (SURVEY.md Appendix A.1 tensors) vs the oracle restatement (oracle/ref_nets.py::plnet_s0_lines).
Index work (junction top-300, nearest-junction matching, min / max, iskeep) is checked EXACT by feeding the device's own dense maps
to the numpy restatements; the network part within 2-byte tolerances; and the whole PLNet::infer path with no host tensors at all."""
import os

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN
from gpu_common import diag
from oracle import ref_chain, ref_nets, ref_post

pytestmark = pytest.mark.gpu
_C = {}


def _ctx(**kw):
    key = tuple(sorted(kw.items()))
    if key not in _C:
        _C[key] = (api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"),
                               max_batch=2, enc_chunk=2, **kw), weights.synthetic_plnet_s0(1234))
    return _C[key]


@pytest.mark.parametrize("seed", [5, 12])
def test_stage0_tensors_vs_oracle(seed):
    ctx, w = _ctx()
    img = synth.gabor_image(480, 752, seed)
    ctx.detect_points(img)
    dev = ctx.debug_plnet_stage0()
    x, _, _ = ref_post.process_image(img)
    ref = ref_nets.plnet_s0_lines(w, x)
    out = {}
    for k in ("loi_features", "loi_features_thin", "loi_features_aux", "jloc", "joff"):
        e = np.abs(dev[k] - ref[k]).max()
        out[k] = float(e / max(np.abs(ref[k]).max(), 1e-6))
        assert out[k] <= 0.02, f"{k}: {out[k]}"                       # fp16 trunk + fp16 line conv, fp32 heads
    # proposals: tan() near pi/2 amplifies the 2-byte error of md for a few pixels; clamp keeps them in the map
    le = np.abs(dev["lines_pred"] - ref["lines_pred"]).max(1)
    out["lines_p99"] = float(np.percentile(le, 99)); out["lines_max"] = float(le.max())
    assert np.percentile(le, 99) <= 0.05
    # ---- index work, exact on the device's own maps
    np.testing.assert_array_equal(dev["juncs_pred"], ref_nets.junctions_topk(dev["jloc"], dev["joff"], 300))
    keep, imin, imax = ref_nets.j2l_match(dev["lines_pred"], dev["juncs_pred"], 10.0)
    np.testing.assert_array_equal(dev["iskeep"].reshape(-1), keep)
    np.testing.assert_array_equal(dev["idx_junc_to_end_min"].reshape(-1), imin)
    np.testing.assert_array_equal(dev["idx_junc_to_end_max"].reshape(-1), imax)
    # ---- against the all-oracle chain: the same junctions (up to the few whose score sits on the top-300 boundary)
    d = np.linalg.norm(dev["juncs_pred"][:, None] - ref["juncs_pred"][None], axis=2).min(1)
    out["junc_within_half_px"] = float((d <= 0.5).mean())
    out["kept_dev"] = int(dev["iskeep"].sum()); out["kept_ref"] = int(ref["iskeep"].sum())
    diag(f"plnet_s0_{seed}", **out)
    assert (d <= 0.5).mean() >= 0.95
    assert abs(out["kept_dev"] - out["kept_ref"]) <= 0.01 * out["kept_ref"] and out["kept_ref"] > 1000     # measured: 0.3 %


@pytest.mark.parametrize("lt,ll,min_lines", [(0.5, 4.0, 600), (0.75, 50.0, 100)])      # permissive / the reference's defaults
def test_plnet_infer_needs_no_host_tensors(lt, ll, min_lines):
    """PLNet::infer end to end on the device: detect_plnet(stage0=None) == detect_plnet fed with the device's own stage-0 tensors
    through the host path (the path the golden tests pin), and the oracle's post-processing of those tensors gives the same lines —
    HUNDREDS of them: the structured synthetic line head (weights.synthetic_plnet_s0) makes the real stage-1 head accept a third of
    the candidates at the reference's thresholds, so the line filter, the junction map and the junction scan carry real volume."""
    ctx, w = _ctx(line_threshold=lt, line_length_threshold=ll)
    img = synth.gabor_image(480, 752, 8)
    feat, lines, junc = ctx.detect_plnet(img, None, want_junctions=True)
    dev = ctx.debug_plnet_stage0()
    feat2, lines2, junc2 = ctx.detect_plnet(img, dev, want_junctions=True)
    np.testing.assert_array_equal(feat, feat2)
    np.testing.assert_array_equal(lines, lines2)
    np.testing.assert_array_equal(junc, junc2)
    la, sc = ctx.debug_plnet_s1(dev)
    ref_lines, jmap = ref_post.line_filter(la, sc, 4, lt, ll)
    ref_lines = ref_post.rescale_lines(ref_lines, np.float32(752 / 512), np.float32(480 / 512))
    diag(f"plnet_infer_device_only_{lt}_{ll}", n_lines=lines.shape[0], n_unique=la.shape[0], n_junc=junc.shape[0], n_points=feat.shape[0],
         score_above_half=float((sc > 0.5).mean()), score_above_075=float((sc > 0.75).mean()))
    np.testing.assert_array_equal(lines, ref_lines)
    assert la.shape[0] > 200 and lines.shape[0] >= min_lines and junc.shape[0] >= 50
    assert 0.05 < (sc > 0.75).mean() < 0.95          # the score filter actually decides something
    det = api.FeatureDetector(ctx)
    acc = []
    ok, f, j = det.DetectLines(img, None, acc, junction_detection=True)
    assert ok and len(acc) == lines.shape[0] and j.shape[1] == junc.shape[0]


def _line_hits(a, b, tol=1.0):
    """share of the lines of `a` that have a line in `b` with BOTH endpoints within `tol` px (either orientation)"""
    if len(a) == 0 or len(b) == 0:
        return 0.0
    pa, pb = a.reshape(-1, 1, 2, 2), b.reshape(1, -1, 2, 2)
    d_same = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 0], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 1], axis=-1))
    d_swap = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 1], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 0], axis=-1))
    return float((np.minimum(d_same, d_swap).min(1) <= tol).mean())


@pytest.mark.parametrize("seed", [5, 8, 12, 33])
def test_lines_end_to_end_vs_the_all_oracle_chain(seed):
    """image -> lines, the device's default chain (fp16 storage through trunk and line conv, f32 stage 1) against the ALL-ORACLE fp32 chain
    (oracle/ref_chain.py = src/plnet.cpp:221-244 end to end) at the reference's thresholds 0.75 / 50: nothing of the device is fed to the
    oracle here (test_plnet_infer_needs_no_host_tensors pins the post-processing on the device's own tensors; this pins the precision of the
    whole line branch).  The reference runs stage 0 in TF32 (plnet.cpp:205) — the same 10-bit mantissa as the fp16 storage used here."""
    ctx, w = _ctx(line_threshold=0.75, line_length_threshold=50.0)
    s1 = weights.load_pack(os.path.join(GOLDEN, "plnet_s1.airfe"))
    img = synth.gabor_image(480, 752, seed)
    feat, lines, junc = ctx.detect_plnet(img, None, want_junctions=True)
    ref = ref_chain.plnet_infer(w, s1, img, want_junctions=True)
    rl, rj = ref["lines"], ref["junctions"]
    hit_dev, hit_ref = _line_hits(lines, rl), _line_hits(rl, lines)
    dj = np.linalg.norm(junc[:, None, 1:3] - rj[None, :, 1:3], axis=2).min(1) if len(junc) and len(rj) else np.ones(1) * 9
    diag(f"plnet_lines_e2e_{seed}", n_dev=len(lines), n_ref=len(rl), dev_lines_with_an_oracle_line=hit_dev, oracle_lines_with_a_device_line=hit_ref,
         junc_dev=len(junc), junc_ref=len(rj), junc_within_1px=float((dj <= 1.0).mean()))
    # WHY the two line sets differ where they do: every oracle line the device does not have, and every device line the
    # oracle does not have, is classified by ITS OWN side's numbers — how far its stage-1 score is above the 0.75 threshold, how far its length above
    # 50 px, whether the other side has junctions within 1 px of both its endpoints at all (the top-300 junction set itself differs in a few places
    # between fp16 and fp32).  A line that is confidently a line (score margin > 0.1, length margin > 3 px) AND whose junctions both exist on the
    # other side may NOT be missing: such a loss would be a defect, not rounding at a threshold.
    def explain(own_la, own_sc, own_lines, other_lines, other_juncs512, other_la, other_sc):
        kept = [u for u in range(len(own_sc)) if own_sc[u] >= np.float32(0.75) and
                np.float32((own_la[u, 2] - own_la[u, 0]) * 4) ** 2 + np.float32((own_la[u, 3] - own_la[u, 1]) * 4) ** 2 >= np.float32(2500.0)]
        assert len(kept) == len(own_lines)
        out = []
        pa, pb = own_lines.reshape(-1, 1, 2, 2), other_lines.reshape(1, -1, 2, 2)
        d_same = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 0], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 1], axis=-1))
        d_swap = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 1], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 0], axis=-1))
        missing = np.nonzero(np.minimum(d_same, d_swap).min(1) > 1.0)[0]
        for i in missing:
            u = kept[i]
            e = own_la[u].reshape(2, 2) * 4
            length = float(np.linalg.norm(e[1] - e[0]))
            jd = [float(np.linalg.norm(other_juncs512 - e[k][None], axis=1).min()) for k in (0, 1)]
            # the SAME candidate on the other side (both end junctions within 1 px, either orientation): its score there, or None when the other
            # side has no such candidate at all (the proposal was not kept there / its junction pair differs)
            oe = other_la.reshape(-1, 2, 2) * 4
            d1 = np.maximum(np.linalg.norm(oe[:, 0] - e[0][None], axis=1), np.linalg.norm(oe[:, 1] - e[1][None], axis=1))
            d2 = np.maximum(np.linalg.norm(oe[:, 0] - e[1][None], axis=1), np.linalg.norm(oe[:, 1] - e[0][None], axis=1))
            dmin = np.minimum(d1, d2)
            k = int(dmin.argmin()) if len(dmin) else -1
            other = float(other_sc[k]) if k >= 0 and dmin[k] <= 1.0 else None
            out.append(dict(score_margin=float(own_sc[u]) - 0.75, length_margin=length - 50.0, junction_dist=max(jd), score_on_the_other_side=other))
        return out
    s0d = ctx.debug_plnet_stage0()
    lad, scd = ctx.debug_plnet_s1(s0d)
    lost = explain(ref["lines_adjusted"], ref["scores_line"], ref_post.line_filter(ref["lines_adjusted"], ref["scores_line"], 4, 0.75, 50.0)[0],
                   ref_post.line_filter(lad, scd, 4, 0.75, 50.0)[0], s0d["juncs_pred"] * 4, lad, scd)
    extra = explain(lad, scd, ref_post.line_filter(lad, scd, 4, 0.75, 50.0)[0],
                    ref_post.line_filter(ref["lines_adjusted"], ref["scores_line"], 4, 0.75, 50.0)[0], ref["stage0"]["juncs_pred"] * 4,
                    ref["lines_adjusted"], ref["scores_line"])
    # a confidently-a-line candidate that EXISTS on the other side too but was scored below the threshold there: the stage-1 score moved by more than
    # its margin between fp16 and fp32 stage-0 tensors
    confident = [d for d in lost + extra if d["score_margin"] > 0.1 and d["length_margin"] > 3.0 and d["junction_dist"] <= 1.0
                 and d["score_on_the_other_side"] is not None]
    diag(f"plnet_lines_e2e_why_{seed}", oracle_only=len(lost), device_only=len(extra),
         at_the_score_threshold=sum(d["score_margin"] <= 0.1 for d in lost + extra), at_the_length_threshold=sum(d["length_margin"] <= 3.0 for d in lost + extra),
         junction_moved=sum(d["junction_dist"] > 1.0 for d in lost + extra), confident_and_missing=len(confident),
         margins=[[round(d["score_margin"], 3), round(d["length_margin"], 1), round(d["junction_dist"], 2),
                   None if d["score_on_the_other_side"] is None else round(d["score_on_the_other_side"], 3)] for d in lost + extra])
    assert len(rl) >= 100 and len(rj) >= 50
    assert abs(len(lines) - len(rl)) <= 0.03 * len(rl)
    assert hit_dev >= 0.95 and hit_ref >= 0.95
    # measured (gpurun_out/diag/plnet_lines_e2e_why_*): of 175-275 lines per image 8-25 differ; most are candidates the other side does not even have
    # (its proposal -> junction-pair assignment differs: a stage-0 effect), the rest sit within 0.1 of the 0.75 threshold on their own side — except
    # 0-1 per image whose stage-1 score moves by 0.1-0.3 between fp16 and fp32 LOI features (the seeded synthetic head turns 2-byte rounding of its
    # inputs into logit changes of ~1; the fp32 mode reproduces the oracle's line set exactly, tests/test_gpu_fp32.py).  Gate: at most 1 % of the lines.
    assert len(confident) <= max(1, int(0.01 * len(rl))), f"confident lines present on one side only: {confident}"
    assert abs(len(junc) - len(rj)) <= 0.03 * len(rj) and (dj <= 1.0).mean() >= 0.95


@pytest.mark.parametrize("seed", [5, 8, 12, 33])
def test_cell_search_match_equals_brute_force_where_it_is_read(seed):
    """The line path's junction-to-line match looks at the 3 x 3 cells of 8 x 8 pixels around an endpoint instead of at all 300 junctions:
    iskeep must be the brute-force kernel's EVERYWHERE, min / max wherever iskeep > 0 (all wireframe_matcher, plnet.cpp:272-307, reads)."""
    ctx, _ = _ctx()
    ctx.detect_points(synth.gabor_image(480, 752, seed))
    k0, a0, b0 = ctx.debug_plnet_j2l(fast=False)
    k1, a1, b1 = ctx.debug_plnet_j2l(fast=True)
    np.testing.assert_array_equal(k0, k1)
    kept = k0 > 0
    np.testing.assert_array_equal(a0[kept], a1[kept])
    np.testing.assert_array_equal(b0[kept], b1[kept])
    diag(f"plnet_j2l_{seed}", kept=int(kept.sum()), same_elsewhere=float((a0[~kept] == a1[~kept]).mean()))
    assert kept.sum() > 1000


def test_point_only_pack_gives_points_only():
    """A detector pack WITHOUT line.* tensors (plain SuperPoint) and no host tensors: points, zero lines, no error — the reference-shaped
    shim prints that at build() (shim/src/plnet.cpp)."""
    ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=2, enc_chunk=2)
    feat, lines, junc = ctx.detect_plnet(synth.gabor_image(480, 752, 8), None, want_junctions=True)
    assert feat.shape[0] > 0 and lines.shape == (0, 4) and junc.shape == (0, 259)
    ctx.close()
