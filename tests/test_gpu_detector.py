"""Detector parity on the GPU (C ABI) vs the CPU oracle.  Index/byte work is checked BIT-EXACT by feeding the
device's own dense maps to the oracle's post-processing; the network body is checked within the north-star
tolerances (descriptors <= 1e-3 cosine distance, keypoints <= 1 px)."""
import numpy as np
import pytest

from airslam_amd import synth
from gpu_common import context, cosine_dist, diag
from oracle import ref_nets, ref_post

pytestmark = pytest.mark.gpu
CFG = dict(max_batch=4, enc_chunk=2)


# (three kernels behind one entry: four output rows per workgroup for 4-byte-aligned rows up to 2048 pixels — every camera of the reference's configs —, one row per
#  workgroup up to 4096 (the 2560-wide case), per pixel for unaligned rows (the 131-wide case))
@pytest.mark.parametrize("h,w,seed", [(480, 752, 0), (480, 640, 2), (720, 1280, 3), (512, 512, 4), (100, 131, 5), (1440, 2560, 6), (1080, 2048, 7)])
def test_preprocess_bit_exact(h, w, seed):
    ctx, _, _ = context("sp", **CFG)
    img = synth.gabor_image(h, w, seed)
    out = ctx.debug_preprocess(img)
    ref, _, _ = ref_post.process_image(img)
    np.testing.assert_array_equal(out, ref)


def test_preprocess_strided_rows():
    ctx, _, _ = context("sp", **CFG)
    big = synth.gabor_image(480, 800, 9)
    view = big[:, 24:776]                   # non-contiguous rows, like a cv::Mat ROI (stride != width)
    np.testing.assert_array_equal(ctx.debug_preprocess(np.ascontiguousarray(view)), ref_post.process_image(view)[0])
    f1 = ctx.detect_points(view)
    f2 = ctx.detect_points(np.ascontiguousarray(view))
    np.testing.assert_array_equal(f1, f2)


def _oracle_maps(sp, img):
    x, ws, hs = ref_post.process_image(img)
    heat, desc = ref_nets.superpoint_forward(sp, x[None])
    return heat[0], desc[0], ws, hs


def test_detector_network_vs_oracle():
    """Default storage (fp16, the reference's kFP16): the north-star tolerances on the dense maps."""
    ctx, sp, _ = context("sp", **CFG)
    img = synth.gabor_image(480, 752, 0)
    feat = ctx.detect_points(img)
    heat, nms, desc = ctx.detector_maps(1)
    oh, od, ws, hs = _oracle_maps(sp, img)
    herr = np.abs(heat[0] - oh)
    cd = cosine_dist(desc[0].reshape(-1, 256), od.transpose(1, 2, 0).reshape(-1, 256))
    diag("detector_maps", heat_max_err=herr.max(), heat_mean_err=herr.mean(), heat_max=oh.max(), desc_cos_max=cd.max(),
         desc_cos_mean=cd.mean(), n_kpts=feat.shape[0], heat_sum_dev=float(heat[0].sum()), heat_sum_ref=float(oh.sum()))
    assert cd.max() <= 1e-3, "dense descriptors: cosine distance above the north-star tolerance"
    # heat = softmax of logits of magnitude ~10: a 2-byte activation error d shows up as a RELATIVE heat error ~d
    assert herr.max() <= 0.01 * max(oh.max(), 1e-3) + 1e-3


def test_detector_network_bf16_storage_is_looser():
    """precision = 0 (bf16): same kernels, three fewer mantissa bits.  With DECORRELATED descriptors (the whitened synthetic head,
    like a trained one) the dense descriptor map misses the 1e-3 cosine tolerance by 20x (round 1 passed only because a plain
    random head makes all descriptors collinear, where the cosine cannot see the error) — which is why fp16 is the default."""
    ctx, sp, _ = context("sp", precision=0, **CFG)
    img = synth.gabor_image(480, 752, 0)
    ctx.detect_points(img)
    heat, nms, desc = ctx.detector_maps(1)
    oh, od, ws, hs = _oracle_maps(sp, img)
    herr = np.abs(heat[0] - oh)
    cd = cosine_dist(desc[0].reshape(-1, 256), od.transpose(1, 2, 0).reshape(-1, 256))
    diag("detector_maps_bf16", heat_max_err=herr.max(), heat_mean_err=herr.mean(), desc_cos_max=cd.max(), desc_cos_mean=cd.mean())
    assert cd.max() <= 0.05 and cd.mean() <= 0.01
    assert herr.max() <= 0.05 * max(oh.max(), 1e-3) + 2e-3


def test_nms_and_decode_bit_exact_on_device_maps():
    """simple_nms, top-K select, descriptor sampling: exact given the SAME dense maps."""
    ctx, sp, _ = context("sp", **CFG)
    img = synth.gabor_image(480, 752, 1)
    feat = ctx.detect_points(img)
    heat, nms, desc = ctx.detector_maps(1)
    np.testing.assert_array_equal(nms[0], ref_post.simple_nms(heat[0], 4))
    ws, hs = np.float32(752 / 512), np.float32(480 / 512)
    ref = ref_post.keypoints_decoder(nms[0], np.ascontiguousarray(desc[0].transpose(2, 0, 1)), 0.004, 4, 400, ws, hs)
    diag("decode_exact", n_dev=feat.shape[0], n_ref=ref.shape[0],
         n_cand=int(((nms[0] >= 0.004)).sum()))
    assert feat.shape == ref.shape
    np.testing.assert_array_equal(feat[:, :3], ref[:, :3])                 # scores, x, y: bit-exact
    np.testing.assert_allclose(feat[:, 3:], ref[:, 3:], atol=2e-6, rtol=0)   # fp32 bilinear + L2 norm
    assert feat.shape[0] > 20, "synthetic image produced too few keypoints for a meaningful test"


@pytest.mark.parametrize("thr,topk,nms", [(0.0005, 1024, 0), (0.2, 400, 4), (0.004, 50, 2)])
def test_select_regimes(thr, topk, nms):
    """count > K (sorted, tie rule) and count <= K (raster order) both go through the exact path."""
    ctx, sp, _ = context("sp", max_batch=2, enc_chunk=2, keypoint_threshold=thr, max_keypoints=topk, nms_radius=nms)
    img = synth.gabor_image(480, 752, 6)
    feat = ctx.detect_points(img)
    heat, nmsm, desc = ctx.detector_maps(1)
    ws, hs = np.float32(752 / 512), np.float32(480 / 512)
    ref = ref_post.keypoints_decoder(nmsm[0], np.ascontiguousarray(desc[0].transpose(2, 0, 1)), thr, 4, topk, ws, hs)
    diag(f"select_{thr}_{topk}_{nms}", n_dev=feat.shape[0], n_ref=ref.shape[0])
    assert feat.shape == ref.shape
    np.testing.assert_array_equal(feat[:, :3], ref[:, :3])


def test_keypoints_vs_oracle_end_to_end():
    ctx, sp, _ = context("sp", **CFG)
    img = synth.gabor_image(480, 752, 0)
    feat = ctx.detect_points(img)
    oh, od, ws, hs = _oracle_maps(sp, img)
    ref = ref_post.keypoints_decoder(ref_post.simple_nms(oh, 4), od, 0.004, 4, 400, ws, hs)
    # <= 1 px: every device keypoint has an oracle keypoint within 1 px (in 512-space) and vice versa, up to the
    # few that sit on the threshold / top-K boundary where 2-byte activations legitimately flip membership
    dx = feat[:, None, 1] / ws - ref[None, :, 1] / ws
    dy = feat[:, None, 2] / hs - ref[None, :, 2] / hs
    d2 = dx * dx + dy * dy
    near = d2.min(1) <= 1.0 + 1e-6
    j = d2.argmin(1)
    cd = cosine_dist(feat[near, 3:], ref[j[near], 3:])
    diag("e2e_keypoints", n_dev=feat.shape[0], n_ref=ref.shape[0], frac_within_1px=near.mean(), desc_cos_max=cd.max(),
         desc_cos_mean=cd.mean(), score_err_max=np.abs(feat[near, 0] - ref[j[near], 0]).max())
    # north star: keypoints <= 1 px, descriptors <= 1e-3 cosine.  The remainder (< 1 %) are detections whose score sits on the
    # threshold / top-K boundary, where ANY rounding of the network flips membership (the reference's own FP16 engine does too)
    assert near.mean() >= 0.99
    assert cd.max() <= 1e-3


def test_batch_dev_equals_host_path():
    import torch
    ctx, sp, _ = context("sp", **CFG)
    ls, rs = synth.stereo_batch(3, 480, 752, 11)
    g = torch.from_numpy(ls).cuda()
    feat = torch.zeros((3, 448, 259), dtype=torch.float32, device="cuda")
    n = torch.zeros((3,), dtype=torch.int32, device="cuda")
    ctx.detect_batch_dev(g, feat, n)
    ctx.sync()
    for b in range(3):
        ref = ctx.detect_points(ls[b])
        assert int(n[b]) == ref.shape[0]
        np.testing.assert_array_equal(feat[b, :ref.shape[0]].cpu().numpy(), ref)


def test_descriptor_head_streaming_gather_gives_the_bits_of_the_tiled_kernel():
    """Round 5: at large batches the descriptor head over the sampled cells (four per keypoint) runs in the streaming GEMM with gathered source rows
    (kernels_gemmr.hip GATHER; from gemmr_min_m rows on) instead of the tiled 8-wave kernel's gather form.  Same fragments, same K order, bias after the sum:
    feature rows (score, x, y, 256 descriptor floats) must be bit-identical — and equal to the batch-1 host path, whose descriptors come from the DENSE map.
    gemmr_wgs = 16 makes every workgroup stream many tiles (the DMA ring wraps, the index list behind the ring is long)."""
    import torch
    from airslam_amd import api, weights
    B = 10
    ls, _ = synth.stereo_batch(B, 480, 752, 23)
    g = torch.from_numpy(ls).cuda()
    outs = []
    for tun in ({"desc_gather_stream": 1}, {"desc_gather_stream": 1, "gemmr_wgs": 16}, {"desc_gather_stream": 0}):
        ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), max_batch=B, enc_chunk=B, max_keypoints=400, tuning=tun, check_launches=1)
        feat = torch.zeros((B, 400, 259), dtype=torch.float32, device="cuda")
        n = torch.zeros((B,), dtype=torch.int32, device="cuda")
        ctx.detect_batch_dev(g, feat, n)
        ctx.sync()
        outs.append((n.cpu().numpy().copy(), feat.cpu().numpy().copy()))
        if tun == {"desc_gather_stream": 1}:
            ref0 = ctx.detect_points(ls[0])                           # batch 1: dense descriptor map + sampler
        ctx.close()
    assert outs[0][0].min() >= 200                                    # B * 400 * 4 = 16000 rows >= gemmr_min_m = 8192: the streaming form ran
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][0], o[0])
        np.testing.assert_array_equal(outs[0][1], o[1])
    np.testing.assert_array_equal(outs[0][1][0, :ref0.shape[0]], ref0)


def test_empty_image_is_an_error_like_the_reference():
    from airslam_amd import api
    ctx, _, _ = context("sp", **CFG)
    ok, f = api.FeatureDetector(ctx).Detect(np.zeros((0, 0), np.uint8))
    assert ok is False and f.shape == (259, 0)


def test_malformed_weight_pack_is_refused(tmp_path):
    """load_pack multiplies untrusted dims: a tensor that claims more elements than the file holds (or whose dims overflow) must make
    airfe_create fail with a message, not allocate 2^64 bytes."""
    import struct
    from airslam_amd import api

    def pack(dims, payload=b""):
        name = b"conv1a.weight"
        return b"AIRFEPK1" + struct.pack("<I", 1) + struct.pack("<I", len(name)) + name + struct.pack("<I", len(dims)) + \
            struct.pack(f"<{len(dims)}I", *dims) + payload

    cases = {"huge.airfe": pack([0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF]), "short.airfe": pack([64, 9], b"\0" * 100),
             "overflow.airfe": pack([0x80000000, 0x80000000, 4, 4])}
    for fn, blob in cases.items():
        p = tmp_path / fn
        p.write_bytes(blob)
        with pytest.raises(api.AirfeError, match="malformed weight pack"):
            api.Context(superpoint=str(p))
