"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

The INDEPENDENT pin of the network bodies.  The reference executes SuperPoint, LightGlue and SuperGlue through TensorRT engines
built from ONNX files that are absent from its checkout (call sites src/super_point.cpp:133, src/light_glue.cpp:159,
src/super_glue.cpp:185; .MISSING_LARGE_BLOBS:1-6), so the arithmetic of those bodies is restated in oracle/ref_nets.py from the
published architectures.  Hugging Face `transformers` (5.15.0 in this image) carries ports of the same three networks written by
other people from the same publications (transformers/models/{superpoint,lightglue,superglue}/modeling_*.py, converted from the
magicleap / cvg checkpoints).  This module loads the weight dicts of `airslam_amd.weights` (state_dict naming of the original
repositories) into those modules and runs them — `tests/test_oracle_hf_pin_cpu.py` compares the results with oracle/ref_nets.py,
`tools/make_hf_fixtures.py` stores them as `tests/golden/hf_pin.npz` for the GPU box (where transformers may be missing), and
`tests/test_gpu_hf_pin.py` compares the HIP library with them directly.

What the mappings say about the layouts (each is a statement the pin tests check):
  * SuperPoint: `encoder.conv_blocks.{0..3}.conv_{a,b}` = conv{1..4}{a,b}; `keypoint_decoder.conv_score_{a,b}` = convP{a,b};
    `descriptor_decoder.conv_descriptor_{a,b}` = convD{a,b}.  `_get_pixel_scores` = softmax-65, dustbin dropped, 8x8
    depth-to-space AND simple_nms(4) — i.e. it pins oracle/ref_post.simple_nms too (the reference's engine has the NMS in the graph:
    src/super_point.cpp:228-262 reads the suppressed map).
  * LightGlue (cvg/LightGlue naming): `Wqkv` rows are interleaved (head, dim, {q,k,v}) — HF's separate q/k/v projections are the
    de-interleaved rows; cross attention's shared `to_qk` is HF's q_proj AND k_proj; `ffn.{0,1,3}` = fc1 / layer_norm / fc2.
  * SuperGlue (magicleap naming): channel c of proj.{0,1,2} outputs / of merge's input is (dim, head) = (c // 4, c % 4) in the
    original `view(dim, heads, N)`; HF stores the head-major permutation (head * 64 + dim).  HF keeps BatchNorm1d in every MLP; the
    oracle's weights are the BatchNorm-folded inference form (what an ONNX export of an eval() module contains), so BatchNorm is
    set to the identity here (mean 0, var 1, gain 1, shift 0, eps 1e-30: x comes back bit for bit).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

W = Dict[str, np.ndarray]


def transformers_version() -> str:
    import transformers
    return transformers.__version__


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _set(p: torch.nn.Parameter, a) -> None:
    a = _t(a)
    assert tuple(p.shape) == tuple(a.shape), (tuple(p.shape), tuple(a.shape))
    with torch.no_grad():
        p.copy_(a)


def _lin(m: torch.nn.Module, w: W, name: str) -> None:
    _set(m.weight, w[name + ".weight"])
    _set(m.bias, w[name + ".bias"])


# ------------------------------------------------------------------ SuperPoint
def superpoint_model(w: W):
    from transformers import SuperPointConfig, SuperPointForKeypointDetection
    m = SuperPointForKeypointDetection(SuperPointConfig()).eval()
    for i in range(4):
        _lin(m.encoder.conv_blocks[i].conv_a, w, f"conv{i + 1}a")
        _lin(m.encoder.conv_blocks[i].conv_b, w, f"conv{i + 1}b")
    _lin(m.keypoint_decoder.conv_score_a, w, "convPa")
    _lin(m.keypoint_decoder.conv_score_b, w, "convPb")
    _lin(m.descriptor_decoder.conv_descriptor_a, w, "convDa")
    _lin(m.descriptor_decoder.conv_descriptor_b, w, "convDb")
    return m


def superpoint_maps(w: W, x: np.ndarray):
    """x [H,W] float32 in [0,1] -> (nms'd score map [H,W], dense L2-normalised descriptors [256,H/8,W/8]) from HF's modules:
    encoder -> keypoint_decoder._get_pixel_scores (softmax, depth-to-space, simple_nms(nms_radius = 4)) and
    descriptor_decoder.conv_descriptor_{a,b} + normalize (the first two lines of its forward)."""
    m = superpoint_model(w)
    assert m.keypoint_decoder.nms_radius == 4
    with torch.no_grad():
        enc = m.encoder(_t(x)[None, None], return_dict=True).last_hidden_state
        heat = m.keypoint_decoder._get_pixel_scores(enc)[0]
        dd = m.descriptor_decoder
        desc = torch.nn.functional.normalize(dd.conv_descriptor_b(dd.relu(dd.conv_descriptor_a(enc))), p=2, dim=1)[0]
    return heat.numpy(), desc.numpy()


# ------------------------------------------------------------------ LightGlue
def lightglue_model(w: W, n_layers: int = 9):
    from transformers import LightGlueConfig, LightGlueForKeypointMatching
    cfg = LightGlueConfig(num_hidden_layers=n_layers, attn_implementation="eager")
    m = LightGlueForKeypointMatching(cfg).eval()
    assert isinstance(m.input_projection, torch.nn.Identity)
    _set(m.positional_encoder.projector.weight, w["posenc.Wr.weight"])
    d, heads = 256, 4
    for i in range(n_layers):
        L = m.transformer_layers[i]
        s = f"transformers.{i}.self_attn"
        wq = w[s + ".Wqkv.weight"].reshape(heads, d // heads, 3, d)
        bq = w[s + ".Wqkv.bias"].reshape(heads, d // heads, 3)
        for j, proj in enumerate((L.self_attention.q_proj, L.self_attention.k_proj, L.self_attention.v_proj)):
            _set(proj.weight, wq[:, :, j].reshape(d, d))
            _set(proj.bias, bq[:, :, j].reshape(d))
        _lin(L.self_attention.o_proj, w, s + ".out_proj")
        _lin(L.self_mlp.fc1, w, s + ".ffn.0"); _lin(L.self_mlp.layer_norm, w, s + ".ffn.1"); _lin(L.self_mlp.fc2, w, s + ".ffn.3")
        c = f"transformers.{i}.cross_attn"
        _lin(L.cross_attention.q_proj, w, c + ".to_qk"); _lin(L.cross_attention.k_proj, w, c + ".to_qk")
        _lin(L.cross_attention.v_proj, w, c + ".to_v"); _lin(L.cross_attention.o_proj, w, c + ".to_out")
        _lin(L.cross_mlp.fc1, w, c + ".ffn.0"); _lin(L.cross_mlp.layer_norm, w, c + ".ffn.1"); _lin(L.cross_mlp.fc2, w, c + ".ffn.3")
    a = f"log_assignment.{n_layers - 1}"
    A = m.match_assignment_layers[n_layers - 1]
    _lin(A.final_projection, w, a + ".final_proj")
    _lin(A.matchability, w, a + ".matchability")
    assert L.self_mlp.layer_norm.eps == 1e-5
    return m


def lightglue_scores(w: W, kpts0, desc0, kpts1, desc1, n_layers: int = 9):
    """kpts [N,2] normalised (PointMatcher::NormalizeKeypoints), desc [N,256] -> HF's full log-assignment [N0+1, N1+1] (its last row /
    column are the dustbins; the reference's binding `scores` is the inner block, src/light_glue.cpp:270-278).  N0 != N1 goes through
    HF's padding mask, the path its own batching takes."""
    m = lightglue_model(w, n_layers)
    n0, n1 = len(kpts0), len(kpts1)
    n = max(n0, n1)
    k = torch.zeros(2, n, 2); d = torch.zeros(2, n, 256); mask = torch.zeros(2, n, dtype=torch.int)
    k[0, :n0] = _t(kpts0); k[1, :n1] = _t(kpts1); d[0, :n0] = _t(desc0); d[1, :n1] = _t(desc1)
    mask[0, :n0] = 1; mask[1, :n1] = 1
    with torch.no_grad():
        enc = m.positional_encoder(k)[0]
        att = None
        if n0 != n1:
            att = torch.zeros(2, 1, 1, n)
            att.masked_fill_(mask[:, None, None, :] == 0, torch.finfo(torch.float32).min)
        x = d
        for i in range(n_layers):
            x = m.transformer_layers[i](x, enc, attention_mask=att)[0]
        s = m.match_assignment_layers[n_layers - 1](x, mask if n0 != n1 else None)[0]
    s = s.numpy()
    return np.ascontiguousarray(np.concatenate([np.concatenate([s[:n0, :n1], s[:n0, -1:]], 1),
                                                np.concatenate([s[-1:, :n1], s[-1:, -1:]], 1)], 0))


def lightglue_matches(w: W, kpts0, desc0, kpts1, desc1, threshold: float, n_layers: int = 9):
    """HF's own `get_matches_from_scores` on HF's log-assignment (N0 == N1 only): -> (matches0 [N0] int, matching_scores0 [N0])."""
    from transformers.models.lightglue.modeling_lightglue import get_matches_from_scores
    s = lightglue_scores(w, kpts0, desc0, kpts1, desc1, n_layers)
    mt, ms = get_matches_from_scores(torch.from_numpy(s)[None], threshold)
    return mt[0].numpy(), ms[0].numpy()


# ------------------------------------------------------------------ SuperGlue
def _sg_perm(d: int = 256, heads: int = 4) -> np.ndarray:
    """perm[h * 64 + k] = k * 4 + h: HF's head-major channel -> the original view(dim, heads, N) channel."""
    dim = d // heads
    h, k = np.meshgrid(np.arange(heads), np.arange(dim), indexing="ij")
    return (k * heads + h).reshape(-1)


def _identity_bn(bn: torch.nn.BatchNorm1d) -> None:
    with torch.no_grad():
        bn.weight.fill_(1.0); bn.bias.zero_(); bn.running_mean.zero_(); bn.running_var.fill_(1.0)
    bn.eps = 1e-30                       # torch refuses eps = 0; 1 + 1e-30 == 1 in fp32 and fp64: (x - 0) / sqrt(1 + eps) * 1 + 0 == x bit for bit


def superglue_model(w: W, n_layers: int = 18, iters: int = 100):
    from transformers import SuperGlueConfig, SuperGlueForKeypointMatching
    cfg = SuperGlueConfig(gnn_layers_types=["self", "cross"] * (n_layers // 2), sinkhorn_iterations=iters)
    m = SuperGlueForKeypointMatching(cfg).eval()
    enc = m.keypoint_encoder.encoder
    assert len(enc) == 5
    for i in range(4):
        _lin(enc[i].linear, w, f"kenc.encoder.{i}")
        _identity_bn(enc[i].batch_norm)
    _lin(enc[4], w, "kenc.encoder.4")
    perm = _sg_perm()
    for i in range(n_layers):
        g = f"gnn.layers.{i}"
        L = m.gnn.layers[i]
        for j, proj in enumerate((L.attention.self.query, L.attention.self.key, L.attention.self.value)):
            _set(proj.weight, w[f"{g}.attn.proj.{j}.weight"][perm])
            _set(proj.bias, w[f"{g}.attn.proj.{j}.bias"][perm])
        _set(L.attention.output.dense.weight, w[g + ".attn.merge.weight"][:, perm])
        _set(L.attention.output.dense.bias, w[g + ".attn.merge.bias"])
        _lin(L.mlp[0].linear, w, g + ".mlp.0")
        _identity_bn(L.mlp[0].batch_norm)
        _lin(L.mlp[1], w, g + ".mlp.3")
    _lin(m.final_projection.final_proj, w, "final_proj")
    with torch.no_grad():
        m.bin_score.copy_(_t(w["bin_score"]).reshape(()))
    return m


def superglue_scores(w: W, kpts0, sc0, desc0, kpts1, sc1, desc1, n_layers: int = 18, iters: int = 100):
    """kpts [N,2] normalised, sc [N], desc [N,256] -> HF's log optimal transport [N0+1, N1+1] (binding `scores` of the reference's
    engine, decoded at src/super_glue.cpp:447-453).  N0 == N1 runs HF's batch-of-2 layout (both images in one batch, cross layers by
    `flip`); N0 != N1 runs the two images as two batch-1 calls of the same modules, with the other image as encoder_hidden_states."""
    from transformers.models.superglue.modeling_superglue import log_optimal_transport
    m = superglue_model(w, n_layers, iters)
    with torch.no_grad():
        def start(k, s, d):
            return _t(d)[None] + m.keypoint_encoder(_t(k)[None], _t(s)[None])[0]
        x0, x1 = start(kpts0, sc0, desc0), start(kpts1, sc1, desc1)
        if x0.shape[1] == x1.shape[1]:
            x = m.gnn(torch.cat([x0, x1], 0), mask=None)[0]
            x0, x1 = x[0:1], x[1:2]
        else:
            for layer, kind in zip(m.gnn.layers, m.gnn.layers_types):
                if kind == "cross":
                    d0, d1 = layer(x0, encoder_hidden_states=x1)[0], layer(x1, encoder_hidden_states=x0)[0]
                else:
                    d0, d1 = layer(x0)[0], layer(x1)[0]
                x0, x1 = x0 + d0, x1 + d1
        p0, p1 = m.final_projection(x0), m.final_projection(x1)
        s = p0 @ p1.transpose(1, 2) / m.config.hidden_size ** 0.5
        z = log_optimal_transport(s, m.bin_score, iterations=m.config.sinkhorn_iterations)[0]
    return z.numpy()


def superglue_matches(w: W, kpts_px0, sc0, desc0, kpts_px1, sc1, desc1, height: int, width: int, threshold: float,
                      n_layers: int = 18, iters: int = 100):
    """HF's `_match_image_pair` end to end on PIXEL keypoints (its own normalize_keypoints: (k - size / 2) / (0.7 max(w, h)), what
    PointMatcher::NormalizeKeypoints computes with scale 0.7, src/point_matcher.cc:39-48,58) -> (matches [2,N], matching_scores [2,N])."""
    m = superglue_model(w, n_layers, iters)
    m.config.matching_threshold = threshold
    n = len(kpts_px0)
    assert len(kpts_px1) == n
    with torch.no_grad():
        k = torch.stack([_t(kpts_px0), _t(kpts_px1)])[None]
        d = torch.stack([_t(desc0), _t(desc1)])[None]
        s = torch.stack([_t(sc0), _t(sc1)])[None]
        mt, ms, _, _ = m._match_image_pair(k, d, s, height, width, mask=torch.ones(1, 2, n, dtype=torch.int))
    return mt[0].numpy(), ms[0].numpy()
