"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

PyTorch-CPU fp32 restatement of the network bodies the reference executes through
TensorRT engines built from ONNX files (call sites src/plnet.cpp:233,510;
src/super_point.cpp:133; src/light_glue.cpp:159; src/super_glue.cpp:185).

PARITY UNPINNED for SuperPoint / LightGlue / SuperGlue: their ONNX files are absent
from /root/reference (.MISSING_LARGE_BLOBS), onnxruntime is not installed, and the
reference has no tests.  The bodies below restate the PUBLISHED architectures
(SuperPoint v1 — DeTone et al.; LightGlue — Lindenberger et al., cvg/LightGlue
lightglue.py; SuperGlue — Sarlin et al., magicleap/SuperGluePretrainedNetwork), with
tensor names/shapes anchored on the reference's bindings.
The PLNet stage-1 head IS pinned: see oracle/onnx_run.py (real graph + weights).

All functions take a dict name -> np.ndarray (airslam_amd.weights naming) so that
the HIP library and the oracle consume byte-identical weights.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as Fn

W = Dict[str, np.ndarray]


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


# ------------------------------------------------------------------ SuperPoint
def superpoint_trunk(w: W, x: torch.Tensor) -> torch.Tensor:
    """x [B,1,H,W] in [0,1] -> conv4b activations [B,128,H/8,W/8]."""
    def c(name, t):
        return Fn.relu(Fn.conv2d(t, _t(w[name + ".weight"]), _t(w[name + ".bias"]), padding=1))
    x = c("conv1a", x); x = c("conv1b", x); x = Fn.max_pool2d(x, 2, 2)
    x = c("conv2a", x); x = c("conv2b", x); x = Fn.max_pool2d(x, 2, 2)
    x = c("conv3a", x); x = c("conv3b", x); x = Fn.max_pool2d(x, 2, 2)
    x = c("conv4a", x); x = c("conv4b", x)
    return x


def superpoint_heads(w: W, f: torch.Tensor):
    """-> heat [B,H,W] (softmax-65, dustbin dropped, 8x8 depth-to-space; NO nms),
    desc [B,256,H/8,W/8] channel-L2-normalised — the two bindings of
    superpoint_v1_sim_int32.onnx / `scores`,`descriptors` of plnet_s0 (A.1)."""
    cpa = Fn.relu(Fn.conv2d(f, _t(w["convPa.weight"]), _t(w["convPa.bias"]), padding=1))
    logits = Fn.conv2d(cpa, _t(w["convPb.weight"]), _t(w["convPb.bias"]))
    p = torch.softmax(logits, 1)[:, :-1]
    b, _, h, wd = p.shape
    p = p.permute(0, 2, 3, 1).reshape(b, h, wd, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, wd * 8)
    cda = Fn.relu(Fn.conv2d(f, _t(w["convDa.weight"]), _t(w["convDa.bias"]), padding=1))
    d = Fn.conv2d(cda, _t(w["convDb.weight"]), _t(w["convDb.bias"]))
    d = Fn.normalize(d, p=2, dim=1)
    return p, d


def superpoint_forward(w: W, x: np.ndarray):
    """x [B,H,W] float32 in [0,1] -> (heat [B,H,W], desc [B,256,H/8,W/8]) numpy."""
    with torch.no_grad():
        f = superpoint_trunk(w, _t(x)[:, None])
        p, d = superpoint_heads(w, f)
    return p.numpy(), d.numpy()


# ------------------------------------------------------------------ LightGlue
def _rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def _rope(freqs, t):
    return t * freqs[0] + _rotate_half(t) * freqs[1]


def _ffn(w: W, p: str, x):
    h = Fn.linear(x, _t(w[p + ".ffn.0.weight"]), _t(w[p + ".ffn.0.bias"]))
    h = Fn.layer_norm(h, (h.shape[-1],), _t(w[p + ".ffn.1.weight"]), _t(w[p + ".ffn.1.bias"]), eps=1e-5)
    h = Fn.gelu(h)
    return Fn.linear(h, _t(w[p + ".ffn.3.weight"]), _t(w[p + ".ffn.3.bias"]))


def lightglue_posenc(w: W, kpts: torch.Tensor):
    """LearnableFourierPositionalEncoding: Linear(2->32, no bias) -> cos/sin, repeat_interleave 2."""
    proj = Fn.linear(kpts, _t(w["posenc.Wr.weight"]))
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0)
    return emb.repeat_interleave(2, dim=-1)            # [2, N, 64]


def _lg_self(w: W, p: str, x, enc, heads=4):
    n, d = x.shape
    qkv = Fn.linear(x, _t(w[p + ".Wqkv.weight"]), _t(w[p + ".Wqkv.bias"]))
    qkv = qkv.unflatten(-1, (heads, -1, 3)).transpose(0, 1)          # [H, N, 64, 3]
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = _rope(enc[:, None], q)
    k = _rope(enc[:, None], k)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), -1) @ v   # [H,N,64]
    ctx = a.transpose(0, 1).flatten(start_dim=-2)
    msg = Fn.linear(ctx, _t(w[p + ".out_proj.weight"]), _t(w[p + ".out_proj.bias"]))
    return x + _ffn(w, p, torch.cat([x, msg], -1))


def _lg_cross(w: W, p: str, x0, x1, heads=4):
    def proj(name, x):
        y = Fn.linear(x, _t(w[f"{p}.{name}.weight"]), _t(w[f"{p}.{name}.bias"]))
        return y.unflatten(-1, (heads, -1)).transpose(0, 1)          # [H,N,64]
    qk0, qk1 = proj("to_qk", x0), proj("to_qk", x1)
    v0, v1 = proj("to_v", x0), proj("to_v", x1)
    scale = qk0.shape[-1] ** -0.5
    qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
    sim = qk0 @ qk1.transpose(-1, -2)
    m0 = torch.softmax(sim, -1) @ v1
    m1 = torch.softmax(sim.transpose(-1, -2), -1) @ v0
    def out(m):
        m = m.transpose(0, 1).flatten(start_dim=-2)
        return Fn.linear(m, _t(w[p + ".to_out.weight"]), _t(w[p + ".to_out.bias"]))
    m0, m1 = out(m0), out(m1)
    x0 = x0 + _ffn(w, p, torch.cat([x0, m0], -1))
    x1 = x1 + _ffn(w, p, torch.cat([x1, m1], -1))
    return x0, x1


def lightglue_forward(w: W, kpts0, desc0, kpts1, desc1, n_layers: int = 9, return_states: bool = False):
    """kpts [N,2] (already normalised by PointMatcher::NormalizeKeypoints), desc [N,256]
    -> log-assignment inner block `scores` [N0,N1] (binding A.4; src/light_glue.cpp:270-278)."""
    with torch.no_grad():
        x0, x1 = _t(desc0), _t(desc1)
        e0, e1 = lightglue_posenc(w, _t(kpts0)), lightglue_posenc(w, _t(kpts1))
        states = []
        for i in range(n_layers):
            x0 = _lg_self(w, f"transformers.{i}.self_attn", x0, e0)
            x1 = _lg_self(w, f"transformers.{i}.self_attn", x1, e1)
            x0, x1 = _lg_cross(w, f"transformers.{i}.cross_attn", x0, x1)
            if return_states:
                states.append((x0.numpy().copy(), x1.numpy().copy()))
        a = f"log_assignment.{n_layers - 1}"
        md0 = Fn.linear(x0, _t(w[a + ".final_proj.weight"]), _t(w[a + ".final_proj.bias"]))
        md1 = Fn.linear(x1, _t(w[a + ".final_proj.weight"]), _t(w[a + ".final_proj.bias"]))
        d = md0.shape[-1]
        md0, md1 = md0 / d ** 0.25, md1 / d ** 0.25
        sim = md0 @ md1.t()
        z0 = Fn.linear(x0, _t(w[a + ".matchability.weight"]), _t(w[a + ".matchability.bias"]))
        z1 = Fn.linear(x1, _t(w[a + ".matchability.weight"]), _t(w[a + ".matchability.bias"]))
        cert = Fn.logsigmoid(z0) + Fn.logsigmoid(z1).t()
        scores = Fn.log_softmax(sim, 1) + Fn.log_softmax(sim, 0) + cert
    if return_states:
        return scores.numpy(), states
    return scores.numpy()


# ------------------------------------------------------------------ SuperGlue
def _conv1d(w: W, name: str, x):          # x [C, N]
    return _t(w[name + ".weight"]) @ x + _t(w[name + ".bias"])[:, None]


def _sg_attention(w: W, g: str, x, src, heads=4):
    d = x.shape[0]
    dim = d // heads
    q = _conv1d(w, g + ".attn.proj.0", x).view(dim, heads, -1)
    k = _conv1d(w, g + ".attn.proj.1", src).view(dim, heads, -1)
    v = _conv1d(w, g + ".attn.proj.2", src).view(dim, heads, -1)
    sc = torch.einsum("dhn,dhm->hnm", q, k) / dim ** 0.5
    prob = torch.softmax(sc, -1)
    o = torch.einsum("hnm,dhm->dhn", prob, v).contiguous().view(d, -1)
    return _conv1d(w, g + ".attn.merge", o)


def _sg_prop(w: W, g: str, x, src):
    msg = _sg_attention(w, g, x, src)
    h = torch.relu(_conv1d(w, g + ".mlp.0", torch.cat([x, msg], 0)))      # BN folded into mlp.0
    return _conv1d(w, g + ".mlp.3", h)


def sinkhorn_log(scores: torch.Tensor, alpha: torch.Tensor, iters: int):
    m, n = scores.shape
    ms, ns = scores.new_tensor(float(m)), scores.new_tensor(float(n))
    c = torch.cat([torch.cat([scores, alpha.expand(m, 1)], 1),
                   torch.cat([alpha.expand(1, n), alpha.expand(1, 1)], 1)], 0)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(c + v[None, :], dim=1)
        v = log_nu - torch.logsumexp(c + u[:, None], dim=0)
    return c + u[:, None] + v[None, :] - norm


def superglue_forward(w: W, kpts0, sc0, desc0, kpts1, sc1, desc1, n_layers: int = 18, iters: int = 100):
    """kpts [N,2] normalised, sc [N], desc [N,256] (row-major per keypoint) -> scores [N0+1,N1+1]
    (binding A.5; consumed by decode at src/super_glue.cpp:447-453)."""
    with torch.no_grad():
        def kenc(k, s):
            x = torch.cat([_t(k).t(), _t(s)[None]], 0)                     # [3,N]
            for i in range(5):
                x = _conv1d(w, f"kenc.encoder.{i}", x)
                if i < 4:
                    x = torch.relu(x)
            return x
        d0 = _t(desc0).t() + kenc(kpts0, sc0)
        d1 = _t(desc1).t() + kenc(kpts1, sc1)
        for i in range(n_layers):
            g = f"gnn.layers.{i}"
            if i % 2 == 1:
                s0, s1 = d1, d0
            else:
                s0, s1 = d0, d1
            de0, de1 = _sg_prop(w, g, d0, s0), _sg_prop(w, g, d1, s1)
            d0, d1 = d0 + de0, d1 + de1
        m0, m1 = _conv1d(w, "final_proj", d0), _conv1d(w, "final_proj", d1)
        scores = (m0.t() @ m1) / 256 ** 0.5
        z = sinkhorn_log(scores, _t(w["bin_score"]).reshape(1, 1), iters)
    return z.numpy()


# ------------------------------------------------------------------ PLNet stage 1 (restated)
def plnet_s1_forward(w: W, juncs, lines_pred, idx_pairs, inverse, iskeep_index, loi, loi_thin, loi_aux):
    """Restatement of output/plnet_s1.onnx as decoded in SURVEY.md B.4 — checked against the real graph by
    tests/test_oracle_plnet_s1.py via oracle/onnx_run.py.  Shapes: juncs [300,2], lines_pred [49152,4],
    idx_pairs [M2,2], inverse [M1], iskeep_index [M1], loi [128,128,128], thin/aux [4,128,128].
    Returns lines_adjusted [M2,4], scores_line [M2]."""
    with torch.no_grad():
        juncs = _t(juncs); lines_pred = _t(lines_pred)
        idx = torch.as_tensor(np.asarray(idx_pairs), dtype=torch.long)
        inv = torch.as_tensor(np.asarray(inverse), dtype=torch.long)
        keep = torch.as_tensor(np.asarray(iskeep_index), dtype=torch.long)
        m2 = idx.shape[0]
        la = torch.cat([juncs[idx[:, 0]], juncs[idx[:, 1]]], 1)            # [M2,4]
        # graph: ScatterElements(zeros, reversed(inverse), reversed(arange)) executed sequentially
        # => perm[u] = FIRST k with inverse[k] == u  (nodes 12-35 of plnet_s1.onnx)
        inv_np = np.asarray(inverse, dtype=np.int64)
        perm_np = np.zeros(max(len(inv_np), m2), dtype=np.int64)
        perm_np[inv_np[::-1]] = np.arange(len(inv_np))[::-1]
        perm = torch.from_numpy(perm_np[:m2].copy())
        li = lines_pred[keep[perm]]                                         # [M2,4]

        def bil(fm, x, y):
            c, h, wd = fm.shape
            px, py = x - 0.5, y - 0.5
            x0 = px.floor().clamp(0, wd - 1); y0 = py.floor().clamp(0, h - 1)
            x1 = (x0 + 1).clamp(0, wd - 1); y1 = (y0 + 1).clamp(0, h - 1)
            x0l, y0l, x1l, y1l = x0.long(), y0.long(), x1.long(), y1.long()
            return (fm[:, y0l, x0l] * (y1 - py) * (x1 - px) + fm[:, y1l, x0l] * (py - y0) * (x1 - px)
                    + fm[:, y0l, x1l] * (y1 - py) * (px - x0) + fm[:, y1l, x1l] * (py - y0) * (px - x0))
        loi = _t(loi); thin = _t(loi_thin); aux = _t(loi_aux)
        e1 = bil(loi, la[:, 0], la[:, 1]).t()
        e2 = bil(loi, la[:, 2], la[:, 3]).t()
        t = torch.linspace(0, 1, 32)[1:-1]

        def along(fm, ln):
            u, v = ln[:, 0:2], ln[:, 2:4]
            pts = u[:, None, :] * t[None, :, None] + v[:, None, :] * (1 - t)[None, :, None]   # [M2,30,2]
            s = bil(fm, pts[..., 0].reshape(-1), pts[..., 1].reshape(-1))                      # [4, M2*30]
            return s.reshape(4, m2, 30)
        p_thin = along(thin, la).permute(1, 0, 2).reshape(m2, 120)        # channel-major: c*30 + j
        p_aux = along(aux, li).permute(1, 0, 2).reshape(m2, 120)
        x = torch.cat([e1, e2, p_thin, p_aux], 1)                          # [M2,496]

        def lin(name, v):
            return Fn.linear(v, _t(w[name + ".weight"]), _t(w[name + ".bias"]))
        h = lin("fc2.4", torch.relu(lin("fc2.2", torch.relu(lin("fc2.0", x)))))
        h = h + torch.relu(lin("fc2_res.0", torch.cat([p_thin, p_aux], 1)))
        sc = torch.softmax(lin("fc2_head", h), -1)[:, 1]
    return la.numpy(), sc.numpy()
