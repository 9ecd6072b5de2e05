"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (airslam_amd/).

PyTorch-CPU fp32 restatement of the network bodies the reference executes through
TensorRT engines built from ONNX files (call sites src/plnet.cpp:233,510;
src/super_point.cpp:133; src/light_glue.cpp:159; src/super_glue.cpp:185).

SuperPoint / LightGlue / SuperGlue: their ONNX files are absent from /root/reference (.MISSING_LARGE_BLOBS), onnxruntime is
not installed, and the reference has no tests.  The bodies below restate the PUBLISHED architectures (SuperPoint v1 — DeTone et
al.; LightGlue — Lindenberger et al., cvg/LightGlue lightglue.py; SuperGlue — Sarlin et al., magicleap/
SuperGluePretrainedNetwork), with tensor names/shapes anchored on the reference's bindings, and are
PINNED TO INDEPENDENT IMPLEMENTATIONS: Hugging Face transformers 5.15.0's ports of the same three networks, loaded with the same
seeded weights (oracle/hf_pin.py holds the layout mappings), agree with the functions below — SuperPoint's suppressed score map
and dense descriptors bit for bit, LightGlue's log-assignment within 1.1e-4 on a range of 52, SuperGlue's optimal-transport
matrix within 3.4e-5 on a range of 65 (tests/test_oracle_hf_pin_cpu.py; the HIP library against HF directly:
tests/test_gpu_hf_pin.py).  What stays unpinned is the stage-0 LINE head of PLNet (plnet_s0_lines: plnet_s0.onnx is absent and
no second implementation of it exists here).
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
def superpoint_trunk(w: W, x: torch.Tensor, taps: dict = None) -> torch.Tensor:
    """x [B,1,H,W] in [0,1] -> conv4b activations [B,128,H/8,W/8].  taps (optional dict): receives the conv3a activations
    [B,128,H/4,W/4], the input of the PLNet line branch (plnet_s0_lines(f3a=...): one trunk pass per image, as the stage-0 engine has)."""
    def c(name, t):
        return Fn.relu(Fn.conv2d(t, _t(w[name + ".weight"]), _t(w[name + ".bias"]), padding=1))
    x = c("conv1a", x); x = c("conv1b", x); x = Fn.max_pool2d(x, 2, 2)
    x = c("conv2a", x); x = c("conv2b", x); x = Fn.max_pool2d(x, 2, 2)
    x = c("conv3a", x)
    if taps is not None:
        taps["conv3a"] = x
    x = c("conv3b", x); x = Fn.max_pool2d(x, 2, 2)
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


# ------------------------------------------------------------------ PLNet stage-0 line branch
def plnet_s0_lines(w: W, x: np.ndarray, topk: int = 300, scale: float = 5.0, j2l: float = 10.0, f3a: torch.Tensor = None):
    """The stage-0 LINE branch: x [H,W] float32 in [0,1] (512x512) -> dict with the Appendix A.1 tensors
    (`juncs_pred` [300,2], `lines_pred` [3*128*128,4], `iskeep` / `idx_junc_to_end_min` / `idx_junc_to_end_max` [1,3,128,128],
    `loi_features` [1,128,128,128], `loi_features_thin` / `_aux` [1,4,128,128]) plus `jloc` / `joff` maps.
    PARITY UNPINNED: plnet_s0.onnx is absent; this restates the PUBLISHED HAWPv3 decoding (hawp/fsl/model/detector.py:
    hafm_decoding, non_maximum_suppression, get_junctions, wireframe_matcher) on the shared trunk with the channel layout of
    airslam_amd.weights.plnet_line_spec.  The reference consumes these tensors at src/plnet.cpp:453-509."""
    with torch.no_grad():
        def c(name, t):
            return Fn.relu(Fn.conv2d(t, _t(w[name + ".weight"]), _t(w[name + ".bias"]), padding=1))
        if f3a is None:
            f = _t(x)[None, None]
            f = c("conv1a", f); f = c("conv1b", f); f = Fn.max_pool2d(f, 2, 2)
            f = c("conv2a", f); f = c("conv2b", f); f = Fn.max_pool2d(f, 2, 2)
            f = c("conv3a", f)                                               # [1,128,128,128]
        else:
            f = f3a                                                          # the point branch's own conv3a activations (superpoint_trunk taps)
        f = c("line.conv1", f)
        o = Fn.conv2d(f, _t(w["line.head.weight"])[:, :, None, None], _t(w["line.head.bias"]))[0]    # [145,128,128]
        loi, h = o[:128], o[128:]
        md, dis, res = torch.sigmoid(h[0:3]), torch.sigmoid(h[3:4]), torch.sigmoid(h[4:5])
        jloc = torch.softmax(h[5:7], 0)[1]
        joff = torch.sigmoid(h[7:9]) - 0.5
        thin, aux = h[9:13], h[13:17]
        hh, ww = jloc.shape
        # hafm_decoding (residual sign pad -1, 0, +1)
        y0, x0 = torch.meshgrid(torch.arange(hh, dtype=torch.float32), torch.arange(ww, dtype=torch.float32), indexing="ij")
        sign = torch.tensor([-1.0, 0.0, 1.0]).reshape(3, 1, 1)
        d = (dis + res * sign).clamp(0.0, 1.0)                               # [3,128,128]
        pi = torch.tensor(np.float32(np.pi))
        md_un = (md[0] - 0.5) * pi * 2.0
        st_un = md[1] * pi / 2.0
        ed_un = -md[2] * pi / 2.0
        cs, ss, yst, yed = md_un.cos(), md_un.sin(), st_un.tan(), ed_un.tan()
        xs = ((cs - ss * yst) * d * scale + x0).clamp(0, ww - 1)
        ys = ((ss + cs * yst) * d * scale + y0).clamp(0, hh - 1)
        xe = ((cs - ss * yed) * d * scale + x0).clamp(0, ww - 1)
        ye = ((ss + cs * yed) * d * scale + y0).clamp(0, hh - 1)
        lines = torch.stack([xs, ys, xe, ye], -1).reshape(-1, 4)             # [3*128*128, 4]
        juncs = junctions_topk(jloc.numpy(), joff.numpy(), topk)
        keep, imin, imax = j2l_match(lines.numpy(), juncs, j2l)
    return dict(juncs_pred=juncs, lines_pred=lines.numpy(), iskeep=keep.reshape(1, 3, hh, ww),
                idx_junc_to_end_min=imin.reshape(1, 3, hh, ww), idx_junc_to_end_max=imax.reshape(1, 3, hh, ww),
                loi_features=loi.numpy()[None], loi_features_thin=thin.numpy()[None], loi_features_aux=aux.numpy()[None],
                jloc=jloc.numpy(), joff=joff.numpy())


def junctions_topk(jloc: np.ndarray, joff: np.ndarray, topk: int = 300) -> np.ndarray:
    """non_maximum_suppression (a * (a == max_pool2d(a, 3, 1, 1))) + get_junctions (top-k scores; x = col + joff_x + 0.5,
    y = row + joff_y + 0.5).  Ties: torch.topk leaves their order unspecified — fixed here as ascending raster index."""
    h, w = jloc.shape
    pad = np.full((h + 2, w + 2), -np.inf, np.float32)
    pad[1:-1, 1:-1] = jloc
    mp = np.max(np.stack([pad[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)]), 0)
    a = np.where(jloc == mp, jloc, np.float32(0)).astype(np.float32).reshape(-1)
    order = np.argsort(-a.astype(np.float64), kind="stable")[:topk]
    out = np.zeros((topk, 2), np.float32)
    valid = a[order] > 0                    # the device emits only suppressed maxima with a positive score (always >= topk of them in practice)
    idx = order[valid]
    out[:idx.size, 0] = ((idx % w).astype(np.float32) + joff[0].reshape(-1)[idx]).astype(np.float32) + np.float32(0.5)
    out[:idx.size, 1] = ((idx // w).astype(np.float32) + joff[1].reshape(-1)[idx]).astype(np.float32) + np.float32(0.5)
    return out


def j2l_match(lines: np.ndarray, juncs: np.ndarray, thr: float = 10.0):
    """HAWP wireframe_matcher: squared distances of both endpoints to every junction (float32, products rounded separately),
    first minimum; idx_min / idx_max; iskeep = (min < max) & (d1 < thr) & (d2 < thr).  Returns three float32 arrays [n]."""
    l = lines.astype(np.float32); j = juncs.astype(np.float32)
    n = l.shape[0]
    i1 = np.zeros(n, np.int64); i2 = np.zeros(n, np.int64)
    c1 = np.full(n, np.inf, np.float32); c2 = np.full(n, np.inf, np.float32)
    for k in range(j.shape[0]):
        ax = (l[:, 0] - j[k, 0]).astype(np.float32); ay = (l[:, 1] - j[k, 1]).astype(np.float32)
        bx = (l[:, 2] - j[k, 0]).astype(np.float32); by = (l[:, 3] - j[k, 1]).astype(np.float32)
        d1 = ((ax * ax).astype(np.float32) + (ay * ay).astype(np.float32)).astype(np.float32)
        d2 = ((bx * bx).astype(np.float32) + (by * by).astype(np.float32)).astype(np.float32)
        u1 = d1 < c1; u2 = d2 < c2
        c1[u1] = d1[u1]; i1[u1] = k
        c2[u2] = d2[u2]; i2[u2] = k
    lo, hi = np.minimum(i1, i2), np.maximum(i1, i2)
    keep = ((lo < hi) & (c1 < np.float32(thr)) & (c2 < np.float32(thr))).astype(np.float32)
    return keep, lo.astype(np.float32), hi.astype(np.float32)


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
