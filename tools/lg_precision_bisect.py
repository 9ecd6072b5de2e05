#!/usr/bin/env python3
"""Where does the LightGlue 2-byte error come from?  CPU-only emulation of the device's rounding points.

The HIP matcher keeps the residual stream in fp32 and rounds to 2 bytes (bf16 or fp16) exactly at: the packed weights (`w`),
the token shadow that feeds every projection (`xb`), q/k after rotary (`qk`), v (`v`), the soft-max numerators fed to P.V (`p`),
the attention output (`o`), the out-projection message (`msg`), the FFN hidden state before LayerNorm (`hpre`, four-launch
form) and after GELU (`h`), and the final projection (`md`).  This script replays the fp32 oracle with those roundings
switched on one at a time / all together and prints the log-assignment error against the un-rounded oracle — it reproduced
the device's measured round-1 numbers (bf16 0.41 max / 0.07 mean, fp16 0.056 / 0.009 at N = 400, 9 layers) and is what the
decision "matcher storage = fp16" (include/airfe.h: matcher_precision) rests on.  Test/analysis infrastructure only.

    python tools/lg_precision_bisect.py [--n 400] [--plain]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

from oracle.ref_nets import _rope, _t, lightglue_forward, lightglue_posenc  # noqa: E402

ALL = ("w", "xb", "qk", "v", "p", "o", "msg", "hpre", "h", "md")


def lightglue_forward_q(w, kpts0, desc0, kpts1, desc1, n_layers=9, fmt="bf16", points=ALL, heads=4):
    dt = torch.bfloat16 if fmt == "bf16" else torch.float16

    def q(name, t):
        return t.to(dt).float() if name in points else t

    def lin(x, p):
        return Fn.linear(x, q("w", _t(w[p + ".weight"])), _t(w[p + ".bias"]))

    def ffn(p, xq, msg):
        h = q("hpre", lin(torch.cat([xq, msg], -1), p + ".ffn.0"))
        h = Fn.layer_norm(h, (512,), _t(w[p + ".ffn.1.weight"]), _t(w[p + ".ffn.1.bias"]), eps=1e-5)
        return lin(q("h", Fn.gelu(h)), p + ".ffn.3")

    def attn(qh, kh, vh):
        s = qh @ kh.transpose(-1, -2) / 8.0
        pr = torch.exp(s - s.max(-1, keepdim=True).values)
        return (q("p", pr) @ vh) / pr.sum(-1, keepdim=True)

    with torch.no_grad():
        x = [_t(desc0), _t(desc1)]
        e = [lightglue_posenc(w, _t(kpts0)), lightglue_posenc(w, _t(kpts1))]
        for i in range(n_layers):
            p = f"transformers.{i}.self_attn"
            for s in range(2):
                xq = q("xb", x[s])
                qkv = lin(xq, p + ".Wqkv").unflatten(-1, (heads, -1, 3)).transpose(0, 1)
                qq = q("qk", _rope(e[s][:, None], qkv[..., 0])); kk = q("qk", _rope(e[s][:, None], qkv[..., 1]))
                ctx = q("o", attn(qq, kk, q("v", qkv[..., 2])).transpose(0, 1).flatten(start_dim=-2))
                x[s] = x[s] + ffn(p, xq, q("msg", lin(ctx, p + ".out_proj")))
            p = f"transformers.{i}.cross_attn"
            xq = [q("xb", x[0]), q("xb", x[1])]
            qk = [q("qk", lin(t, p + ".to_qk").unflatten(-1, (heads, -1)).transpose(0, 1)) for t in xq]
            v = [q("v", lin(t, p + ".to_v").unflatten(-1, (heads, -1)).transpose(0, 1)) for t in xq]
            for s in range(2):
                ctx = q("o", attn(qk[s], qk[1 - s], v[1 - s]).transpose(0, 1).flatten(start_dim=-2))
                x[s] = x[s] + ffn(p, xq[s], q("msg", lin(ctx, p + ".to_out")))
        a = f"log_assignment.{n_layers - 1}"
        wf = q("w", _t(w[a + ".final_proj.weight"]) * 0.25); bf = _t(w[a + ".final_proj.bias"]) * 0.25
        md = [q("md", Fn.linear(q("xb", t), wf, bf)) for t in x]
        sim = md[0] @ md[1].t()
        z = [Fn.linear(t, _t(w[a + ".matchability.weight"]), _t(w[a + ".matchability.bias"])) for t in x]
        scores = Fn.log_softmax(sim, 1) + Fn.log_softmax(sim, 0) + Fn.logsigmoid(z[0]) + Fn.logsigmoid(z[1]).t()
    return scores.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--plain", action="store_true", help="the unstructured Kaiming draw of round 1")
    args = ap.parse_args()
    from airslam_amd import weights
    from oracle import ref_post
    from planted import normalised, planted_pair
    w = weights.synthetic_lightglue(1234, structured=not args.plain)
    f0, f1 = planted_pair(args.n, args.n, args.n * 4)
    a, b = normalised(f0)[:, 1:], normalised(f1)[:, 1:]
    ka = (a[:, :2], a[:, 2:], b[:, :2], b[:, 2:])
    ref = lightglue_forward(w, *ka)
    ridx, _ = ref_post.filter_matches(ref, 0.1)
    print(f"oracle: {len(ridx)} matches; emulation with nothing rounded differs by {np.abs(lightglue_forward_q(w, *ka, points=()) - ref).max():.2e}")
    for fmt in ("bf16", "fp16"):
        s = lightglue_forward_q(w, *ka, fmt=fmt)
        idx, _ = ref_post.filter_matches(s, 0.1)
        same = {tuple(p) for p in idx} == {tuple(p) for p in ridx}
        print(f"{fmt} every point : max {np.abs(s - ref).max():.4f} mean {np.abs(s - ref).mean():.4f}  matches {len(idx)} identical set: {same}")
    for pt in ALL:
        s = lightglue_forward_q(w, *ka, fmt="bf16", points=(pt,))
        print(f"bf16 only {pt:5s}: max {np.abs(s - ref).max():.4f} mean {np.abs(s - ref).mean():.4f}")


if __name__ == "__main__":
    main()
