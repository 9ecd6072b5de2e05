#!/usr/bin/env python3
"""What 2-byte MFMA operands cost PLNet stage 1, measured on the CPU before a kernel is written: the oracle chain
(oracle/ref_chain.py, fp32) up to the stage-0 tensors, then the stage-1 head (the REAL weights of output/plnet_s1.onnx) once in fp32 and once with every matrix
product's OPERANDS rounded to fp16 (weights, the 496 sampled features, both hidden layers; fp32 accumulation, fp32 bias / ReLU / residual / 2-way head) — the
arithmetic of `cfg.line_precision = 1` (kernels_ext.hip, plnet_s1_h_kernel) and of the reference's own engine (BuilderFlag::kFP16, src/plnet.cpp:216).
    python tools/s1_fp16_emulation.py [seed ...]      -> one line per image + a summary (profiles/r05_s1_fp16_emulation.txt)"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as Fn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from airslam_amd import synth, weights  # noqa: E402
from oracle import ref_chain, ref_nets, ref_post  # noqa: E402


def h16(t):
    return t.to(torch.float16).to(torch.float32)


def s1_scores_split(w, x, pta):
    """`cfg.line_precision = 3` as the kernels compute it (plnet_s1h_kernel, s1h_junc_proj_kernel): every operand as a (hi, lo) PAIR of fp16 values
    (hi = fp16(v), lo = fp16((v - hi) * 2^11): 22 bits of mantissa, lo never a denormal), every product as three fp16 MFMAs hi.hi + 2^-11 (hi.lo + lo.hi) with fp32
    accumulation (the lo.lo term, 2^-22 relative, is dropped) — the junction projections (the 256 LOI columns of fc2.0) included."""
    def sp(v):
        hi = h16(v)
        return hi, h16((v - hi) * 2048.0)

    def lin(name, v):
        W, b = torch.from_numpy(w[name + ".weight"]), torch.from_numpy(w[name + ".bias"])
        vh, vl = sp(v)
        Wh, Wl = sp(W)
        return b + vh @ Wh.t() + (vh @ Wl.t() + vl @ Wh.t()) * (1.0 / 2048.0)
    h = lin("fc2.4", torch.relu(lin("fc2.2", torch.relu(lin("fc2.0", x)))))
    h = h + torch.relu(lin("fc2_res.0", pta))
    W, b = torch.from_numpy(w["fc2_head.weight"]), torch.from_numpy(w["fc2_head.bias"])
    return torch.softmax(h @ W.t() + b, -1)[:, 1].numpy()


def s1_scores(w, x, pta, half, proj_f32=False):
    """x [M,496] fp32 features -> scores_line; half: operands through fp16.  proj_f32: the 256 LOI columns of fc2.0 stay fp32 (the device applies them once
    per junction in fp32: s1_junc_proj_kernel)"""
    q = h16 if half else (lambda v: v)

    def lin(name, v, keep32_cols=0):
        W, b = torch.from_numpy(w[name + ".weight"]), torch.from_numpy(w[name + ".bias"])
        if keep32_cols:
            return v[:, :keep32_cols] @ W[:, :keep32_cols].t() + q(v[:, keep32_cols:]) @ q(W[:, keep32_cols:]).t() + b
        return q(v) @ q(W).t() + b
    h = lin("fc2.4", torch.relu(lin("fc2.2", torch.relu(lin("fc2.0", x, 256 if proj_f32 else 0)))))
    h = h + torch.relu(lin("fc2_res.0", pta))
    W, b = torch.from_numpy(w["fc2_head.weight"]), torch.from_numpy(w["fc2_head.bias"])
    return torch.softmax(h @ W.t() + b, -1)[:, 1].numpy()


def features(s0, pairs, inv, keep):
    """the 496 sampled features of every candidate (the body of ref_nets.plnet_s1_forward up to `x`)"""
    cap = {}
    orig = Fn.linear

    def spy(v, W, b=None):
        if W.shape == (128, 496):
            cap["x"] = v.clone()
        if W.shape == (128, 240):
            cap["pta"] = v.clone()
        return orig(v, W, b)
    Fn.linear = spy
    try:
        la, sc = ref_nets.plnet_s1_forward(S1, s0["juncs_pred"], s0["lines_pred"], pairs, inv, keep, s0["loi_features"][0], s0["loi_features_thin"][0], s0["loi_features_aux"][0])
    finally:
        Fn.linear = orig
    return la, sc, cap["x"], cap["pta"]


if __name__ == "__main__":
    seeds = [int(a) for a in sys.argv[1:]] or [5, 8, 12, 33]
    S1 = weights.load_pack(os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe"))
    sp = weights.synthetic_plnet_s0(1234)
    print("stage-1 weights: max |w| per layer", {k: float(np.abs(v).max()) for k, v in S1.items() if k.endswith("weight")})
    tot = dict(lines=0, flipped=0, flipped_p=0, cand=0)
    worst = 0.0
    split_tot = [0, 0.0]
    for seed in seeds:
        img = synth.gabor_image(480, 752, seed)
        ref = ref_chain.plnet_infer(sp, S1, img)
        s0 = ref["stage0"]
        keep, inv, pairs = ref_post.wireframe_matcher(s0["iskeep"], s0["idx_junc_to_end_min"], s0["idx_junc_to_end_max"])
        la, sc32, x, pta = features(s0, pairs, inv, keep)
        assert np.allclose(sc32, s1_scores(S1, x, pta, False), atol=1e-6)
        out = {}
        for name, kw in (("fp16_all", dict(half=True)), ("fp16_lines_fp32_junctions", dict(half=True, proj_f32=True))):
            sc16 = s1_scores(S1, x, pta, **kw)
            l32, _ = ref_post.line_filter(la, sc32, 4, 0.75, 50.0)
            l16, _ = ref_post.line_filter(la, sc16, 4, 0.75, 50.0)
            err = np.abs(sc16 - sc32)
            k32, k16 = sc32 > 0.75, sc16 > 0.75
            flips = np.nonzero(k32 != k16)[0]
            out[name] = dict(score_err_max=float(err.max()), score_err_mean=float(err.mean()), lines_fp32=len(l32), lines_fp16=len(l16), candidates=len(sc32),
                             score_flips=len(flips), flip_margins=[round(float(abs(sc32[i] - 0.75)), 5) for i in flips],
                             flips_beyond_2x_err=int(sum(abs(sc32[i] - 0.75) > 2 * err.max() for i in flips)), x_absmax=float(x.abs().max()))
        scs = s1_scores_split(S1, x, pta)
        es = np.abs(scs - sc32)
        out["split_fp16_pairs"] = dict(score_err_max=float(es.max()), score_flips=int(((scs > 0.75) != (sc32 > 0.75)).sum()),
                                       lines=len(ref_post.line_filter(la, scs, 4, 0.75, 50.0)[0]))
        split_tot[0] += out["split_fp16_pairs"]["score_flips"]; split_tot[1] = max(split_tot[1], out["split_fp16_pairs"]["score_err_max"])
        print(f"seed {seed}: " + "; ".join(f"{k}: {v}" for k, v in out.items()))
        o = out["fp16_lines_fp32_junctions"]
        tot["lines"] += o["lines_fp32"]; tot["flipped"] += abs(o["lines_fp16"] - o["lines_fp32"]) ; tot["flipped_p"] += o["score_flips"]; tot["cand"] += o["candidates"]
        worst = max(worst, o["score_err_max"])
    print(f"SUMMARY fp16 operands (junction projections fp32): {tot['flipped_p']} of {tot['cand']} candidates change side of the 0.75 threshold "
          f"({tot['lines']} lines kept in fp32: {100.0 * tot['flipped_p'] / max(tot['lines'], 1):.2f} %), largest score error {worst:.5f}")
    print(f"SUMMARY split fp16 pairs (hi.hi + hi.lo + lo.hi on the 2-byte matrix pipe, fp32 accumulate): {split_tot[0]} candidates change side, largest score error {split_tot[1]:.2e}")
