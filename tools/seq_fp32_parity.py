#!/usr/bin/env python3
"""BASELINE.json configs[1] / SURVEY cfg 2: a stereo SEQUENCE through PLNet + LightGlue in fp32 (`precision = 2`, `matcher_precision = 2`) on
one MI355X, every output diffed against the fp32 CPU oracle (oracle/ref_chain.py + ref_nets.lightglue_forward + ref_post.filter_matches ≙
map_builder.cc:85-86 per frame: Detect(left, right, features, lines, junctions) + MatchingPoints).  EuRoC MH_01 is external and the reference's
ONNX files are absent, so the sequence is synthetic (airslam_amd.synth.stereo_sequence) and both sides run the same seeded weights (stage 1:
the reference's real plnet_s1 weights).

    python tools/seq_fp32_parity.py --oracle-only --frames 200 --cache /tmp/airfe_cache/seq_oracle_200.npz     # CPU only: the oracle's outputs
    python tools/seq_fp32_parity.py --frames 200 --cache /tmp/airfe_cache/seq_oracle_200.npz --out profiles/r03_seq_fp32_parity.json   # on the GPU

Without --cache the oracle runs live (about 2 s per frame on 32 host threads).  The oracle is the CHECKER here, as in tests/."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

H, W, K = 480, 752, 400


def oracle_frame(sp, s1, lg, left, right):
    from oracle import ref_chain, ref_nets, ref_post
    a = ref_chain.plnet_infer(sp, s1, left, want_junctions=True, top_k=K)
    b = ref_chain.plnet_infer(sp, s1, right, want_junctions=False, top_k=K)
    fa, fb = a["features"], b["features"]
    na, nb = ref_post.normalize_keypoints(fa, W, H, 0.5), ref_post.normalize_keypoints(fb, W, H, 0.5)
    sc = ref_nets.lightglue_forward(lg, na[:, 1:3], na[:, 3:], nb[:, 1:3], nb[:, 3:])
    m, ms = ref_post.filter_matches(sc, 0.1)
    return dict(fl=fa.astype(np.float32), fr=fb.astype(np.float32), ll=a["lines"], lr=b["lines"], jl=a["junctions"].astype(np.float32),
                m=np.asarray(m, np.int32).reshape(-1, 2), ms=np.asarray(ms, np.float32))


def line_hits(a, b, tol=1.0):
    if len(a) == 0 or len(b) == 0:
        return float(len(a) == len(b))
    pa, pb = a.reshape(-1, 1, 2, 2), b.reshape(1, -1, 2, 2)
    d0 = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 0], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 1], axis=-1))
    d1 = np.maximum(np.linalg.norm(pa[:, :, 0] - pb[:, :, 1], axis=-1), np.linalg.norm(pa[:, :, 1] - pb[:, :, 0], axis=-1))
    return float((np.minimum(d0, d1).min(1) <= tol).mean())


def compare(dev, ref):
    """one frame: device outputs vs oracle outputs -> dict of numbers.  Keypoints are paired by POSITION, not by row: inside the top-K two
    keypoints whose scores differ by less than the heat map's tolerance may swap rows, which is not an error (detect_point's order is by
    score); matches are compared as coordinate quadruples for the same reason."""
    out = {}
    for side in ("l", "r"):
        fd, fo = dev["f" + side], ref["f" + side]
        out[f"kp_{side}_count_equal"] = float(fd.shape[0] == fo.shape[0])
        if len(fd) and len(fo):
            d = np.linalg.norm(fd[:, None, 1:3] - fo[None, :, 1:3], axis=2)
            nn = d.argmin(1)
            dist = d[np.arange(len(fd)), nn]
            out[f"kp_{side}_same_rows"] = float(fd.shape == fo.shape and np.array_equal(fd[:, 1:3], fo[:, 1:3]))
            out[f"kp_{side}_within_1px"] = float((dist <= 1.0).mean())
            out[f"kp_{side}_max_px"] = float(dist.max())
            ok = dist <= 1.0
            out[f"score_{side}_max_abs"] = float(np.abs(fd[ok, 0] - fo[nn[ok], 0]).max()) if ok.any() else 0.0
            num = (fd[ok, 3:] * fo[nn[ok], 3:]).sum(1)
            den = np.linalg.norm(fd[ok, 3:], axis=1) * np.linalg.norm(fo[nn[ok], 3:], axis=1)
            out[f"desc_{side}_max_cosine_dist"] = float((1.0 - num / np.maximum(den, 1e-30)).max()) if ok.any() else 0.0
        out[f"lines_{side}_dev"] = len(dev["l" + side]); out[f"lines_{side}_ref"] = len(ref["l" + side])
        out[f"lines_{side}_dev_hit"] = line_hits(dev["l" + side], ref["l" + side]); out[f"lines_{side}_ref_hit"] = line_hits(ref["l" + side], dev["l" + side])
    jd, jo = dev["jl"], ref["jl"]
    out["junc_dev"] = len(jd); out["junc_ref"] = len(jo)
    out["junc_within_1px"] = float((np.linalg.norm(jd[:, None, 1:3] - jo[None, :, 1:3], axis=2).min(1) <= 1.0).mean()) if len(jd) and len(jo) else float(len(jd) == len(jo))

    def quads(r):          # a match as (xl, yl, xr, yr) in quarter pixels: independent of the row order of the keypoints
        m = r["m"]
        if len(m) == 0:
            return {}
        q = np.concatenate([r["fl"][m[:, 0], 1:3], r["fr"][m[:, 1], 1:3]], axis=1).astype(np.float64)
        return {tuple(np.round(x * 4).astype(np.int64).tolist()): float(s) for x, s in zip(q, r["ms"])}
    qd, qo = quads(dev), quads(ref)
    out["matches_dev"] = len(qd); out["matches_ref"] = len(qo)
    out["match_sets_identical"] = float(qd.keys() == qo.keys())
    out["match_jaccard"] = len(qd.keys() & qo.keys()) / max(len(qd.keys() | qo.keys()), 1)
    common = qd.keys() & qo.keys()
    if common:
        out["match_score_max_abs"] = float(max(abs(qd[k] - qo[k]) for k in common))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--seed", type=int, default=4100)
    ap.add_argument("--cache", default=None, help="npz of the oracle's outputs: read if present, written otherwise")
    ap.add_argument("--oracle-only", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from airslam_amd import synth, weights
    sp, lg = weights.synthetic_plnet_s0(1234), weights.synthetic_lightglue(1234)
    s1_path = os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")
    s1 = weights.load_pack(s1_path)
    keys = ("fl", "fr", "ll", "lr", "jl", "m", "ms")
    cache = None
    if args.cache and os.path.exists(args.cache):
        z = np.load(args.cache)
        if int(z["frames"]) >= args.frames and int(z["seed"]) == args.seed:
            cache = z
    refs = []
    t0 = time.time()
    if cache is None:
        import torch
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        for i, (left, right) in enumerate(synth.stereo_sequence(args.frames, H, W, args.seed)):
            refs.append(oracle_frame(sp, s1, lg, left, right))
            if i % 10 == 0:
                print(f"oracle frame {i}: {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
        if args.cache:
            os.makedirs(os.path.dirname(os.path.abspath(args.cache)), exist_ok=True)
            flat = {}
            for i, r in enumerate(refs):          # descriptors in fp16 to keep the file small (cosine ~1e-7 from that); score / x / y stay fp32
                for k in keys:
                    if k in ("fl", "fr", "jl"):
                        flat[f"{k}_{i}_sxy"] = r[k][:, :3].astype(np.float32)
                        flat[f"{k}_{i}_d"] = r[k][:, 3:].astype(np.float16)
                    else:
                        flat[f"{k}_{i}"] = r[k]
            np.savez_compressed(args.cache, frames=args.frames, seed=args.seed, **flat)
    else:
        refs = [{k: (np.concatenate([cache[f"{k}_{i}_sxy"], cache[f"{k}_{i}_d"].astype(np.float32)], axis=1) if k in ("fl", "fr", "jl")
                     else np.asarray(cache[f"{k}_{i}"])) for k in keys} for i in range(args.frames)]
    oracle_s = time.time() - t0
    if args.oracle_only:
        print(f"oracle: {args.frames} frames in {oracle_s:.0f} s")
        return
    from airslam_amd import api
    ctx = api.Context(superpoint=sp, lightglue=lg, plnet_s1=s1_path, precision=2, matcher_precision=2, max_batch=2, enc_chunk=2, max_keypoints=K,
                      image_width=W, image_height=H)
    pm = api.PointMatcher(ctx, W, H, 0)
    rows = []
    t1 = time.time()
    for i, (left, right) in enumerate(synth.stereo_sequence(args.frames, H, W, args.seed)):
        fl, ll, jl = ctx.detect_plnet(left, None, want_junctions=True)
        fr, lr, _ = ctx.detect_plnet(right, None, want_junctions=False)
        _, mm = pm.MatchingPoints(np.asfortranarray(fl.T), np.asfortranarray(fr.T))     # [259, N] as the reference's Eigen matrices
        dev = dict(fl=fl, fr=fr, ll=ll, lr=lr, jl=jl, m=np.asarray([(a, b) for a, b, _ in mm], np.int32).reshape(-1, 2),
                   ms=np.asarray([1.0 - d for _, _, d in mm], np.float32))
        rows.append(compare(dev, refs[i]))
    dev_s = time.time() - t1
    ctx.close()
    summ = {"frames": args.frames, "oracle_from_cache": cache is not None, "device_seconds": dev_s, "oracle_seconds": None if cache is not None else oracle_s}
    for k in sorted({k for r in rows for k in r}):
        v = np.array([r[k] for r in rows if k in r], np.float64)
        summ[k] = {"min": float(v.min()), "mean": float(v.mean()), "max": float(v.max()), "n": int(v.size)}
    txt = json.dumps(summ, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
