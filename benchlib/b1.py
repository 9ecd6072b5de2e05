"""--workload b1: what ONE stereo keyframe costs through the batch-1 host API — the regime AirSLAM's feature thread runs in (map_builder.cc:83-109)."""
import json
import time

import numpy as np

from . import common as cm


def run(args, rank, world, local, dev):
    """--workload b1: what ONE stereo keyframe costs through the batch-1 host API — the regime AirSLAM's feature thread runs in
    (map_builder.cc:83-109: Detect(left, right, features, lines, junctions) + MatchingPoints, one pair at a time, host buffers in and out).
    Prints p50 / p99 / mean of the keyframe as one call (airfe_stereo_keyframe), as the reference's two calls and as three (infer, infer, match);
    `value` = 1000 / p50 of the one-call form (pairs/s of a single stream)."""
    from airslam_amd import api, synth, weights
    H, W, K = args.height, args.width, args.max_keypoints
    sg = args.matcher == "superglue"
    mw = weights.synthetic_superglue(1234) if sg else weights.synthetic_lightglue(1234)
    plnet = args.detector == "plnet"
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234) if plnet else weights.synthetic_superpoint(1234),
                      plnet_s1=cm.S1_PACK if plnet else None,
                      device=local, precision=1 if args.dtype == "fp16" else 0, matcher_precision=1 if args.matcher_dtype == "fp16" else 0,
                      max_batch=2, enc_chunk=2, max_keypoints=K, image_width=W, image_height=H, matcher=1 if sg else 0, tuning=args.tuning,
                      **(dict(superglue=mw) if sg else dict(lightglue=mw)))
    det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, W, H, 1 if sg else 0)
    pairs = [synth.stereo_pair(H, W, 1000 + i) for i in range(8)]
    t_l, t_r, t_m, t_k, t_d2, t_m2, t_t1, t_t2, t_kt1, t_kt2, nmatch, nlines = [], [], [], [], [], [], [], [], [], [], [], []
    fused = plnet and not sg                                       # airfe_stereo_keyframe: the PLNet + LightGlue keyframe (map_builder.cc:85-86)
    equal = True                                                   # the one- / two- / three-call forms return the same counts (reported on the line, not asserted: ADVICE r04)
    for i in range(args.warmup + args.steps):
        left, right = pairs[i % len(pairs)]
        acc = []
        t0 = time.perf_counter()
        if plnet:
            ok, fl, jl = det.DetectLines(left, None, acc, junction_detection=True)       # left: points + lines + junctions
            t1 = time.perf_counter()
            ok2, fr, _ = det.DetectLines(right, None, [], junction_detection=False)      # right: no junctions (feature_detector.cc:100-101)
        else:
            ok, fl = det.Detect(left)
            t1 = time.perf_counter()
            ok2, fr = det.Detect(right)
        t2 = time.perf_counter()
        n, _ = pm.MatchingPoints(fl, fr)
        t3 = time.perf_counter()
        if fused:
            k = ctx.stereo_keyframe(left, right)                                         # the same keyframe as ONE call
            t4 = time.perf_counter()
            okk, fl2, fr2, _ = det.DetectKeyframe(left, right, [], [])                   # ... and as the reference's own two calls
            t5 = time.perf_counter()
            n2, _ = pm.MatchingPoints(fl2, fr2)
            t6 = time.perf_counter()
            equal = equal and (len(k["idx"]) == n == n2 and len(k["linesL"]) == len(acc))
            # the normal-frame step (map_builder.cc:94-101): Detect(image, features) + MatchingPoints(last keyframe, features) — as two calls, as one
            t7 = time.perf_counter()
            if i % 8 == 0:
                kf_ref = fl2                                                             # a new "last keyframe" every 8 frames
            okt, ft = det.Detect(right)
            nt, _ = pm.MatchingPoints(kf_ref, ft)
            t8 = time.perf_counter()
            _, tidx, _ = ctx.track_frame(right, ref_feat=kf_ref.T if i % 8 == 0 else None)           # its features go up once, then stay on the device
            t9 = time.perf_counter()
            equal = equal and len(tidx) == nt
            # a keyframe candidate also runs the temporal match (map_builder.cc:96): one call with both pairs in ONE LightGlue forward, against
            # the one-call keyframe + a MatchingPoints call
            kt = ctx.stereo_keyframe(left, right, track=True)                               # (reference = the features uploaded above)
            t10 = time.perf_counter()
            k2 = ctx.stereo_keyframe(left, right)
            nt2, _ = pm.MatchingPoints(kf_ref, np.asfortranarray(k2["featL"].T))
            t11 = time.perf_counter()
            equal = equal and (len(kt["track_idx"]) == nt2 and len(kt["idx"]) == len(k2["idx"]))
        if i >= args.warmup:
            t_l.append(t1 - t0); t_r.append(t2 - t1); t_m.append(t3 - t2); nmatch.append(n); nlines.append(len(acc))
            if fused:
                t_k.append(t4 - t3); t_d2.append(t5 - t4); t_m2.append(t6 - t5); t_t2.append(t8 - t7); t_t1.append(t9 - t8); t_kt1.append(t10 - t9); t_kt2.append(t11 - t10)
    pair = np.array(t_l) + np.array(t_r) + np.array(t_m)

    def pct(a):
        a = np.asarray(a) * 1e3
        return {"p50": float(np.percentile(a, 50)), "p99": float(np.percentile(a, 99)), "mean": float(a.mean())}

    lat = {"pair": pct(t_k) if fused else pct(pair), "three_calls": {"pair": pct(pair), "detect_left": pct(t_l), "detect_right": pct(t_r), "match": pct(t_m)}}
    if fused:
        lat["two_calls"] = {"pair": pct(np.array(t_d2) + np.array(t_m2)), "detect_stereo": pct(t_d2), "match": pct(t_m2)}
        lat["tracked_frame"] = {"one_call": pct(t_t1), "two_calls": pct(t_t2)}        # airfe_track_frame vs Detect + MatchingPoints (points only)
        # a keyframe WITH its temporal match (map_builder.cc:85-86 + :96): airfe_stereo_keyframe_tracked vs airfe_stereo_keyframe + MatchingPoints
        lat["keyframe_with_temporal_match"] = {"one_call": pct(t_kt1), "keyframe_call_plus_match_call": pct(t_kt2)}
    head = lat["pair"]["p50"]
    out = cm.line(
        args,
        metric="batch-1 stereo keyframe latency, host images in / host matrices out, PCIe and synchronisation included ("
               + ("PLNet points + lines, junctions on the left" if plnet else "SuperPoint") + " x2 + " + ("SuperGlue" if sg else "LightGlue") + "): "
               + ("ONE call (airfe_stereo_keyframe = map_builder.cc:85-86); latency_ms also has the same keyframe as the reference's two calls "
                  "(stereo Detect overload + MatchingPoints) and as three (PLNet::infer twice + MatchingPoints), identical results" if fused
                  else "three reference-shaped calls (detect, detect, MatchingPoints)"),
        value=1e3 / head, unit="pairs/s", world=1, steps=args.steps, warmup=args.warmup, ms_per_step=head,
        config={"workload": f"ONE synthetic {W}x{H} stereo pair per step through the batch-1 host API (airslam_amd.api over the C ABI's host-buffer entries), "
                            f"max_keypoints={K}; seeded synthetic weights (reference ONNX files are absent)",
                "matches_mean": float(np.mean(nmatch)), "lines_mean_left": float(np.mean(nlines)), "detector": args.detector, "matcher": args.matcher},
        latency_ms=lat, call_forms_agree=bool(equal))
    if rank == 0:
        print(json.dumps(out))
    ctx.close()
