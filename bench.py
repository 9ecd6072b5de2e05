#!/usr/bin/env python3
"""Headline benchmark: stereo detect+match pairs/s/GPU @752x480 (BASELINE.json `metric`).

A step = one pass of the hot path as the reference's front end runs it on a stereo keyframe (map_builder.cc:85-86:
`Detect(left, right, features, lines, junctions)` + `MatchingPoints(left, right)`): PLNet on both images — points, the line branch,
wireframe_matcher, stage 1 and the line filter, junctions + descriptors on the left one (feature_detector.cc:94-104) — and one LightGlue
match, at the reference's 512x512 internal resolution (src/plnet.cpp:17-21), over one batch of `--pairs` synthetic stereo pairs that
are already resident in HBM.  `--detector superpoint` is the point-only step (2x SuperPoint-VGG detect + LightGlue: what
`Detect(left, right, features)` runs with use_superpoint = 1); it is also timed in the default run and reported beside the headline.
One process per GPU; ranks shard pairs (weak scaling) and gather their matches to rank 0 every step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

def latest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (counter passes are separate runs: tools/gpu_profile.sh), or None"""
    import glob
    import re
    c = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)) if re.match(r"r\d\d_" + re.escape(suffix) + "$", os.path.basename(f))]
    return sorted(c)[-1] if c else None


def counter_profile(suffix):
    """(document, path, age) of the newest committed profiles/rNN_<suffix>: age = {"profile", "profile_csrc_sha", "tree_csrc_sha", "stale"} — counters measured on other
    kernel sources than the ones this process runs are NOT reported (document = None): a number from an older kernel must never ride on a changed one."""
    from airslam_amd.build import csrc_sha
    pf = latest_profile(suffix)
    if not pf:
        return None, None, None
    with open(pf) as fh:
        doc = json.load(fh)
    now = csrc_sha()
    age = {"profile": "profiles/" + os.path.basename(pf), "profile_csrc_sha": doc.get("csrc_sha"), "tree_csrc_sha": now, "stale": doc.get("csrc_sha") != now}
    return (None if age["stale"] else doc), pf, age


# which roof a stage is priced against: matrix stages by their algorithmic FLOPs against the dense 2-byte MFMA peak, the rest by
# their algorithmic bytes against the HBM peak; plnet_stage1 is a chain of gathers and a small MLP on the 2-byte MFMA (cfg.line_precision = 3): latency-bound, no single roof
STAGE_BOUND = {"conv1_fused": "mfma", "conv3x3_cin64": "mfma", "conv3x3_cin128": "mfma", "head_gemm": "mfma", "lg_gemm": "mfma", "lg_attention": "mfma"}
PEAK_MFMA_F32_TFLOPS = 157.0


DOMINANT_STAGE = "conv1_fused"   # the dominant KERNEL (conv64r_kernel<POOL, FUSE1A>: conv1a + conv1b + pool, ~21 % of a step) is a stage of
                                # its own: its launches keep their HIP events inside the timed region
PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def cpu_baseline(sp, lg, h, w, n_pairs, max_kp, warm=3, s1=None, line_threshold=0.75, line_length_threshold=50.0, gpu_nmatch=None):
    """The CPU oracle (PyTorch-CPU fp32 networks + numpy restatement of the reference's C++ post-processing) timed on
    the host cores, on a bounded sample of the same workload: `warm` untimed pairs, then the MEDIAN per-pair time of
    `n_pairs` pairs (median of >= 20 after 3 warm-ups).  s1 (the stage-1 weights) selects the PLNet step: one trunk
    pass per image feeding the point heads AND the line branch, wireframe_matcher, stage 1, line filter, junctions on the left."""
    from airslam_amd import synth
    from oracle import margins, ref_chain, ref_nets, ref_post
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # more threads than this only adds sync overhead at batch 1
    # the SAME images rank 0 puts on the GPU (synth.stereo_batch(B, h, w, 1000)): its first n_pairs + warm pairs, ready before the clock
    ls, rs = synth.stereo_batch(n_pairs + warm, h, w, 1000)
    pairs = list(zip(ls, rs))
    times, nmatch, nfrag = [], [], []
    for i, (left, right) in enumerate(pairs):
        t0 = time.perf_counter()
        feats = []
        for side, img in enumerate((left, right)):
            x, ws, hs = ref_post.process_image(img)
            if s1 is None:
                heat, desc = ref_nets.superpoint_forward(sp, x[None])
                feats.append(ref_post.keypoints_decoder(ref_post.simple_nms(heat[0], 4), desc[0], 0.004, 4, max_kp, ws, hs))
                continue
            # PLNet::infer end to end (src/plnet.cpp:221-244): one trunk pass feeding the point heads and the line branch, wireframe_matcher,
            # stage 1, line filter; junctions on the left image only (feature_detector.cc:100-101)
            feats.append(ref_chain.plnet_infer(sp, s1, img, want_junctions=side == 0, top_k=max_kp, line_threshold=line_threshold,
                                               line_length_threshold=line_length_threshold)["features"])
        k = 0
        if feats[0].shape[0] and feats[1].shape[0]:
            a = ref_post.normalize_keypoints(feats[0], w, h, 0.5)
            b = ref_post.normalize_keypoints(feats[1], w, h, 0.5)
            s = ref_nets.lightglue_forward(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
            k = len(ref_post.filter_matches(s, 0.1)[0])
            if i >= warm:
                nfrag.append(len(margins.fragile_rows(s, 0.05)))      # rows within 0.05 of a decision boundary: the share of matches a 2-byte matcher may legitimately flip
        if i >= warm:                                   # the first pairs warm the thread pool and the allocator
            times.append(time.perf_counter() - t0)
            nmatch.append(k)
    med = float(np.median(times))
    agree = None
    if gpu_nmatch is not None and len(gpu_nmatch) >= warm + n_pairs:     # same pairs on both sides: the match counts are a (coarse) parity signal
        g = np.asarray(gpu_nmatch[warm:warm + n_pairs], np.float64)
        agree = dict(cpu_matches_mean=float(np.mean(nmatch)), gpu_matches_mean_same_pairs=float(g.mean()),
                     max_abs_count_diff=int(np.abs(g - np.asarray(nmatch)).max()))
    return dict(value=1.0 / med, unit="pairs/s", cores=torch.get_num_threads(), kind="port", same_pairs_as_gpu=agree,
                fragile_share_of_matches=(float(sum(nfrag)) / max(sum(nmatch), 1)) if nfrag else None,     # (tests gate 6 %: tests/test_gpu_stereo.py)
                sample=f"median of {n_pairs} synthetic {w}x{h} stereo pairs after {warm} warm-ups ({sum(times):.1f} s), fp32 PyTorch-CPU "
                       f"oracle + numpy post-processing ({'PLNet points + lines + junctions' if s1 is not None else 'SuperPoint'} + LightGlue), "
                       f"{float(np.mean(nmatch)):.0f} matches per pair")


def latency_b1(args, rank, world, local, dev):
    """--workload b1: what ONE stereo keyframe costs through the batch-1 host API — the regime AirSLAM's feature thread runs in
    (map_builder.cc:83-109: Detect(left, right, features, lines, junctions) + MatchingPoints, one pair at a time, host buffers in and out).
    Prints p50 / p99 / mean of the keyframe as one call (airfe_stereo_keyframe), as the reference's two calls and as three (infer, infer, match);
    `value` = 1000 / p50 of the one-call form (pairs/s of a single stream)."""
    from airslam_amd import api, synth, weights
    H, W, K = args.height, args.width, args.max_keypoints
    root = os.path.dirname(os.path.abspath(__file__))
    sg = args.matcher == "superglue"
    mw = weights.synthetic_superglue(1234) if sg else weights.synthetic_lightglue(1234)
    plnet = args.detector == "plnet"
    ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234) if plnet else weights.synthetic_superpoint(1234),
                      plnet_s1=os.path.join(root, "tests", "golden", "plnet_s1.airfe") if plnet else None,
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
    out = {"metric": "batch-1 stereo keyframe latency, host images in / host matrices out, PCIe and synchronisation included ("
                     + ("PLNet points + lines, junctions on the left" if plnet else "SuperPoint") + " x2 + " + ("SuperGlue" if sg else "LightGlue") + "): "
                     + ("ONE call (airfe_stereo_keyframe = map_builder.cc:85-86); latency_ms also has the same keyframe as the reference's two calls "
                        "(stereo Detect overload + MatchingPoints) and as three (PLNet::infer twice + MatchingPoints), identical results" if fused
                        else "three reference-shaped calls (detect, detect, MatchingPoints)"),
           "value": 1e3 / head, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate",
           "data": "synthetic", "latency_ms": lat, "call_forms_agree": bool(equal),
           "config": {"workload": f"ONE synthetic {W}x{H} stereo pair per step through the batch-1 host API (airslam_amd.api over the C ABI's host-buffer entries), "
                                  f"max_keypoints={K}; seeded synthetic weights (reference ONNX files are absent)",
                      "matches_mean": float(np.mean(nmatch)), "lines_mean_left": float(np.mean(nlines)), "detector": args.detector, "matcher": args.matcher},
           "roofline": None, "cpu_baseline": None, "collective": None}
    if rank == 0:
        print(json.dumps(out))
    ctx.close()


def seq_workload(args, rank, world, local, dev):
    """--workload seq: BASELINE.json configs[3] — per rank S stereo SEQUENCES of --frames frames (synth.stereo_sequence, seeds 10 + rank * S + s), driven exactly as
    MapBuilder::ExtractFeatureThread drives the front end (src/map_builder.cc:83-141 with the shipped use_superpoint: 1): PLNet stereo keyframes, SuperPoint-only
    normal frames matched against the last keyframe, promotions (airslam_amd/seq.py).  S = 1 runs the one-call host entries (latency regime); S > 1 batches the S
    sequences of a time-step through the device-resident entries.  Every K = 8 frames the temporal match lists go to rank 0 in one collective on a side stream.
    A step = one time-step = one frame of each of the S sequences; value = frames/s of the whole job."""
    from airslam_amd import api, seq, synth, weights
    from airslam_amd import dist as adist
    H, W, K = args.height, args.width, args.max_keypoints
    frames = args.frames if args.steps_given is None else args.warmup + args.steps_given
    warm = min(args.warmup, frames - 1)
    scene_len, KG = args.scene_len, 8
    sweep = sorted(set([args.sequences] + ([1, 4, 8, 16, 32] if args.sweep else [])))
    Smax = max(sweep)
    import multiprocessing as mp
    jobs = [(frames, H, W, 10 + rank * Smax + s_, scene_len) for s_ in range(Smax)]
    with mp.get_context("spawn").Pool(min(Smax, os.cpu_count() or 1)) as pool:     # (2.6 s per 200-frame sequence on one core)
        arrs = pool.map(synth.stereo_sequence_arrays, jobs)
    Lh = np.stack([a[0] for a in arrs], 1)                 # [frames][Smax][H][W]
    Rh = np.stack([a[1] for a in arrs], 1)
    Ld, Rd = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)          # resident in HBM before the clock starts (Smax x frames x 0.72 MB)
    s1_path = os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")
    lg = weights.synthetic_lightglue(1234)
    pol = seq.KeyframeConfig(image_width=W, image_height=H, tracking_point_rate=args.tracking_point_rate, min_init_stereo_feature=args.min_init_stereo,
                             min_num_match=args.min_num_match, max_num_match=max(80, args.min_num_match + 10))
    prec, mprec = (1 if args.dtype == "fp16" else 0), (1 if args.matcher_dtype == "fp16" else 0)

    def run(S):
        common = dict(device=local, precision=prec, matcher_precision=mprec, max_keypoints=K, image_width=W, image_height=H, tuning=args.tuning)
        kf = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=s1_path, lightglue=lg, max_batch=max(S, 2), enc_chunk=max(min(2 * S, args.chunk), 2), **common)
        nf = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=lg, max_batch=max(S, 2), enc_chunk=max(min(S, args.chunk), 2), **common)
        gat = seq.MatchGatherer(KG, S, K, dev)
        results, lat, pending = [], [], []
        if S == 1:
            fe = seq.SequenceFrontEnd(kf, nf, pol)

            def step(t):
                r = fe.step(Lh[t, 0], Rh[t, 0])          # host images in, host matrices out: the batch-1 API takes host buffers (PCIe included)
                if r.matches_idx is not None:
                    m = len(r.matches_idx)
                    idx = torch.zeros((1, K, 2), dtype=torch.int32); sc = torch.zeros((1, K)); idx[0, :m] = torch.from_numpy(r.matches_idx); sc[0, :m] = torch.from_numpy(r.matches_score)
                    h = gat.add(idx.to(dev, non_blocking=True), sc.to(dev, non_blocking=True), torch.tensor([m], dtype=torch.int32).to(dev, non_blocking=True))
                    if h is not None:
                        pending.append(h)
                return [r]
        else:
            bs = seq.BatchedSequences(kf, nf, S, pol, device=dev, copy_results=False)      # (results are read inside the step that made them)

            def step(t):
                rs = bs.step(Ld[t, :S], Rd[t, :S])
                tset = [i for i in range(S) if rs[i].matches_idx is not None]
                if tset:                                   # (device tensors of this step's temporal matches, rows in tset order; the others count 0)
                    h = gat.add(bs.tidx[:len(tset)], bs.tsc[:len(tset)], bs.tnm[:len(tset)], stream=bs.stream)
                    if h is not None:
                        pending.append(h)
                return rs

        def barrier():
            torch.cuda.synchronize(dev)
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize(dev)
        for t in range(warm):
            results.append(step(t))
        barrier()
        t0 = time.perf_counter()
        for t in range(warm, frames):
            ta = time.perf_counter()
            results.append(step(t))
            lat.append(time.perf_counter() - ta)
        for h in pending:
            h.result()
        barrier()
        dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
        kf.close(); nf.close()
        flat = [r for rs in results[warm:] for r in rs]
        a = np.asarray(lat) * 1e3
        return dict(S=S, dt=dt, frames_per_s=S * (frames - warm) * world / dt, ms_per_step=dt / (frames - warm) * 1e3,
                    latency_ms={"p50": float(np.percentile(a, 50)), "p99": float(np.percentile(a, 99)), "mean": float(a.mean()), "max": float(a.max())},
                    schedule={"frames": len(flat), "keyframe_candidates": sum(r.candidate for r in flat), "keyframes": sum(r.frame_type != seq.NORMAL for r in flat),
                              "promotions": sum(r.promoted for r in flat), "normal_frames": sum(r.frame_type == seq.NORMAL for r in flat), "dropped_before_init": sum(r.dropped for r in flat),
                              "temporal_matches_mean": float(np.mean([len(r.matches_idx) for r in flat if r.matches_idx is not None] or [0])),
                              "stereo_matches_mean": float(np.mean([len(r.stereo_idx) for r in flat if r.stereo_idx is not None] or [0])),
                              "lines_mean_keyframe_left": float(np.mean([len(r.lines_left) for r in flat if r.lines_left is not None] or [0]))},
                    gathers=gat.gathers, results=results,
                    wall_split_ms_per_step=(None if S == 1 else {"queue_device_work": bs.t_queue / frames * 1e3, "wait_for_device": bs.t_wait / frames * 1e3,
                                                                 "host_side_of_the_loop": bs.t_host / frames * 1e3, "host_syncs": bs.syncs / frames}))

    runs = {S: run(S) for S in sweep}
    head = runs[args.sequences]
    cpu = None
    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        # the oracle's restatement of the same loop on sequence 0's first frames (bounded: ~20 frames of fp32 PyTorch-CPU + numpy), taking its own decisions
        from oracle import ref_seq
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        chain = ref_seq.Chain(weights.synthetic_plnet_s0(1234), weights.synthetic_superpoint(1234), weights.load_pack(s1_path), lg, W, H, K,
                              policy=dict(tracking_point_rate=args.tracking_point_rate, min_init_stereo_feature=args.min_init_stereo,
                             min_num_match=args.min_num_match, max_num_match=max(80, args.min_num_match + 10)))
        nfr = min(args.cpu_pairs + 2, frames)
        ts, types = [], []
        for t in range(nfr):
            ta = time.perf_counter()
            o = chain.step(Lh[t, 0], Rh[t, 0])
            ts.append(time.perf_counter() - ta); types.append(o["frame_type"])
        dev_types = [rs[0].frame_type for rs in head["results"][:nfr]]
        cpu = dict(value=(nfr - 2) / sum(ts[2:]), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                   sample=f"frames 2..{nfr - 1} of sequence 0 ({sum(ts[2:]):.1f} s; frames 0-1 warm the thread pool): oracle/ref_seq.Chain — fp32 PyTorch-CPU networks + numpy "
                          f"post-processing, the same loop taking its own keyframe decisions",
                   same_schedule_as_gpu=(types == dev_types), frame_types_cpu=types, frame_types_gpu=dev_types)
    if rank == 0:
        out = {"metric": "stereo sequence frames/sec (BASELINE configs[3]: per sequence PLNet stereo keyframes + SuperPoint-only normal frames matched against the last keyframe "
                         "+ promotions, map_builder.cc:83-141 with use_superpoint: 1)",
               "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": world, "steps": frames - warm, "warmup": warm, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate",
               "data": "synthetic", "latency_ms_per_time_step": head["latency_ms"],
               "config": {"workload": f"{args.sequences} synthetic {W}x{H} stereo sequence(s) per GPU x {frames} frames (seeds 10 + rank * S + s, a new scene every {scene_len} frames, "
                                      f"2-3 px pan per frame), images resident in HBM" + (" and in host memory (S = 1 goes through the batch-1 host entries: PCIe included)" if args.sequences == 1 else "")
                                      + f"; keyframe policy = AddKeyframeCheck of vo_euroc.yaml with tracking_point_rate {args.tracking_point_rate} and min_init_stereo_feature {args.min_init_stereo} "
                                      f"(the synthetic matcher weights match ~35 % of the keypoints; the yaml's 0.65 / 90 would make every second frame a keyframe candidate and leave "
                                      f"some sequences uninitialised for a scene); max_keypoints={K}; seeded synthetic weights except PLNet stage 1 (real)",
                          "sequences_per_gpu": args.sequences, "frames": frames, "gather_every_frames": KG, "gathers": head["gathers"], "schedule": head["schedule"],
                          "driver": "airslam_amd.seq.SequenceFrontEnd (one-call host entries)" if args.sequences == 1 else "airslam_amd.seq.BatchedSequences (*_batch_dev entries)"},
               "sweep": {str(S): {"frames_per_s": r["frames_per_s"], "ms_per_time_step": r["ms_per_step"], "latency_ms": r["latency_ms"], "schedule": r["schedule"],
                                   "wall_split_ms_per_step": r["wall_split_ms_per_step"]} for S, r in runs.items()},
               "roofline": None, "cpu_baseline": cpu, "collective": args.collective}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def side_workloads(args, rank, world, local, dev):
    """The other configurations of BASELINE.json behind the same contract (one JSON line, K timed steps between barriers):
    SuperGlue as the matcher (configs[4]; SuperPoint detector), PLNet through the batch-1 host API (--plnet-host), the matcher-only loop-closure
    replay."""
    import tempfile
    from airslam_amd import api, mapfile, synth, weights
    from airslam_amd import dist as adist
    B, H, W, K = args.pairs, args.height, args.width, args.max_keypoints
    prec = 1 if args.dtype == "fp16" else 0
    mprec = 1 if args.matcher_dtype == "fp16" else 0
    sg = args.matcher == "superglue"
    cfg = dict(device=local, precision=prec, matcher_precision=mprec, max_batch=B, enc_chunk=min(args.chunk, B), max_keypoints=K,
               image_width=W, image_height=H, matcher=1 if sg else 0, tuning=args.tuning)
    mw = weights.synthetic_superglue(1234) if sg else weights.synthetic_lightglue(1234)
    mkw = dict(superglue=mw) if sg else dict(lightglue=mw)
    n_match = [0.0]

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    root = os.path.dirname(os.path.abspath(__file__))
    if args.plnet_host:
        pairs = [synth.stereo_pair(H, W, 1000 + rank * 64 + i) for i in range(min(B, 8))]
        ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=os.path.join(root, "tests", "golden", "plnet_s1.airfe"),
                          **dict(cfg, max_batch=2, enc_chunk=2), **mkw)
        det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, W, H, 1 if sg else 0)
        lines_n = [0]

        def step():
            for i in range(B):
                left, right = pairs[i % len(pairs)]
                acc = []
                ok, fl, jl = det.DetectLines(left, None, acc, junction_detection=True)      # left: points + lines + junctions
                ok2, fr, _ = det.DetectLines(right, None, [], junction_detection=False)     # right: no junctions (feature_detector.cc:100-101)
                n, _ = pm.MatchingPoints(fl, fr)
                n_match[0] = n; lines_n[0] = len(acc)
        what = (f"{B} stereo pairs per step through the batch-1 HOST API (PCIe and one sync per call included): 2x PLNet::infer "
                f"(points + on-device line branch + real stage-1 weights + junctions on the left) + 1x {'SuperGlue' if sg else 'LightGlue'}")
    else:
        ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), **cfg, **mkw)
        ls, rs = synth.stereo_batch(B, H, W, 1000 + rank)
        L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
        fl = torch.zeros((B, K, 259), device=dev); fr = torch.zeros((B, K, 259), device=dev)
        nl = torch.zeros((B,), dtype=torch.int32, device=dev); nr = torch.zeros((B,), dtype=torch.int32, device=dev)
        idx = torch.zeros((B, K, 2), dtype=torch.int32, device=dev); sc = torch.zeros((B, K), device=dev)
        nm = torch.zeros((B,), dtype=torch.int32, device=dev)
        i0 = torch.zeros((B, K), dtype=torch.int32, device=dev); i1 = torch.zeros((B, K), dtype=torch.int32, device=dev)
        m0 = torch.zeros((B, K), device=dev); m1 = torch.zeros((B, K), device=dev)
        stream = torch.cuda.Stream(device=dev)
        sh = stream.cuda_stream
        if args.workload == "loop":
            # build a small "map": detect F frames once, write them as AirSLAM feature records, read them back, replay the matcher
            F = 40
            frames = []
            for f in range(F):
                img = np.roll(synth.stereo_pair(H, W, 2000 + rank)[0], (5 * f, 11 * f), axis=(0, 1))
                frames.append(ctx.detect_points(img))
            with tempfile.TemporaryDirectory() as td:
                mapfile.write_records(os.path.join(td, "map.airfemap"), frames)
                frames = mapfile.read_records(os.path.join(td, "map.airfemap"))
            pairs_l = mapfile.loop_closure_pairs(F)[:B]
            B2 = len(pairs_l)
            qa = torch.zeros((B2, K, 259)); qb = torch.zeros((B2, K, 259))
            na = torch.zeros((B2,), dtype=torch.int32); nb = torch.zeros((B2,), dtype=torch.int32)
            for i, (q, c2) in enumerate(pairs_l):
                qa[i, :frames[q].shape[0]] = torch.from_numpy(frames[q]); na[i] = frames[q].shape[0]
                qb[i, :frames[c2].shape[0]] = torch.from_numpy(frames[c2]); nb[i] = frames[c2].shape[0]
            qa, qb, na, nb = qa.to(dev), qb.to(dev), na.to(dev), nb.to(dev)

            def step():
                if sg:
                    ctx.match_superglue_batch_dev(qa, na, qb, nb, i0[:B2], i1[:B2], m0[:B2], m1[:B2], stream=sh)
                else:
                    ctx.match_lightglue_batch_dev(qa, na, qb, nb, idx[:B2], sc[:B2], nm[:B2], stream=sh)
            B = B2
            what = (f"matcher only: {B} (query, candidate) frame pairs per step replayed from AirSLAM feature records (loop closure, "
                    f"map_refiner.cc:213-230: each query against its 5 best candidates), {'SuperGlue' if sg else 'LightGlue'}, {K} keypoints max")
        else:
            def step():
                ctx.detect_batch_dev(L, fl, nl, stream=sh)
                ctx.detect_batch_dev(R, fr, nr, stream=sh)
                if sg:
                    ctx.match_superglue_batch_dev(fl, nl, fr, nr, i0, i1, m0, m1, stream=sh)
                else:
                    ctx.match_lightglue_batch_dev(fl, nl, fr, nr, idx, sc, nm, stream=sh)
            what = (f"{B} synthetic {W}x{H} stereo pairs per step per GPU, resident in HBM: 2x SuperPoint detect + "
                    f"{'SuperGlue (18 layers, 100 Sinkhorn iterations)' if sg else 'LightGlue'}, max_keypoints={K}")

    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
    stages = {}
    if not args.no_profile:
        ctx.profile(True)
        for _ in range(max(1, min(args.stage_steps, 2))):
            step()
        torch.cuda.synchronize(dev)
        stages = ctx.profile_read()
        ctx.profile(False)
    barrier()
    if rank == 0:
        if not args.plnet_host:
            if sg:
                n_match[0] = float((i0 >= 0).sum(1).float().mean())
            else:
                n_match[0] = float(nm.float().mean())
        ns = max(1, min(args.stage_steps, 2))
        out = {"metric": "stereo detect+match pairs/sec" if args.workload == "stereo" else "matched frame pairs/sec (matcher only)",
               "value": B * args.steps * world / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate",
               "data": "synthetic",
               "config": {"workload": what + "; seeded synthetic weights (reference ONNX files are absent)", "pairs_per_step_per_gpu": B,
                          "detector": "plnet (batch-1 host API)" if args.plnet_host else "superpoint", "matcher": args.matcher, "matches_mean": n_match[0]},
               "roofline": None, "cpu_baseline": None, "collective": args.collective}
        if args.plnet_host:
            out["config"]["lines_last_frame"] = lines_n[0]
        if stages:
            tot = sum(v["ms"] for v in stages.values())
            out["stages"] = {k: {"ms_per_step": v["ms"] / ns, "share": v["ms"] / tot if tot else 0,
                                 "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None}
                             for k, v in stages.items() if v["launches"]}
            fl_tot = sum(v["flops"] for v in stages.values()) / ns
            ms_tot = sum(v["ms"] for v in stages.values()) / ns
            if fl_tot > 0 and ms_tot > 0:
                ach = fl_tot / (ms_tot * 1e-3) / 1e12
                util, usrc = None, None
            pf = latest_profile("pmc_summary.json")
            if pf:      # counter-derived MFMA utilisation of the encoder kernels (separate --pmc pass, tools/pmc_summary.py)
                with open(pf) as fh:
                    ks = json.load(fh)["kernels"]
                enc = {k: v for k, v in ks.items() if ("conv64r_kernel" in k or "conv128r_kernel" in k) and "mfma_util" in v}
                wsum = sum(v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"] for v in enc.values())
                if wsum > 0:
                    util = {"encoder_time_weighted": sum(v["mfma_util"] * v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"]
                                                         for v in enc.values()) / wsum,
                            "per_kernel": {k.split("(")[0].replace("void airfe::", ""): round(v["mfma_util"], 3) for k, v in enc.items()}}
                    usrc = "profiles/" + os.path.basename(pf) + ": SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)"
            out["roofline"] = {"bound": "mfma", "mfma_util_counters": util, "mfma_util_source": usrc, "kernel": "all bracketed matrix stages of one step (algorithmic FLOPs / their event time)",
                                   "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_TFLOPS, "traffic": None}
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: WORLD_SIZE under a launcher, else 1")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; --workload seq: --frames minus the warm-up)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per step per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--max-keypoints", type=int, default=400)
    ap.add_argument("--chunk", type=int, default=128, help="images per launch of the full-resolution conv layers (a stereo step's 2 x pairs images may "
                                                            "be one chunk: 128 measured 1.7 %% faster on the conv64 stage than 64, profiles/r03_probe7_*)")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="detector (encoder) storage type; fp16 = the reference's kFP16 engines and the type that meets the parity gates")
    ap.add_argument("--matcher-dtype", default="fp16", choices=["bf16", "fp16"],
                    help="matcher storage type (fp16 = the reference's kFP16 engines, light_glue.cpp:115)")
    ap.add_argument("--cpu-pairs", type=int, default=20, help="CPU-baseline sample size (0 = skip)")
    ap.add_argument("--matcher", default="lightglue", choices=["lightglue", "superglue"],
                    help="superglue: 18-layer GNN + 100 Sinkhorn iterations (BASELINE configs[4]: --width 1280 --height 720 --max-keypoints 1024)")
    ap.add_argument("--detector", default=None, choices=["superpoint", "plnet"],
                    help="plnet (default): PLNet::infer on both images (points + line branch + stage 1 + line filter, junctions on the left), the "
                         "reference's keyframe step; superpoint: the point-only step (the default of --workload track: what a normal frame runs with the "
                         "shipped use_superpoint: 1, feature_detector.cc:36-41)")
    ap.add_argument("--sequences", type=int, default=8, help="--workload seq: sequences per GPU (1 = the one-call host entries, > 1 = lock-step batches)")
    ap.add_argument("--frames", type=int, default=200, help="--workload seq: frames per sequence")
    ap.add_argument("--scene-len", type=int, default=40, help="--workload seq: frames per synthetic scene (a scene change forces a promotion)")
    ap.add_argument("--sweep", action="store_true", help="--workload seq: also run S = 1, 4, 8, 16, 32")
    ap.add_argument("--tracking-point-rate", type=float, default=0.2, help="--workload seq: AddKeyframeCheck's tracking_point_rate (yaml: 0.65; see the workload text)")
    ap.add_argument("--min-num-match", type=int, default=30, help="--workload seq: AddKeyframeCheck's min_num_match (yaml: 30; the tests raise it so that promotions happen)")
    ap.add_argument("--min-init-stereo", type=int, default=60, help="--workload seq: min_init_stereo_feature (yaml: 90 of ~400 trained-matcher stereo matches; the synthetic "
                                                                     "matcher finds ~130 per pair, of which some sequences keep 80-90 inside the camera's band: they would stay uninitialised for a whole scene)")
    ap.add_argument("--plnet-host", action="store_true", help="PLNet + matcher through the batch-1 HOST API instead (PCIe and one sync per call included)")
    ap.add_argument("--workload", default="stereo", choices=["stereo", "track", "loop", "frontend", "b1", "seq"],
                    help="seq: BASELINE configs[3] — whole stereo SEQUENCES driven as map_builder.cc:83-141 drives the front end (airslam_amd/seq.py); "
                         "track: the NORMAL-frame step of the VO loop (map_builder.cc:94-101: Detect(left, features) — SuperPoint with the shipped use_superpoint: 1, "
                         "--detector plnet for the use_superpoint: 0 form — then MatchingPoints(last_keyframe, frame)); loop: matcher only, replaying a map file's feature records (loop closure, "
                         "map_refiner.cc:213-230); b1: LATENCY of one stereo keyframe through the batch-1 host API, as the SLAM loop calls it (map_builder.cc:83-109): "
                         "p50 / p99 over --steps pairs, PLNet + LightGlue, host images in, host matrices out; frontend: the WHOLE per-keyframe front end, device-resident — rectify both raw images (camera.cc:161-182), "
                         "the stereo step, AssignPointsToLines on both frames + MatchLines with the stereo band (frame.cc:125,147-184), BoW words of the "
                         "left features (bow/database.cc:57-89)")
    ap.add_argument("--line-precision", type=int, default=0, choices=[0, 2, 3], help="PLNet stage 1: 3 = fp16 (hi, lo) operand pairs on the 2-byte MFMA, 2 = f32-input MFMA "
                                                                                     "(the same lines); 0 = the library's default (3)")
    ap.add_argument("--tuning", default="", help="airfe_tuning overrides for A/B runs, e.g. assign_fused=0,overlap_lines=0 (include/airfe.h)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--stage-steps", type=int, default=3, help="extra untimed steps for the per-stage table")
    args = ap.parse_args()
    args.tuning = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tuning.split(",") if kv} or None
    args.steps_given = args.steps
    if args.steps is None:
        args.steps = 200
    if args.detector is None:
        args.detector = "superpoint" if args.workload == "track" else "plnet"
    gpus_given = args.gpus is not None
    if not gpus_given:               # `torchrun --nproc-per-node=8 bench.py` without --gpus: the launcher's world size is the answer
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU under torch.distributed.run, exactly what the
    # contract's own launch line does) instead of silently measuring one GPU.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from airslam_amd import api, dist as adist, synth, weights

    rank, world, local = adist.init_from_env(os.environ.get("AIRFE_DIST_BACKEND"))   # default: nccl (= RCCL) on GPUs
    if os.environ.get("AIRFE_ONE_DEVICE"):     # test hook: several ranks share GPU 0 (with AIRFE_DIST_BACKEND=gloo; RCCL refuses that)
        local = 0
    if gpus_given and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    args.collective = ({"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                        "per_step": "one packed gather of the match lists to rank 0 (airslam_amd/dist.py)"} if world > 1 else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, H, W, K = args.pairs, args.height, args.width, args.max_keypoints

    if args.workload == "track" and (args.plnet_host or args.matcher == "superglue"):
        raise SystemExit("--workload track runs the device-resident PLNet / SuperPoint + LightGlue path")
    if args.workload == "b1":
        return latency_b1(args, rank, world, local, dev)
    if args.workload == "seq":
        return seq_workload(args, rank, world, local, dev)
    if args.plnet_host or args.matcher == "superglue" or args.workload == "loop":
        return side_workloads(args, rank, world, local, dev)

    plnet = args.detector == "plnet"
    s1_path = os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")      # the REAL stage-1 weights (output/plnet_s1.onnx of the reference)
    sp = weights.synthetic_plnet_s0(1234) if plnet else weights.synthetic_superpoint(1234)
    lg = weights.synthetic_lightglue(1234)
    ctx = api.Context(superpoint=sp, lightglue=lg, plnet_s1=s1_path if plnet else None, device=local, precision=1 if args.dtype == "fp16" else 0,
                      matcher_precision=1 if args.matcher_dtype == "fp16" else 0, max_batch=B,
                      enc_chunk=args.chunk, max_keypoints=K, image_width=W, image_height=H, tuning=args.tuning, line_precision=args.line_precision)

    ls, rs = synth.stereo_batch(B, H, W, 1000 + rank)
    L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
    fl = torch.zeros((B, K, 259), device=dev); fr = torch.zeros((B, K, 259), device=dev)
    nl = torch.zeros((B,), dtype=torch.int32, device=dev); nr = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
    sc = torch.zeros((B, K), device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)

    CL, CJ = 1024, 1024                  # line / junction capacity per image (the true counts come back in `found`: checked below)
    if plnet:
        lines = torch.zeros((2 * B, CL, 4), dtype=torch.float64, device=dev); nlines = torch.zeros((2 * B,), dtype=torch.int32, device=dev)
        junc = torch.zeros((B, CJ, 259), device=dev); njunc = torch.zeros((B,), dtype=torch.int32, device=dev)
        found = torch.zeros((3 * B,), dtype=torch.int32, device=dev)

    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    def points_step():
        ctx.stereo_batch_dev(L, R, fl, fr, nl, nr, idx, sc, nm, stream=sh)

    frontend = args.workload == "frontend"
    if frontend:
        if not plnet:
            raise SystemExit("--workload frontend runs the PLNet detector (lines are what the extra stages work on)")
        # rectification maps of a mildly distorted stereo rig (the construction stays reference code: camera.cc:60-75); raw = the synthetic images
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        r2 = ((xx - W / 2) ** 2 + (yy - H / 2) ** 2) / float(W * W)
        for side, sgn in ((0, 1.0), (1, -1.0)):
            ctx.set_rectify_maps(side, (xx + sgn * 0.7 + (xx - W / 2) * 0.02 * r2).astype(np.float32), (yy + 0.3 * sgn + (yy - H / 2) * 0.02 * r2).astype(np.float32))
        ctx.bow_load(weights.synthetic_vocabulary(1234))
        rawL, rawR = L, R
        L, R = torch.empty_like(rawL), torch.empty_like(rawR)
        CE = 16 * CL
        rel = [dict(rp=torch.zeros((B, CL + 1), dtype=torch.int32, device=dev), pi=torch.zeros((B, CE), dtype=torch.int32, device=dev),
                    pd=torch.zeros((B, CE), dtype=torch.float64, device=dev), tot=torch.zeros((B,), dtype=torch.int32, device=dev)) for _ in range(2)]
        line_matches = torch.zeros((B, CL), dtype=torch.int32, device=dev)
        words = torch.zeros((B, K), dtype=torch.int32, device=dev); wweights = torch.zeros((B, K), device=dev)
        band = (1.0, 200.0, 5.0)           # Camera::MinXDiff / MaxXDiff / MaxYDiff of a rig like EuRoC's (frame.cc:143-145)

    track = args.workload == "track"
    if track:
        # the last keyframe's features (map_builder.cc:100 `_last_keyframe_feature->GetAllFeatures()`): the left images, detected once;
        # the "new frames" are the right images (the same scenes seen from a shifted camera)
        if plnet:
            ctx.detect_plnet_batch_dev(L, fl, nl, lines[:B], nlines[:B], None, None, found[:B], stream=sh)
        else:
            ctx.detect_batch_dev(L, fl, nl, stream=sh)
        torch.cuda.synchronize(dev)

    def step():
        if track:
            # Detect(image_left, features) on a normal frame = PLNet::infer(points + lines, no junctions: feature_detector.cc:36-60) on ONE
            # image, then MatchingPoints(features_last_keyframe, left_features) (map_builder.cc:94-101)
            if plnet:
                ctx.detect_plnet_batch_dev(R, fr, nr, lines[B:], nlines[B:], None, None, found[B:2 * B], stream=sh)
            else:
                ctx.detect_batch_dev(R, fr, nr, stream=sh)
            ctx.match_lightglue_batch_dev(fl, nl, fr, nr, idx, sc, nm, stream=sh)
        elif frontend:
            ctx.rectify_batch_dev(0, rawL, L, stream=sh)
            ctx.rectify_batch_dev(1, rawR, R, stream=sh)
            ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm, found, stream=sh)
            ctx.assign_points_to_lines_batch_dev(lines[:B], nlines[:B], fl, nl, rel[0]["rp"], rel[0]["pi"], rel[0]["pd"], rel[0]["tot"], stream=sh)
            ctx.assign_points_to_lines_batch_dev(lines[B:], nlines[B:], fr, nr, rel[1]["rp"], rel[1]["pi"], rel[1]["pd"], rel[1]["tot"], stream=sh)
            ctx.match_lines_batch_dev(rel[0]["rp"], rel[0]["pi"], nlines[:B], nl, rel[1]["rp"], rel[1]["pi"], nlines[B:], nr, idx, nm, line_matches,
                                      stereo_filter=band, feat0_t=fl, feat1_t=fr, stream=sh)
            ctx.bow_transform_dev(fl, words, wweights, stream=sh)
        elif plnet:
            ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm, found, stream=sh)
        else:
            points_step()
        if world > 1:          # the step's match lists to rank 0: copied on the compute stream, gathered on a SIDE stream behind an event —
            gatherer.add(idx, sc, nm, stream=stream)      # the next step's kernels do not wait for the collective; barrier() drains it (torch.cuda.synchronize)

    gatherer = None
    if world > 1:
        from airslam_amd import seq as aseq
        gatherer = aseq.MatchGatherer(1, B, K, dev)
        args.collective["per_step"] = "one packed gather of the match lists to rank 0 on a side stream behind an event (airslam_amd.seq.MatchGatherer, K = 1)"

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: only the dominant kernel's stage carries HIP events (on the launch stream); bracketing EVERY stage
    # costs ~8 % of a step, so the full per-stage table comes from a second, untimed pass below.
    if not args.no_profile:
        ctx.profile(stages=[DOMINANT_STAGE])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dom = ctx.profile_read()[DOMINANT_STAGE] if not args.no_profile else None
    ctx.profile(False)
    dt = adist.max_over_ranks(dt, dev)
    stages = {}
    if not args.no_profile:          # every rank takes part: a step contains the match gather when world > 1
        ctx.profile(True)
        for _ in range(args.stage_steps):
            step()
        torch.cuda.synchronize(dev)
        stages = ctx.profile_read()
        ctx.profile(False)
    barrier()
    points_only = None
    if plnet and not track and not frontend:            # the point-only step (Detect(left, right, features) with use_superpoint = 1) on the same context and inputs
        for _ in range(args.warmup):
            points_step()
        barrier()
        tp = time.perf_counter()
        for _ in range(args.steps):
            points_step()
        barrier()
        points_only = B * args.steps * world / adist.max_over_ranks(time.perf_counter() - tp, dev)
        step()                         # (the counts reported below are the PLNet step's)
        barrier()

    if rank == 0:
        total_pairs = B * args.steps * world
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": ("tracked frames/sec (normal-frame step: 1x " + ("PLNet @512x512 internal: points + lines" if plnet else "SuperPoint-VGG detect")
                       + " on the new frame + LightGlue against the last keyframe)") if track else
                      "stereo detect+match pairs/sec (" + ("2x PLNet @512x512 internal: points + lines, junctions on the left" if plnet
                                                          else "2x SuperPoint-VGG detect @512x512 internal") + " + LightGlue match)",
            "value": total_pairs / dt, "unit": "frames/s" if track else "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate",
            "data": "synthetic",
            "parity_dtype": ("fp16 storage / fp32 accumulate: the reference's own engine type (kFP16, src/super_point.cpp:97, src/light_glue.cpp:115) and the only "
                             "2-byte type inside the north-star tolerances (descriptors 4e-4 cosine, LightGlue 0.03 of 0.05); bf16 FAILS them (2e-2 cosine, "
                             "0.25 log-assignment: DESIGN.md §1) and is selectable with --dtype bf16 --matcher-dtype bf16 only as a non-compliant speed run"),
            "config": {"workload": f"{B} synthetic {W}x{H} uint8 stereo pairs per step per GPU, resident in HBM; "
                                   f"max_keypoints={K}, nms_radius=4, LightGlue 9 layers; seeded synthetic weights "
                                   f"(reference ONNX files are absent)",
                       "pairs_per_step_per_gpu": B, "internal_resolution": 512, "parallelism": f"frame-sharded x{world}",
                       "keypoints_left_right_mean": [float(nl.float().mean()), float(nr.float().mean())],
                       "matches_mean": float(nm.float().mean()), "detector": args.detector},
            "collective": args.collective,
        }
        if args.tuning:
            out["config"]["tuning"] = args.tuning
        if plnet:
            out["config"]["line_precision"] = args.line_precision or "library default (include/airfe.h)"
        if track:
            out["config"]["workload"] = (f"{B} synthetic {W}x{H} uint8 frames per step per GPU, resident in HBM, each matched against its last keyframe's "
                                         f"features (map_builder.cc:94-101); max_keypoints={K}; seeded synthetic weights (reference ONNX files are absent)")
        if plnet:
            fh_ = found.cpu().numpy()
            if (fh_[:2 * B] > CL).any() or (fh_[2 * B:] > CJ).any():
                raise SystemExit("bench: line / junction capacity overflow")
            out["config"]["workload"] += ("; PLNet line branch: published HAWPv3 head with seeded synthetic weights, stage 1 with the REAL weights of "
                                          "output/plnet_s1.onnx, line_threshold / line_length_threshold at the reference's 0.75 / 50")
            out["config"]["lines_mean"] = float(nlines.float().mean())
            out["config"]["junctions_mean_left"] = float(njunc.float().mean())
            if track:
                out["config"]["lines_mean"] = float(nlines[B:].float().mean())
                del out["config"]["junctions_mean_left"]
        if frontend:
            lm = line_matches.cpu().numpy(); nlh = nlines.cpu().numpy()
            if (rel[0]["tot"].cpu().numpy() > CE).any() or (rel[1]["tot"].cpu().numpy() > CE).any():
                raise SystemExit("bench: point-line relation capacity overflow")
            out["metric"] = ("keyframe front ends/sec, device-resident end to end: rectify x2 + 2x PLNet (points, lines, junctions on the left) + LightGlue + "
                             "AssignPointsToLines x2 + MatchLines (stereo band) + BoW words of the left features")
            out["unit"] = "stereo keyframes/s"
            out["config"]["workload"] += "; + rectification of both raw images, point-line association, stereo line matching, BoW quantisation (synthetic vocabulary, 10^4 words)"
            out["config"]["points_on_lines_mean_left"] = float(rel[0]["tot"].float().mean())
            out["config"]["stereo_line_matches_mean"] = float(np.mean([(lm[b, :nlh[b]] >= 0).sum() for b in range(B)]))
        if plnet and not track and not frontend:
            out["config"]["points_only_pairs_per_s"] = points_only
            out["config"]["points_only_note"] = ("the point-only step (2x detect + LightGlue, airfe_stereo_batch_dev: `--detector superpoint`) timed on the "
                                                 "same context and inputs over the same number of steps")
        if dom:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
            traffic, tsrc = None, None
            tdoc, tf, tage = counter_profile("hbm_traffic.json")
            if tdoc:      # measured in separate --pmc passes (never together with other tracing), see profiles/README.md
                rec = tdoc.get("conv1_fused", {})
                if rec.get("hbm_bytes_per_launch") and rec.get("images_per_launch"):      # per image x the images one launch covers in THIS run
                    traffic = rec["hbm_bytes_per_launch"] / rec["images_per_launch"] * (2.0 * B * args.steps / max(dom["launches"], 1))
                tsrc = "profiles/" + os.path.basename(tf) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch of that kernel, scaled to this run's images per launch)"
            util, usrc = None, None
            pdoc, pf, page = counter_profile("pmc_summary.json")
            if pdoc:      # counter-derived MFMA utilisation of the encoder kernels (separate --pmc pass, tools/pmc_summary.py)
                ks = pdoc["kernels"]
                enc = {k: v for k, v in ks.items() if ("conv64r_kernel" in k or "conv128r_kernel" in k) and "mfma_util" in v}
                wsum = sum(v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"] for v in enc.values())
                if wsum > 0:
                    util = {"encoder_time_weighted": sum(v["mfma_util"] * v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"]
                                                         for v in enc.values()) / wsum,
                            "per_kernel": {k.split("(")[0].replace("void airfe::", ""): round(v["mfma_util"], 3) for k, v in enc.items()}}
                    usrc = "profiles/" + os.path.basename(pf) + ": SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)"
            out["roofline"] = {"bound": "mfma", "mfma_util_counters": util, "mfma_util_source": usrc, "kernel": "conv64r_kernel<POOL, FUSE1A> (conv1a + conv1b + 2x2 max-pool in one launch over an encoder chunk)",
                               "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_TFLOPS,
                               "traffic": traffic, "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1), "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                               "launches": dom["launches"],
                               # `achieved` / `frac` are measured live in THIS run (HIP events on the launch stream); `traffic` and `mfma_util_counters` come from
                               # committed rocprofv3 PMC passes and are reported ONLY when those passes were taken on the kernel sources this process runs:
                               "counters_age": {"traffic": tage, "mfma_util": page,
                                                "rule": "profile_csrc_sha == tree_csrc_sha (sha256 over csrc/*.hip, csrc/*.h, include/airfe*.h), else null"}}
        if stages:
            tot = sum(s["ms"] for s in stages.values())
            out["stages_note"] = f"separate untimed pass of {args.stage_steps} steps with every stage bracketed by events"
            out["stages"] = {k: {"ms_per_step": v["ms"] / args.stage_steps, "share": v["ms"] / tot if tot else 0,
                                 "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None,
                                 "algo_gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                             for k, v in stages.items() if v["launches"]}
            for k, v in out["stages"].items():      # each stage against its own roof (algorithmic FLOPs or bytes of the stage / its event time / the peak)
                if STAGE_BOUND.get(k) == "mfma" and v["tflops"]:
                    v["bound"], v["frac"] = "mfma", v["tflops"] / PEAK_MFMA_TFLOPS
                elif k == "plnet_stage1":
                    v["bound"], v["frac"] = "latency (gather chains + a 4-layer MLP per 32-line tile: DESIGN.md 3)", None
                elif v["algo_gbs"]:
                    v["bound"], v["frac"] = "hbm", v["algo_gbs"] / PEAK_HBM_GBS
            if out.get("roofline"):
                # the WHOLE step against the matrix peak: algorithmic FLOPs of every matrix stage of one step / the timed ms_per_step / peak
                fl_step = sum(v["flops"] for v in stages.values()) / args.stage_steps
                out["roofline"]["step_frac"] = fl_step / (ms_step * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS
                out["roofline"]["step_gflop"] = fl_step / 1e9
        if world == 1 and args.cpu_pairs > 0 and not track and not frontend:
            out["cpu_baseline"] = cpu_baseline(sp, lg, H, W, args.cpu_pairs, K, s1=weights.load_pack(s1_path) if plnet else None,
                                               gpu_nmatch=nm.cpu().numpy())
            cb = out["cpu_baseline"]["same_pairs_as_gpu"]
            if cb and args.dtype == "fp16" and args.matcher_dtype == "fp16":       # a bench whose outputs drifted from the oracle's says so ON the line (ADVICE r04:
                out["cpu_baseline"]["parity_ok"] = bool(cb["max_abs_count_diff"] <= max(12, 0.15 * cb["cpu_matches_mean"]))      # an assert here lost the line)
                if not out["cpu_baseline"]["parity_ok"]:
                    print(f"bench.py: GPU and CPU-oracle match counts disagree on the same pairs: {cb}", file=sys.stderr)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
