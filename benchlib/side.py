"""The other configurations of BASELINE.json behind the same contract (one JSON line, K timed steps between barriers): SuperGlue as the matcher (configs[4];
SuperPoint detector), PLNet through the batch-1 host API (--plnet-host), the matcher-only loop-closure replay (--workload loop)."""
import json
import os
import tempfile
import time

import numpy as np
import torch

from . import common as cm


def run(args, rank, world, local, dev):
    """The other configurations of BASELINE.json behind the same contract (one JSON line, K timed steps between barriers):
    SuperGlue as the matcher (configs[4]; SuperPoint detector), PLNet through the batch-1 host API (--plnet-host), the matcher-only loop-closure
    replay."""
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

    barrier = lambda: cm.barrier(dev, world)
    if args.plnet_host:
        pairs = [synth.stereo_pair(H, W, 1000 + rank * 64 + i) for i in range(min(B, 8))]
        ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=cm.S1_PACK,
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
        out = cm.line(
            args, metric="stereo detect+match pairs/sec" if args.workload == "stereo" else "matched frame pairs/sec (matcher only)",
            value=B * args.steps * world / dt, unit="pairs/s", world=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
            config={"workload": what + "; seeded synthetic weights (reference ONNX files are absent)", "pairs_per_step_per_gpu": B,
                    "detector": "plnet (batch-1 host API)" if args.plnet_host else "superpoint", "matcher": args.matcher, "matches_mean": n_match[0]},
            collective=args.collective)
        if args.plnet_host:
            out["config"]["lines_last_frame"] = lines_n[0]
        if stages:
            out["stages"], fl_tot = cm.stage_table(stages, ns)
            ms_tot = sum(v["ms"] for v in stages.values()) / ns
            if fl_tot > 0 and ms_tot > 0:
                ach = fl_tot / (ms_tot * 1e-3) / 1e12
                pdoc, pf, page = cm.counter_profile("pmc_summary.json")
                util, usrc = cm.encoder_mfma_util(pdoc, pf)
                ms_step = dt / args.steps * 1e3
                out["roofline"] = {"bound": "mfma", "mfma_util_counters": util, "mfma_util_source": usrc, "kernel": "all bracketed matrix stages of one step (algorithmic FLOPs / their event time)",
                                   "achieved": ach, "peak": cm.PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / cm.PEAK_MFMA_TFLOPS, "traffic": None,
                                   "step_frac": fl_tot / (ms_step * 1e-3) / 1e12 / cm.PEAK_MFMA_TFLOPS, "step_gflop": fl_tot / 1e9, "counters_age": {"mfma_util": page}}
        if world == 1 and args.cpu_pairs > 0 and args.workload == "stereo" and not args.plnet_host:
            from . import cpu
            gm = (i0 >= 0).sum(1).cpu().numpy() if sg else nm.cpu().numpy()
            out["cpu_baseline"] = cpu.stereo(weights.synthetic_superpoint(1234), None if sg else mw, H, W, min(args.cpu_pairs, 6), K, warm=1, sg=mw if sg else None,
                                             gpu_nmatch=gm, seed=1000 + rank)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()
