"""The headline workload and its two neighbours, all device-resident batches through the C ABI's *_batch_dev entries:
  stereo    (default) the reference's keyframe step, map_builder.cc:85-86: Detect(left, right, features, lines, junctions) + MatchingPoints(left, right)
  track     the normal-frame step, map_builder.cc:94-101
  frontend  rectify x2 + the keyframe step + AssignPointsToLines x2 + MatchLines + BoW
and, for the stereo workload, the same step HOST TO HOST (`--io host`): pinned host images in, features / lines / junctions / matches back in pinned host memory,
copies on their own streams beside the compute stream (the reference does one H2D and one D2H inside every infer(): src/plnet.cpp:231,237)."""
import json
import os
import sys
import time

import numpy as np
import torch

from . import common as cm


def host_to_host(ctx, dev, ls, rs, B, H, W, K, CL, CJ, steps, warm, world):
    """Steady state of: H2D of the step's 2 B images (copy-in stream, copy engine) -> airfe_stereo_plnet_batch_dev + airfe_pack_rows_dev (compute stream: the counts
    and the VALID feature / match / line / junction rows of the step back to back in one device block, offsets from a scan of the counts on the device) -> D2H of the
    offsets, then — the host knows the size one step later — of the packed bytes (copy-out stream, copy engine) — with two buffer sets in flight: while step i
    computes, the host finishes step i - 1's copies and takes step i - 2's results.  The junction buffer alone is 68 MB per step at its capacity of 1024 rows per
    image for ~150 junctions: at capacity the step moves 125 MB back, packed 78 MB.  -> dict(pairs_per_s, ms_per_step, bytes, GB/s)."""
    from airslam_amd import dist as adist
    Lp, Rp = torch.from_numpy(ls).pin_memory(), torch.from_numpy(rs).pin_memory()
    s_in, s_cmp, s_out, s_cnt = (torch.cuda.Stream(device=dev) for _ in range(4))      # (the offsets of step i and the packed bytes of step i - 1 on streams of their own:
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)                # on one stream the second would queue behind the first's wait for step i)
    i32 = torch.int32
    sets = []
    for _ in range(2):
        d = dict(fl=z(B, K, 259), fr=z(B, K, 259), nl=z(B, dt=i32), nr=z(B, dt=i32), idx=z(B, K, 2, dt=i32), sc=z(B, K), nm=z(B, dt=i32),
                 lines=z(2 * B, CL, 4, dt=torch.float64), nlines=z(2 * B, dt=i32), junc=z(B, CJ, 259), njunc=z(B, dt=i32), found=z(3 * B, dt=i32))
        jobs = [(d[k], None, None, 4, d[k].numel()) for k in ("nl", "nr", "nm", "nlines", "njunc", "found")]      # jobs 0-5: the count arrays themselves
        for b in range(B):
            jobs += [(d["fl"][b], None, d["nl"][b:b + 1], 1036, K), (d["fr"][b], None, d["nr"][b:b + 1], 1036, K), (d["idx"][b], None, d["nm"][b:b + 1], 8, K),
                     (d["sc"][b], None, d["nm"][b:b + 1], 4, K), (d["junc"][b], None, d["njunc"][b:b + 1], 1036, CJ)]
        for b in range(2 * B):
            jobs.append((d["lines"][b], None, d["nlines"][b:b + 1], 32, CL))
        cap_bytes = sum((j[3] * j[4] + 15) // 16 * 16 for j in jobs)
        sets.append(dict(L=torch.empty((B, H, W), dtype=torch.uint8, device=dev), R=torch.empty((B, H, W), dtype=torch.uint8, device=dev), d=d,
                         plan=ctx.copy_rows_plan(jobs), packed=torch.zeros((cap_bytes,), dtype=torch.uint8, device=dev),
                         off=torch.zeros((len(jobs) + 1,), dtype=torch.int64, device=dev), h_packed=torch.zeros((cap_bytes,), dtype=torch.uint8).pin_memory(),
                         h_off=torch.zeros((len(jobs) + 1,), dtype=torch.int64).pin_memory(),
                         ev_in=torch.cuda.Event(), ev_done=torch.cuda.Event(), ev_cnt=torch.cuda.Event(), ev_out=torch.cuda.Event()))
    bytes_in = 2 * B * H * W
    moved = {"d2h": 0, "matches": 0, "lines": 0}

    def take(st):
        """the consumer: step i - 2's packed rows are in the pinned block; job j's rows start at h_off[j] (jobs 0-5 are the count arrays)"""
        off, hp = st["h_off"], st["h_packed"]
        cnt = lambda j, n: hp[int(off[j]):int(off[j]) + 4 * n].view(torch.int32)
        fnd = cnt(5, 3 * B)
        if int(fnd[:2 * B].max()) > CL or int(fnd[2 * B:].max()) > CJ:
            raise SystemExit("bench: line / junction capacity overflow")
        moved["matches"], moved["lines"] = float(cnt(2, B).float().mean()), float(cnt(3, 2 * B).float().mean())

    def queue(i):
        st = sets[i % 2]
        if i >= 2:
            st["ev_out"].synchronize()
            take(st)
            s_in.wait_event(st["ev_done"])                   # step i - 2's kernels have read this set's images
        with torch.cuda.stream(s_in):
            st["L"].copy_(Lp, non_blocking=True); st["R"].copy_(Rp, non_blocking=True)
            st["ev_in"].record(s_in)
        s_cmp.wait_event(st["ev_in"])
        if i >= 2:
            s_cmp.wait_event(st["ev_out"])                   # (the packed block of this set has left the device)
        d = st["d"]
        ctx.stereo_plnet_batch_dev(st["L"], st["R"], d["fl"], d["fr"], d["nl"], d["nr"], d["lines"], d["nlines"], d["junc"], d["njunc"], d["idx"], d["sc"], d["nm"],
                                   d["found"], stream=s_cmp.cuda_stream)
        ctx.pack_rows_dev(st["plan"], st["packed"], st["off"], stream=s_cmp.cuda_stream)
        st["ev_done"].record(s_cmp)
        s_cnt.wait_event(st["ev_done"])
        with torch.cuda.stream(s_cnt):
            st["h_off"].copy_(st["off"], non_blocking=True)
            st["ev_cnt"].record(s_cnt)

    def finish(j):
        """step j's packed rows to the host: the host has the size now"""
        st = sets[j % 2]
        st["ev_cnt"].synchronize()
        total = int(st["h_off"][-1])
        with torch.cuda.stream(s_out):
            st["h_packed"][:total].copy_(st["packed"][:total], non_blocking=True)
            st["ev_out"].record(s_out)
        moved["d2h"] += total + st["h_off"].numel() * 8

    for i in range(warm):
        queue(i)
        if i > 0:
            finish(i - 1)
    if warm:
        finish(warm - 1)
    cm.barrier(dev, world)
    moved["d2h"] = 0
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        queue(i)
        if i > warm:
            finish(i - 1)
    finish(warm + steps - 1)
    cm.barrier(dev, world)
    dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
    take(sets[(warm + steps - 1) % 2])
    bytes_out = moved["d2h"] / steps
    return dict(pairs_per_s=B * steps * world / dt, ms_per_step=dt / steps * 1e3, steps=steps,
                h2d_bytes_per_step=bytes_in, d2h_bytes_per_step=bytes_out, pcie_gbs={"h2d": bytes_in * steps / dt / 1e9, "d2h": bytes_out * steps / dt / 1e9},
                matches_mean_last_step=moved["matches"], lines_mean_last_step=moved["lines"],
                what=(f"pinned host images -> H2D ({bytes_in / 1e6:.1f} MB per step, copy-in stream) -> the same step + airfe_pack_rows_dev (compute stream: counts and valid "
                      f"feature / match / line / junction rows back to back in one device block) -> D2H of the offsets, then of the packed bytes ({bytes_out / 1e6:.1f} MB per "
                      f"step, copy-out stream, copy engine) -> pinned host memory; two buffer sets in flight: while step i computes, the host finishes step i - 1's copies "
                      f"and takes step i - 2's results"))


def run(args, rank, world, local, dev):
    from airslam_amd import api, synth, weights
    from airslam_amd import dist as adist
    B, H, W, K = args.pairs, args.height, args.width, args.max_keypoints
    plnet = args.detector == "plnet"
    sp = weights.synthetic_plnet_s0(1234) if plnet else weights.synthetic_superpoint(1234)
    lg = weights.synthetic_lightglue(1234)
    ctx = api.Context(superpoint=sp, lightglue=lg, plnet_s1=cm.S1_PACK if plnet else None, device=local, precision=1 if args.dtype == "fp16" else 0,
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

    # --inflight 2: consecutive steps alternate between TWO contexts (own arena, own stream, own result buffers; the same resident images), so that two steps are
    # in flight on the device: the matcher of step i runs beside the encoder of step i + 1.  A step stays one airfe_stereo_plnet_batch_dev call over `--pairs` pairs.
    second = None
    if args.inflight > 1:
        if args.inflight != 2 or args.workload != "stereo":
            raise SystemExit("--inflight: 1 or 2, stereo workload only")
        c2 = api.Context(superpoint=sp, lightglue=lg, plnet_s1=cm.S1_PACK if plnet else None, device=local, precision=1 if args.dtype == "fp16" else 0,
                         matcher_precision=1 if args.matcher_dtype == "fp16" else 0, max_batch=B,
                         enc_chunk=args.chunk, max_keypoints=K, image_width=W, image_height=H, tuning=args.tuning, line_precision=args.line_precision)
        second = dict(ctx=c2, stream=torch.cuda.Stream(device=dev), fl=torch.zeros_like(fl), fr=torch.zeros_like(fr), nl=torch.zeros_like(nl), nr=torch.zeros_like(nr),
                      idx=torch.zeros_like(idx), sc=torch.zeros_like(sc), nm=torch.zeros_like(nm))
        if plnet:
            second.update(lines=torch.zeros_like(lines), nlines=torch.zeros_like(nlines), junc=torch.zeros_like(junc), njunc=torch.zeros_like(njunc), found=torch.zeros_like(found))
    ctxs = [ctx] + ([second["ctx"]] if second else [])

    def points_step(slot=0):
        if slot:
            b = second
            b["ctx"].stereo_batch_dev(L, R, b["fl"], b["fr"], b["nl"], b["nr"], b["idx"], b["sc"], b["nm"], stream=b["stream"].cuda_stream)
            return
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

    gatherer = None
    if world > 1:
        from airslam_amd import seq as aseq
        gatherer = aseq.MatchGatherer(1, B, K, dev, buffers=2)
        gatherer2 = aseq.MatchGatherer(1, B, K, dev, buffers=2) if second else None
        args.collective["per_step"] = ("one packed gather of the match lists to rank 0 on a side stream behind an event (airslam_amd.seq.MatchGatherer, K = 1, two buffer "
                                       "sets: the next step's kernels do not wait for the collective)")

    def step(slot=0):
        if slot:               # the second context's turn (--inflight 2)
            b = second
            if plnet:
                b["ctx"].stereo_plnet_batch_dev(L, R, b["fl"], b["fr"], b["nl"], b["nr"], b["lines"], b["nlines"], b["junc"], b["njunc"], b["idx"], b["sc"], b["nm"], b["found"],
                                                stream=b["stream"].cuda_stream)
            else:
                points_step(1)
            if world > 1:
                gatherer2.add(b["idx"], b["sc"], b["nm"], stream=b["stream"])
            return
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
        if world > 1:          # the step's match lists to rank 0: copied on the compute stream, gathered on a SIDE stream behind an event;
            gatherer.add(idx, sc, nm, stream=stream)      # barrier() drains it (torch.cuda.synchronize)

    torch.cuda.synchronize(dev)
    for i in range(args.warmup * len(ctxs)):
        step(i % len(ctxs))
    cm.barrier(dev, world)
    # Timed region: only the dominant kernel's stage carries HIP events (on the launch stream); bracketing EVERY stage
    # costs ~8 % of a step, so the full per-stage table comes from a second, untimed pass below.
    if not args.no_profile:
        for c_ in ctxs:
            c_.profile(stages=[cm.DOMINANT_STAGE])
    host = cm.HostClock()
    t0 = time.perf_counter()
    for i in range(args.steps):
        with host:
            step(i % len(ctxs))
    cm.barrier(dev, world)
    dt = time.perf_counter() - t0
    dom = None
    if not args.no_profile:
        doms = [c_.profile_read()[cm.DOMINANT_STAGE] for c_ in ctxs]
        dom = {k: sum(d[k] for d in doms) for k in doms[0]}
    for c_ in ctxs:
        c_.profile(False)
    dt = adist.max_over_ranks(dt, dev)
    host_ms = adist.all_over_ranks(host.ms(), dev)            # per rank: wall time the host spends queueing one step (the rest of a step it is free)
    host_cpu_ms = adist.all_over_ranks(host.cpu_ms(), dev)    # ... and the CPU time of that (a launch that blocks on a full queue is wall time only)
    first_core = adist.all_over_ranks(float(args.cores[0]) if args.cores else -1.0, dev)
    n_cores = adist.all_over_ranks(float(len(args.cores)) if args.cores else 0.0, dev)
    stages = {}
    if not args.no_profile and args.stage_steps > 0:          # every rank takes part: a step contains the match gather when world > 1
        ctx.profile(True)
        for _ in range(args.stage_steps):
            step()
        torch.cuda.synchronize(dev)
        stages = ctx.profile_read()
        ctx.profile(False)
    cm.barrier(dev, world)
    headline = plnet and not track and not frontend
    points_only = None
    if headline:            # the point-only step (Detect(left, right, features) with use_superpoint = 1) on the same context and inputs
        for _ in range(args.warmup):
            points_step()
        cm.barrier(dev, world)
        if second:
            points_step(1)
            cm.barrier(dev, world)
        tp = time.perf_counter()
        for i in range(args.steps):
            points_step(i % len(ctxs))
        cm.barrier(dev, world)
        points_only = B * args.steps * world / adist.max_over_ranks(time.perf_counter() - tp, dev)
        step()                         # (the counts reported below are the PLNet step's)
        cm.barrier(dev, world)
    h2h = None
    if headline and args.io_steps > 0:
        h2h = host_to_host(ctx, dev, ls, rs, B, H, W, K, CL, CJ, args.io_steps, min(args.warmup, 4), world)
        step()
        cm.barrier(dev, world)

    if rank == 0:
        total_pairs = B * args.steps * world
        ms_step = dt / args.steps * 1e3
        resident = total_pairs / dt
        io_host = args.io == "host" and h2h is not None
        config = {"workload": f"{B} synthetic {W}x{H} uint8 stereo pairs per step per GPU, " + ("in pinned HOST memory (PCIe both ways inside the timed region); " if io_host
                              else "resident in HBM; ") + f"max_keypoints={K}, nms_radius=4, LightGlue 9 layers; seeded synthetic weights (reference ONNX files are absent)",
                  "pairs_per_step_per_gpu": B, "internal_resolution": 512, "parallelism": f"frame-sharded x{world}",
                  "keypoints_left_right_mean": [float(nl.float().mean()), float(nr.float().mean())],
                  "matches_mean": float(nm.float().mean()), "detector": args.detector}
        out = cm.line(
            args,
            metric=("tracked frames/sec (normal-frame step: 1x " + ("PLNet @512x512 internal: points + lines" if plnet else "SuperPoint-VGG detect")
                    + " on the new frame + LightGlue against the last keyframe)") if track else
                   "stereo detect+match pairs/sec (" + ("2x PLNet @512x512 internal: points + lines, junctions on the left" if plnet
                                                       else "2x SuperPoint-VGG detect @512x512 internal") + " + LightGlue match)",
            value=h2h["pairs_per_s"] if io_host else resident, unit="frames/s" if track else "pairs/s", world=world,
            steps=h2h["steps"] if io_host else args.steps, warmup=args.warmup, ms_per_step=h2h["ms_per_step"] if io_host else ms_step, config=config,
            parity_dtype=("fp16 storage / fp32 accumulate: the reference's own engine type (kFP16, src/super_point.cpp:97, src/light_glue.cpp:115) and the only "
                          "2-byte type inside the north-star tolerances (descriptors 4e-4 cosine, LightGlue 0.03 of 0.05); bf16 FAILS them (2e-2 cosine, "
                          "0.25 log-assignment: DESIGN.md §1) and is selectable with --dtype bf16 --matcher-dtype bf16 only as a non-compliant speed run"),
            collective=args.collective)
        out["host"] = {"queue_ms_per_step_per_rank": host_ms, "queue_cpu_ms_per_step_per_rank": host_cpu_ms,
                       "cores_per_rank": [[int(f), int(f) + int(n) - 1] if n else None for f, n in zip(first_core, n_cores)],
                       "note": "wall / CPU time the host thread spends queueing one step's launches (the entries are asynchronous; a launch call that finds the queue "
                               "full waits: wall time only); the device needs ms_per_step for them.  cores_per_rank: [first, last] core of each rank's own share "
                               "(os.sched_setaffinity by local rank), null = not pinned"}
        if io_host:
            out["value_resident"] = resident
            out["ms_per_step_resident"] = ms_step
            out["steps_resident"] = args.steps
        if h2h is not None:
            out["host_to_host"] = dict(h2h, ratio_to_resident=h2h["pairs_per_s"] / resident)
        if args.tuning:
            out["config"]["tuning"] = args.tuning
        out["config"]["steps_in_flight"] = len(ctxs)
        if second:
            out["config"]["steps_in_flight_note"] = ("consecutive steps alternate between two contexts (own arena, stream and result buffers; the same resident images): a step is "
                                                     "still ONE airfe_stereo_plnet_batch_dev call over pairs_per_step_per_gpu pairs, and step i + 1 is queued while step i runs; the "
                                                     "dominant kernel's launch duration is measured WITH the other step's kernels beside it")
            if not (torch.equal(nm, second["nm"]) and torch.equal(idx, second["idx"])):
                raise SystemExit("bench: the two contexts in flight returned different matches for the same pairs")
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
        if headline:
            out["config"]["points_only_pairs_per_s"] = points_only
            out["config"]["points_only_note"] = ("the point-only step (2x detect + LightGlue, airfe_stereo_batch_dev: `--detector superpoint`) timed on the "
                                                 "same context and inputs over the same number of steps")
        if dom:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
            traffic, tsrc = None, None
            tdoc, tf, tage = cm.counter_profile("hbm_traffic.json")
            if tdoc:      # measured in separate --pmc passes (never together with other tracing), see profiles/README.md
                rec = tdoc.get("conv1_fused", {})
                if rec.get("hbm_bytes_per_launch") and rec.get("images_per_launch"):      # per image x the images one launch covers in THIS run
                    traffic = rec["hbm_bytes_per_launch"] / rec["images_per_launch"] * (2.0 * B * args.steps / max(dom["launches"], 1))
                tsrc = "profiles/" + os.path.basename(tf) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch of that kernel, scaled to this run's images per launch)"
            pdoc, pf, page = cm.counter_profile("pmc_summary.json")
            util, usrc = cm.encoder_mfma_util(pdoc, pf)
            out["roofline"] = {"bound": "mfma", "mfma_util_counters": util, "mfma_util_source": usrc, "kernel": "conv64r_kernel<POOL, FUSE1A> (conv1a + conv1b + 2x2 max-pool in one launch over an encoder chunk)",
                               "achieved": ach, "peak": cm.PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / cm.PEAK_MFMA_TFLOPS,
                               "traffic": traffic, "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1), "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                               "launches": dom["launches"],
                               # `achieved` / `frac` are measured live in THIS run (HIP events on the launch stream); `traffic` and `mfma_util_counters` come from
                               # committed rocprofv3 PMC passes and are reported ONLY when those passes were taken on the kernel sources this process runs:
                               "counters_age": {"traffic": tage, "mfma_util": page,
                                                "rule": "profile_csrc_sha == tree_csrc_sha (sha256 over csrc/*.hip, csrc/*.h, include/airfe*.h), else null"}}
        if stages:
            out["stages_note"] = f"separate untimed pass of {args.stage_steps} steps with every stage bracketed by events"
            out["stages"], fl_step = cm.stage_table(stages, args.stage_steps)
            if out.get("roofline"):
                # the WHOLE step against the matrix peak: algorithmic FLOPs of every matrix stage of one step / the timed ms_per_step / peak
                out["roofline"]["step_frac"] = fl_step / (ms_step * 1e-3) / 1e12 / cm.PEAK_MFMA_TFLOPS
                out["roofline"]["step_gflop"] = fl_step / 1e9
        if world == 1 and args.cpu_pairs > 0 and not track and not frontend:
            from . import cpu
            out["cpu_baseline"] = cpu.stereo(sp, lg, H, W, args.cpu_pairs, K, s1=weights.load_pack(cm.S1_PACK) if plnet else None, gpu_nmatch=nm.cpu().numpy())
            cb = out["cpu_baseline"]["same_pairs_as_gpu"]
            if cb and args.dtype == "fp16" and args.matcher_dtype == "fp16":       # a bench whose outputs drifted from the oracle's says so ON the line (ADVICE r04:
                out["cpu_baseline"]["parity_ok"] = bool(cb["max_abs_count_diff"] <= max(12, 0.15 * cb["cpu_matches_mean"]))      # an assert here lost the line)
                if not out["cpu_baseline"]["parity_ok"]:
                    print(f"bench.py: GPU and CPU-oracle match counts disagree on the same pairs: {cb}", file=sys.stderr)
        print(json.dumps(out))
    ctx.close()
    if second:
        second["ctx"].close()
    if world > 1:
        torch.distributed.destroy_process_group()
