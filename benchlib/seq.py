"""--workload seq: BASELINE.json configs[3] — per rank S stereo SEQUENCES of --frames frames (synth.stereo_sequence, seeds 10 + rank * S + s), driven exactly as
MapBuilder::ExtractFeatureThread drives the front end (src/map_builder.cc:83-141 with the shipped use_superpoint: 1): PLNet stereo keyframes, SuperPoint-only
normal frames matched against the last keyframe, promotions.  S = 1 runs the one-call host entries (latency regime); S > 1 runs the S sequences of a time-step in
lock-step through the device-resident entries — by default from the C++ driver (include/airfe_seq.h), two groups of sequences half a step apart
(airslam_amd.seq.NativePipeline), `--seq-driver python` = the Python driver of round 5 (airslam_amd.seq.BatchedSequences).  Every K = 8 frames the temporal match
lists go to rank 0 in one collective per group on a side stream.  A step = one time-step = one frame of each of the S sequences; value = frames/s of the whole job."""
import json
import os
import time

import numpy as np
import torch

from . import common as cm

CF = {n: i for i, n in enumerate(("frame_type", "candidate", "promoted", "dropped", "enough_match", "good_stereo_point", "n_left", "n_right", "n_lines_left",
                                  "n_lines_right", "n_junctions", "n_stereo", "n_matches"))}


def _schedule(counts):
    """counts: [steps * S, 13] int32 records (airfe_seq_frame's integer fields) of the timed frames"""
    c = np.asarray(counts).reshape(-1, 13)
    col = lambda n: c[:, CF[n]]
    tm, sm, ln = col("n_matches"), col("n_stereo"), col("n_lines_left")
    return {"frames": int(len(c)), "keyframe_candidates": int(col("candidate").sum()), "keyframes": int((col("frame_type") != 0).sum()),
            "promotions": int(col("promoted").sum()), "normal_frames": int((col("frame_type") == 0).sum()), "dropped_before_init": int(col("dropped").sum()),
            "temporal_matches_mean": float(tm[tm >= 0].mean()) if (tm >= 0).any() else 0.0,
            "stereo_matches_mean": float(sm[sm >= 0].mean()) if (sm >= 0).any() else 0.0,
            "lines_mean_keyframe_left": float(ln[ln >= 0].mean()) if (ln >= 0).any() else 0.0}


def _counts_of(results):
    """FrameResults (the Python drivers) -> the same [S, 13] records"""
    n = lambda a: -1 if a is None else len(a)
    return np.array([[r.frame_type, r.candidate, r.promoted, r.dropped, r.enough_match, r.good_stereo_point, len(r.features_left), n(r.features_right), n(r.lines_left),
                      n(r.lines_right), n(r.junctions), n(r.stereo_idx), n(r.matches_idx)] for r in results], np.int32)


def run(args, rank, world, local, dev):
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
    with mp.get_context("spawn").Pool(min(Smax, len(args.cores) if args.cores else (os.cpu_count() or 1))) as pool:     # (2.6 s per 200-frame sequence on one core)
        arrs = pool.map(synth.stereo_sequence_arrays, jobs)
    Lh = np.stack([a[0] for a in arrs], 1)                 # [frames][Smax][H][W]
    Rh = np.stack([a[1] for a in arrs], 1)
    Ld, Rd = torch.from_numpy(Lh).to(dev), torch.from_numpy(Rh).to(dev)          # resident in HBM before the clock starts (Smax x frames x 0.72 MB)
    lg = weights.synthetic_lightglue(1234)
    policy = dict(tracking_point_rate=args.tracking_point_rate, min_init_stereo_feature=args.min_init_stereo, min_num_match=args.min_num_match,
                  max_num_match=max(80, args.min_num_match + 10))
    pol = seq.KeyframeConfig(image_width=W, image_height=H, **policy)
    prec, mprec = (1 if args.dtype == "fp16" else 0), (1 if args.matcher_dtype == "fp16" else 0)
    common = dict(device=local, precision=prec, matcher_precision=mprec, max_keypoints=K, image_width=W, image_height=H, tuning=args.tuning)

    def contexts(S):
        kf = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=cm.S1_PACK, lightglue=lg, max_batch=max(S, 2), enc_chunk=max(min(2 * S, args.chunk), 2), **common)
        nf = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=lg, max_batch=max(S, 2), enc_chunk=max(min(S, args.chunk), 2), **common)
        return kf, nf

    def run_S(S):
        native = S > 1 and args.seq_driver == "native"
        G = 1
        if native:
            G = args.groups if args.groups > 0 else (2 if S >= 4 and S % 2 == 0 else 1)
            if S % G:
                raise SystemExit(f"--groups {G} does not divide --sequences {S}")
        Sg = S // G
        ctxs = [contexts(Sg) for _ in range(G)]
        gats = [seq.MatchGatherer(KG, Sg, K, dev) for _ in range(G)]
        counts, lat, pending = [], [], []
        split = None
        if S == 1:
            fe = seq.SequenceFrontEnd(ctxs[0][0], ctxs[0][1], pol)

            def step(t):
                r = fe.step(Lh[t, 0], Rh[t, 0])          # host images in, host matrices out: the batch-1 API takes host buffers (PCIe included)
                if r.matches_idx is not None:
                    m = len(r.matches_idx)
                    idx = torch.zeros((1, K, 2), dtype=torch.int32); sc = torch.zeros((1, K)); idx[0, :m] = torch.from_numpy(r.matches_idx); sc[0, :m] = torch.from_numpy(r.matches_score)
                    h = gats[0].add(idx.to(dev, non_blocking=True), sc.to(dev, non_blocking=True), torch.tensor([m], dtype=torch.int32).to(dev, non_blocking=True))
                    if h is not None:
                        pending.append(h)
                counts.append(_counts_of([r]))

            flush = lambda: None
            driver = "airslam_amd.seq.SequenceFrontEnd (one-call host entries)"
        elif not native:
            bs = seq.BatchedSequences(ctxs[0][0], ctxs[0][1], S, pol, device=dev, copy_results=False)      # (results are read inside the step that made them)

            def step(t):
                rs = bs.step(Ld[t, :S], Rd[t, :S])
                nt = sum(r.matches_idx is not None for r in rs)
                if nt:                                     # (device tensors of this step's temporal matches, rows in tset order; the others count 0)
                    h = gats[0].add(bs.tidx[:nt], bs.tsc[:nt], bs.tnm[:nt], stream=bs.stream)
                    if h is not None:
                        pending.append(h)
                counts.append(_counts_of(rs))

            flush = lambda: None
            driver = "airslam_amd.seq.BatchedSequences (Python lock-step driver over the *_batch_dev entries)"
        else:
            groups = [seq.NativeSequences(k, n, Sg, pol, device=dev, temporal_buffers=True) for k, n in ctxs]
            order = []                                     # (group, counts) in completion order

            def done(x):
                g = groups.index(x)
                c = x.counts()
                nt = int((c[:, CF["n_matches"]] >= 0).sum())
                if nt:
                    h = gats[g].add(x.tidx[:nt], x.tsc[:nt], x.tnm[:nt], stream=x.stream)
                    if h is not None:
                        pending.append(h)
                order.append((g, c))
                if len(order) == G:                        # one time-step complete: its S records in sequence order
                    counts.append(np.concatenate([c_ for _, c_ in sorted(order, key=lambda p: p[0])]))
                    order.clear()
            if G == 1:
                def step(t):
                    groups[0].step_raw(Ld[t, :S], Rd[t, :S])
                    done(groups[0])
                flush = lambda: None
            else:
                pipe = seq.NativePipeline(groups)
                pipe.on_group_done = done

                def step(t):
                    pipe.step(Ld[t, :S], Rd[t, :S])
                flush = pipe.flush
            driver = (f"include/airfe_seq.h (C++ lock-step driver, csrc/airfe_seq.hip) x {G} group(s) of {Sg} sequence(s)"
                      + (", half a step apart (airslam_amd.seq.NativePipeline)" if G > 1 else ""))

        for t in range(warm):
            step(t)
        flush()
        cm.barrier(dev, world)
        if native:
            for x in groups:
                x.wall_split()                             # (reset)
        n_warm_records = len(counts)
        host = cm.HostClock()
        t0 = time.perf_counter()
        for t in range(warm, frames):
            ta = time.perf_counter()
            with host:
                step(t)
            lat.append(time.perf_counter() - ta)
        flush()
        for h in pending:
            h.result()
        cm.barrier(dev, world)
        dt = adist.max_over_ranks(time.perf_counter() - t0, dev)
        timed = frames - warm
        if native:
            ws = [x.wall_split() for x in groups]
            split = {"queue_device_work": sum(w["queue_s"] for w in ws) / timed * 1e3, "wait_for_device": sum(w["wait_s"] for w in ws) / timed * 1e3,
                     "host_side_of_the_loop": sum(w["host_s"] for w in ws) / timed * 1e3, "host_syncs": sum(w["host_syncs"] for w in ws) / timed / G,
                     "python_around_the_driver": host.ms() - sum(w["queue_s"] + w["wait_s"] + w["host_s"] for w in ws) / timed * 1e3}
        elif S > 1:
            split = {"queue_device_work": bs.t_queue / frames * 1e3, "wait_for_device": bs.t_wait / frames * 1e3, "host_side_of_the_loop": bs.t_host / frames * 1e3,
                     "host_syncs": bs.syncs / frames}
        n_gathers = sum(g.gathers for g in gats)
        # the matrix work of a time-step against the peak: a few more time-steps (frames wrap around) with every stage of every context bracketed by events
        roof = None
        if S > 1 and not args.no_profile and args.stage_steps > 0:
            every = [c for pair in ctxs for c in pair]
            for c in every:
                c.profile(True)
            ns = max(args.stage_steps, 8)
            for t in range(ns):
                step(t % frames)
            flush()
            torch.cuda.synchronize(dev)
            st = cm.merge_stages([c.profile_read() for c in every])
            for c in every:
                c.profile(False)
            tab, fl_step = cm.stage_table(st, ns)
            ms_b = sum(v["ms"] for v in st.values()) / ns
            ms_step = dt / timed * 1e3
            roof = {"bound": "mfma", "kernel": "all bracketed matrix stages of one time-step (algorithmic FLOPs / their event time)",
                    "achieved": fl_step / (ms_b * 1e-3) / 1e12 if ms_b > 0 else 0.0, "peak": cm.PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": (fl_step / (ms_b * 1e-3) / 1e12 / cm.PEAK_MFMA_TFLOPS) if ms_b > 0 else 0.0, "traffic": None,
                    "step_gflop": fl_step / 1e9, "step_frac": fl_step / (ms_step * 1e-3) / 1e12 / cm.PEAK_MFMA_TFLOPS,
                    "bracketed_ms_per_time_step": ms_b,
                    "note": "a time-step of S sequences is S small batches' worth of latency-bound launches: step_frac is the whole time-step against the dense MFMA peak",
                    "stages": {k: {"ms_per_step": round(v["ms_per_step"], 4), "frac": v.get("frac"), "bound": v.get("bound")} for k, v in tab.items()}}
            counts[:] = counts[:n_warm_records + timed]
        if native:
            for x in groups:
                x.close()
        for k, n in ctxs:
            k.close(); n.close()
        timed_counts = counts[n_warm_records:n_warm_records + timed]
        a = np.asarray(lat) * 1e3
        return dict(S=S, dt=dt, latency_ms={"p50": float(np.percentile(a, 50)), "p99": float(np.percentile(a, 99)), "mean": float(a.mean()), "max": float(a.max())}, frames_per_s=S * timed * world / dt, ms_per_step=dt / timed * 1e3, host_ms_per_step=host.ms(),
                    schedule=_schedule(np.concatenate(timed_counts)), gathers=n_gathers, counts=counts, wall_split_ms_per_step=split, roofline=roof,
                    driver=driver, groups=G)

    runs = {S: run_S(S) for S in sweep}
    head = runs[args.sequences]
    host_ms = adist.all_over_ranks(head["host_ms_per_step"], dev)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        from . import cpu as cpu_leg
        nfr = min(args.cpu_pairs + 2, frames)
        cpu = cpu_leg.sequence(cm.S1_PACK, lg, W, H, K, policy, Lh, Rh, nfr, [int(c[0, CF["frame_type"]]) for c in head["counts"][:nfr]])
    if rank == 0:
        out = cm.line(
            args,
            metric="stereo sequence frames/sec (BASELINE configs[3]: per sequence PLNet stereo keyframes + SuperPoint-only normal frames matched against the last keyframe "
                   "+ promotions, map_builder.cc:83-141 with use_superpoint: 1)",
            value=head["frames_per_s"], unit="frames/s", world=world, steps=frames - warm, warmup=warm, ms_per_step=head["ms_per_step"],
            config={"workload": f"{args.sequences} synthetic {W}x{H} stereo sequence(s) per GPU x {frames} frames (seeds 10 + rank * S + s, a new scene every {scene_len} frames, "
                                f"2-3 px pan per frame), images resident in HBM" + (" and in host memory (S = 1 goes through the batch-1 host entries: PCIe included)" if args.sequences == 1 else "")
                                + f"; keyframe policy = AddKeyframeCheck of vo_euroc.yaml with tracking_point_rate {args.tracking_point_rate} and min_init_stereo_feature {args.min_init_stereo} "
                                f"(the synthetic matcher weights match ~35 % of the keypoints; the yaml's 0.65 / 90 would make every second frame a keyframe candidate and leave "
                                f"some sequences uninitialised for a scene); max_keypoints={K}; seeded synthetic weights except PLNet stage 1 (real)",
                    "sequences_per_gpu": args.sequences, "frames": frames, "gather_every_frames": KG, "gathers": head["gathers"], "schedule": head["schedule"],
                    "driver": head["driver"], "groups": head["groups"]},
            host={"python_ms_per_time_step_per_rank": host_ms, "wall_split_ms_per_step": head["wall_split_ms_per_step"],
                  "cores_of_rank0": [args.cores[0], args.cores[-1]] if args.cores else None},
            latency_ms_per_time_step=head["latency_ms"],
            sweep={str(S): {"frames_per_s": r["frames_per_s"], "ms_per_time_step": r["ms_per_step"], "latency_ms": r["latency_ms"], "schedule": r["schedule"], "driver": r["driver"],
                            "wall_split_ms_per_step": r["wall_split_ms_per_step"], "step_frac": r["roofline"] and r["roofline"]["step_frac"]} for S, r in runs.items()},
            roofline=head["roofline"], cpu_baseline=cpu, collective=args.collective)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()
