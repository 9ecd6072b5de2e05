#!/usr/bin/env python3
"""Headline benchmark: stereo detect+match pairs/s/GPU @752x480 (BASELINE.json `metric`).

A step = one pass of the hot path (2x SuperPoint-VGG detect + 1x LightGlue match, 512x512 internal resolution as the
reference does, src/plnet.cpp:17-21) over one batch of `--pairs` synthetic stereo pairs that are already resident in
HBM.  One process per GPU; ranks shard pairs (weak scaling) and gather their matches to rank 0 every step.

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

DOMINANT_STAGE = "conv3x3_cin64"   # largest single share of a step; its events stay on inside the timed region
PEAK_MFMA_TFLOPS = 2500.0   # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def cpu_baseline(sp, lg, h, w, n_pairs, max_kp, warm=3):
    """The CPU oracle (PyTorch-CPU fp32 networks + numpy restatement of the reference's C++ post-processing) timed on
    the host cores, on a bounded sample of the same workload: `warm` untimed pairs, then the MEDIAN per-pair time of
    `n_pairs` pairs (median of >= 20 after 3 warm-ups)."""
    from airslam_amd import synth
    from oracle import ref_nets, ref_post
    torch.set_num_threads(min(os.cpu_count() or 1, 32))   # more threads than this only adds sync overhead at batch 1
    base = synth.stereo_pair(h, w, 100)
    pairs = [(np.roll(base[0], 7 * i, axis=1), np.roll(base[1], 7 * i, axis=1)) for i in range(n_pairs + warm)]   # inputs ready before the clock
    times, nmatch = [], []
    for i, (left, right) in enumerate(pairs):
        t0 = time.perf_counter()
        feats = []
        for img in (left, right):
            x, ws, hs = ref_post.process_image(img)
            heat, desc = ref_nets.superpoint_forward(sp, x[None])
            feats.append(ref_post.keypoints_decoder(ref_post.simple_nms(heat[0], 4), desc[0], 0.004, 4, max_kp, ws, hs))
        k = 0
        if feats[0].shape[0] and feats[1].shape[0]:
            a = ref_post.normalize_keypoints(feats[0], w, h, 0.5)
            b = ref_post.normalize_keypoints(feats[1], w, h, 0.5)
            s = ref_nets.lightglue_forward(lg, a[:, 1:3], a[:, 3:], b[:, 1:3], b[:, 3:])
            k = len(ref_post.filter_matches(s, 0.1)[0])
        if i >= warm:                                   # the first pairs warm the thread pool and the allocator
            times.append(time.perf_counter() - t0)
            nmatch.append(k)
    med = float(np.median(times))
    return dict(value=1.0 / med, unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"median of {n_pairs} synthetic {w}x{h} stereo pairs after {warm} warm-ups ({sum(times):.1f} s), fp32 PyTorch-CPU "
                       f"oracle + numpy post-processing, {float(np.mean(nmatch)):.0f} matches per pair")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per step per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=752)
    ap.add_argument("--max-keypoints", type=int, default=400)
    ap.add_argument("--chunk", type=int, default=32, help="images per pass through the full-resolution conv layers")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="detector (encoder) storage type; fp16 = the reference's kFP16 engines and the type that meets the parity gates")
    ap.add_argument("--matcher-dtype", default="fp16", choices=["bf16", "fp16"],
                    help="matcher storage type (fp16 = the reference's kFP16 engines, light_glue.cpp:115)")
    ap.add_argument("--cpu-pairs", type=int, default=20, help="CPU-baseline sample size (0 = skip)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--stage-steps", type=int, default=3, help="extra untimed steps for the per-stage table")
    args = ap.parse_args()

    from airslam_amd import api, dist as adist, synth, weights

    rank, world, local = adist.init_from_env(os.environ.get("AIRFE_DIST_BACKEND"))   # default: nccl (= RCCL) on GPUs
    if os.environ.get("AIRFE_ONE_DEVICE"):     # test hook: several ranks share GPU 0 (with AIRFE_DIST_BACKEND=gloo; RCCL refuses that)
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, H, W, K = args.pairs, args.height, args.width, args.max_keypoints

    sp = weights.synthetic_superpoint(1234)
    lg = weights.synthetic_lightglue(1234)
    ctx = api.Context(superpoint=sp, lightglue=lg, device=local, precision=1 if args.dtype == "fp16" else 0,
                      matcher_precision=1 if args.matcher_dtype == "fp16" else 0, max_batch=B,
                      enc_chunk=args.chunk, max_keypoints=K, image_width=W, image_height=H)

    ls, rs = synth.stereo_batch(B, H, W, 1000 + rank)
    L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
    fl = torch.zeros((B, K, 259), device=dev); fr = torch.zeros((B, K, 259), device=dev)
    nl = torch.zeros((B,), dtype=torch.int32, device=dev); nr = torch.zeros((B,), dtype=torch.int32, device=dev)
    idx = torch.zeros((B, K, 2), dtype=torch.int32, device=dev)
    sc = torch.zeros((B, K), device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)

    stream = torch.cuda.Stream(device=dev)
    sh = stream.cuda_stream

    def step():
        ctx.stereo_batch_dev(L, R, fl, fr, nl, nr, idx, sc, nm, stream=sh)
        if world > 1:
            with torch.cuda.stream(stream):
                adist.gather_matches(idx, sc, nm, dst=0)

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

    if rank == 0:
        total_pairs = B * args.steps * world
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "stereo detect+match pairs/sec (2x SuperPoint-VGG detect @512x512 internal + LightGlue match)",
            "value": total_pairs / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate",
            "data": "synthetic",
            "config": {"workload": f"{B} synthetic {W}x{H} uint8 stereo pairs per step per GPU, resident in HBM; "
                                   f"max_keypoints={K}, nms_radius=4, LightGlue 9 layers; seeded synthetic weights "
                                   f"(reference ONNX files are absent)",
                       "pairs_per_step_per_gpu": B, "internal_resolution": 512, "parallelism": f"frame-sharded x{world}",
                       "keypoints_left_right_mean": [float(nl.float().mean()), float(nr.float().mean())],
                       "matches_mean": float(nm.float().mean())},
        }
        if dom:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
            traffic, tsrc = None, None
            tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01e_hbm_traffic.json")
            if os.path.exists(tf):      # measured in separate --pmc passes (never together with other tracing), see profiles/README.md
                with open(tf) as fh:
                    traffic = json.load(fh).get("conv3x3_cin64_stage", {}).get("hbm_bytes_per_launch")
                tsrc = "profiles/r01e_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
            out["roofline"] = {"bound": "mfma", "kernel": "conv64r_kernel (conv1a fused into conv1b + pool, conv2a, conv2b + pool, conv3a)",
                               "achieved": ach, "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_TFLOPS,
                               "traffic": traffic, "traffic_source": tsrc,
                               "algorithmic_bytes_per_launch": dom["bytes"] / max(dom["launches"], 1), "avg_launch_ms": dom["ms"] / max(dom["launches"], 1),
                               "launches": dom["launches"]}
        if stages:
            tot = sum(s["ms"] for s in stages.values())
            out["stages_note"] = f"separate untimed pass of {args.stage_steps} steps with every stage bracketed by events"
            out["stages"] = {k: {"ms_per_step": v["ms"] / args.stage_steps, "share": v["ms"] / tot if tot else 0,
                                 "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None,
                                 "algo_gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
                             for k, v in stages.items() if v["launches"]}
        if world == 1 and args.cpu_pairs > 0:
            out["cpu_baseline"] = cpu_baseline(sp, lg, H, W, args.cpu_pairs, K)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
