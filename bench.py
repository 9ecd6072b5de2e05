#!/usr/bin/env python3
"""Headline benchmark: stereo detect+match pairs/s/GPU @752x480 (BASELINE.json `metric`).

A step = one pass of the hot path as the reference's front end runs it on a stereo keyframe (map_builder.cc:85-86:
`Detect(left, right, features, lines, junctions)` + `MatchingPoints(left, right)`): PLNet on both images — points, the line branch,
wireframe_matcher, stage 1 and the line filter, junctions + descriptors on the left one (feature_detector.cc:94-104) — and one LightGlue
match, at the reference's 512x512 internal resolution (src/plnet.cpp:17-21), over one batch of `--pairs` synthetic stereo pairs that
are already resident in HBM.  `--detector superpoint` is the point-only step (2x SuperPoint-VGG detect + LightGlue: what
`Detect(left, right, features)` runs with use_superpoint = 1); it is also timed in the default run and reported beside the headline.
One process per GPU; ranks shard pairs (weak scaling) and gather their matches to rank 0 every step.
This file is the command line; the workloads live in benchlib/ (stereo.py = the headline + track + frontend, seq.py = BASELINE configs[3], b1.py = batch-1 latency,
side.py = SuperGlue / loop closure / host API) and share one line builder (benchlib/common.py: `line`).

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
    ap.add_argument("--io", default="resident", choices=["resident", "host"],
                    help="stereo workload: `value` = the step with inputs resident in HBM (the contract's default) or host to host — pinned host images in, features / "
                         "lines / junctions / matches back in pinned host memory, PCIe on copy streams beside the compute stream (the resident rate rides along as value_resident)")
    ap.add_argument("--io-steps", type=int, default=20, help="steps of the host-to-host pass the default line reports beside the resident rate (0 = skip)")
    ap.add_argument("--inflight", type=int, default=1, choices=[1, 2],
                    help="stereo workload: steps in flight on the device — 2 = consecutive steps alternate between two contexts (own arena and stream), so that the matcher of "
                         "step i runs beside the encoder of step i + 1")
    ap.add_argument("--seq-driver", default="native", choices=["native", "python"],
                    help="--workload seq, S > 1: the C++ lock-step driver (include/airfe_seq.h) or round 5's Python driver (airslam_amd.seq.BatchedSequences)")
    ap.add_argument("--groups", type=int, default=0, help="--workload seq, native driver: groups of sequences half a step apart (0 = 2 from 4 sequences on, else 1)")
    ap.add_argument("--no-affinity", action="store_true", help="do not give each local rank its own share of the host cores (os.sched_setaffinity)")
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

    from airslam_amd import dist as adist

    rank, world, local = adist.init_from_env(os.environ.get("AIRFE_DIST_BACKEND"))   # default: nccl (= RCCL) on GPUs
    n_local = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # one process per GPU, each queueing hundreds of launches per step from one host thread: every local rank gets its own share of the cores
    args.cores = None if args.no_affinity else adist.pin_rank_to_cores(local, n_local)
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

    if args.workload == "track" and (args.plnet_host or args.matcher == "superglue"):
        raise SystemExit("--workload track runs the device-resident PLNet / SuperPoint + LightGlue path")
    if args.workload == "b1":
        from benchlib import b1 as wl
    elif args.workload == "seq":
        from benchlib import seq as wl
    elif args.plnet_host or args.matcher == "superglue" or args.workload == "loop":
        from benchlib import side as wl
    else:
        from benchlib import stereo as wl
    return wl.run(args, rank, world, local, dev)


if __name__ == "__main__":
    main()
