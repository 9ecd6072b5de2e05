"""bench.py's `cpu_baseline` legs: the ONLY place of the bench that imports oracle/ (the checker, timed on the GPU box's host cores on a bounded sample —
never the thing measured as `value`)."""
import os
import time

import numpy as np
import torch


def stereo(sp, lg, h, w, n_pairs, max_kp, warm=3, s1=None, line_threshold=0.75, line_length_threshold=50.0, gpu_nmatch=None, sg=None, seed=1000):
    """The CPU oracle (PyTorch-CPU fp32 networks + numpy restatement of the reference's C++ post-processing) timed on
    the host cores, on a bounded sample of the same workload: `warm` untimed pairs, then the MEDIAN per-pair time of
    `n_pairs` pairs (median of >= 20 after 3 warm-ups).  s1 (the stage-1 weights) selects the PLNet step: one trunk
    pass per image feeding the point heads AND the line branch, wireframe_matcher, stage 1, line filter, junctions on the left.  sg (SuperGlue weights): the matcher
    is SuperGlue (18 layers, 100 Sinkhorn iterations, decode at 0.2: BASELINE configs[4]) instead of LightGlue."""
    from airslam_amd import synth
    from oracle import margins, ref_chain, ref_nets, ref_post
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32))   # more threads than this only adds sync overhead at batch 1
    # the SAME images rank 0 puts on the GPU (synth.stereo_batch(B, h, w, 1000)): its first n_pairs + warm pairs, ready before the clock
    ls, rs = synth.stereo_batch(n_pairs + warm, h, w, seed)
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
        if sg is not None and feats[0].shape[0] and feats[1].shape[0]:
            a = ref_post.normalize_keypoints(feats[0], w, h, 0.7)
            b = ref_post.normalize_keypoints(feats[1], w, h, 0.7)
            z = ref_nets.superglue_forward(sg, a[:, 1:3], a[:, 0], a[:, 3:], b[:, 1:3], b[:, 0], b[:, 3:])
            k = int((ref_post.superglue_decode(z, 0.2)[0] >= 0).sum())
        elif feats[0].shape[0] and feats[1].shape[0]:
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
                       f"oracle + numpy post-processing ({'PLNet points + lines + junctions' if s1 is not None else 'SuperPoint'} + {'SuperGlue' if sg is not None else 'LightGlue'}), "
                       f"{float(np.mean(nmatch)):.0f} matches per pair")


def sequence(s1_path, lg, W, H, K, policy, Lh, Rh, n_frames, dev_types):
    """the oracle's restatement of the feature thread's loop (oracle/ref_seq.Chain) on sequence 0's first frames (bounded: ~20 frames of fp32 PyTorch-CPU + numpy),
    taking its own decisions; dev_types: the frame types the device took on the same frames"""
    from airslam_amd import weights
    from oracle import ref_seq
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32))
    chain = ref_seq.Chain(weights.synthetic_plnet_s0(1234), weights.synthetic_superpoint(1234), weights.load_pack(s1_path), lg, W, H, K, policy=policy)
    ts, types = [], []
    for t in range(n_frames):
        ta = time.perf_counter()
        o = chain.step(Lh[t, 0], Rh[t, 0])
        ts.append(time.perf_counter() - ta); types.append(o["frame_type"])
    return dict(value=(n_frames - 2) / sum(ts[2:]), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                sample=f"frames 2..{n_frames - 1} of sequence 0 ({sum(ts[2:]):.1f} s; frames 0-1 warm the thread pool): oracle/ref_seq.Chain — fp32 PyTorch-CPU networks + numpy "
                       f"post-processing, the same loop taking its own keyframe decisions",
                same_schedule_as_gpu=(types == list(dev_types)), frame_types_cpu=types, frame_types_gpu=list(dev_types))
