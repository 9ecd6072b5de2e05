#!/usr/bin/env python3
"""Probe: the C++ sequence driver as two pipelined groups at the bench's sizes — which time-step fails, with what, and whether serialised kernels change it."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from airslam_amd import api, seq, synth, weights

H, W, K = 480, 752, 400
S, N = int(sys.argv[1]), int(sys.argv[2])
G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
arrs = [synth.stereo_sequence_arrays((N, H, W, 10 + s, 40)) for s in range(S)]
Ld = torch.from_numpy(np.stack([a[0] for a in arrs], 1)).to(dev)
Rd = torch.from_numpy(np.stack([a[1] for a in arrs], 1)).to(dev)
lg = weights.synthetic_lightglue(1234)
s1 = os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")
pol = seq.KeyframeConfig(image_width=W, image_height=H, tracking_point_rate=0.2, min_init_stereo_feature=60, min_num_match=30, max_num_match=80)
Sg = S // G
ctxs = []
for _ in range(G):
    common = dict(device=0, precision=1, matcher_precision=1, max_keypoints=K, image_width=W, image_height=H, check_launches=1)
    kf = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1=s1, lightglue=lg, max_batch=max(Sg, 2), enc_chunk=max(min(2 * Sg, 128), 2), **common)
    nf = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=lg, max_batch=max(Sg, 2), enc_chunk=max(min(Sg, 128), 2), **common)
    ctxs.append((kf, nf))
groups = [seq.NativeSequences(k, n, Sg, pol, device=dev, temporal_buffers=True) for k, n in ctxs]
pipe = seq.NativePipeline(groups)
seen = []
pipe.on_group_done = lambda x: seen.append(x.counts().copy())
try:
    for t in range(N):
        pipe.step(Ld[t, :S], Rd[t, :S])
    pipe.flush()
    c = np.concatenate(seen)
    print("ok:", S, "sequences x", N, "frames in", G, "groups; keyframes", int((c[:, 0] != 0).sum()), "promotions", int(c[:, 2].sum()), "max lines", int(c[:, 8].max()))
except api.AirfeError as e:
    print("FAILED at time-step", t, ":", e)
