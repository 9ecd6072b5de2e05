"""Batch-1 latency through the reference-shaped host API (FeatureDetector.DetectStereo + PointMatcher.MatchingPoints):
host uint8 images in, host feature matrices / matches out, i.e. PCIe and the synchronisations included.
    python tools/latency_b1.py            (on an MI355X)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airslam_amd import api, synth, weights  # noqa: E402


def main():
    ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=weights.synthetic_lightglue(1234), max_batch=2,
                      enc_chunk=2, max_keypoints=400)
    det, pm = api.FeatureDetector(ctx), api.PointMatcher(ctx, 752, 480, 0)
    pairs = [synth.stereo_pair(480, 752, s) for s in range(4)]
    for left, right in pairs:                       # warm-up
        ok, f0, f1 = det.DetectStereo(left, right)
        pm.MatchingPoints(f0, f1)
    t_det, t_match = [], []
    for i in range(40):
        left, right = pairs[i % 4]
        t0 = time.perf_counter()
        ok, f0, f1 = det.DetectStereo(left, right)
        t1 = time.perf_counter()
        pm.MatchingPoints(f0, f1)
        t2 = time.perf_counter()
        t_det.append(t1 - t0)
        t_match.append(t2 - t1)
    d, m = np.median(t_det) * 1e3, np.median(t_match) * 1e3
    ctx.profile(True)
    reps = 10
    for i in range(reps):
        left, right = pairs[i % 4]
        ok, f0, f1 = det.DetectStereo(left, right)
        pm.MatchingPoints(f0, f1)
    st = {k: round(v["ms"] / reps, 4) for k, v in ctx.profile_read().items() if v["launches"]}
    ctx.profile(False)
    print(json.dumps({"metric": "batch-1 stereo detect+match latency, host buffers in/out (PCIe + syncs included)",
                      "detect_stereo_ms": d, "match_ms": m, "pair_ms": d + m, "pairs_per_s": 1e3 / (d + m),
                      "keypoints": [int(f0.shape[1]), int(f1.shape[1])], "gpu_stage_ms": st,
                      "gpu_stage_total_ms": round(sum(st.values()), 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
