"""md5 of the LightGlue scores of a few planted pairs and of one batched stereo step: run once per library variant, compare the lines
(the pipelined attention kernel does the same arithmetic in the same order as the one it replaces: the hashes must be equal)."""
import hashlib
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from airslam_amd import api, synth, weights
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)
from planted import normalised, planted_pair

lg = weights.synthetic_lightglue(1234)
ctx = api.Context(tuning=tuning_from_env(), lightglue=lg, superpoint=weights.synthetic_superpoint(1234), max_batch=16, enc_chunk=16, max_keypoints=400)
out = []
for n0, n1 in [(400, 400), (317, 400), (64, 65), (33, 400), (400, 96), (129, 97)]:
    f0, f1 = planted_pair(n0, n1, 7 * n0 + n1)
    a, b = np.ascontiguousarray(normalised(f0)[:, 1:]), np.ascontiguousarray(normalised(f1)[:, 1:])
    s = ctx.lightglue_scores(a, b)
    out.append("%dx%d %s nan=%d" % (n0, n1, hashlib.md5(s.tobytes()).hexdigest()[:12], int(np.isnan(s).sum())))
B = 16
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
fl, fr, nl, nr, idx, sc, nm = z(B, 400, 259), z(B, 400, 259), z(B, dt=torch.int32), z(B, dt=torch.int32), z(B, 400, 2, dt=torch.int32), z(B, 400), z(B, dt=torch.int32)
ctx.stereo_batch_dev(L, R, fl, fr, nl, nr, idx, sc, nm)
ctx.sync()
out.append("stereo16 sc %s idx %s nm %s" % tuple(hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:12] for t in (sc, idx, nm)))
heat, _, _ = ctx.detector_maps(4)
out.append("heat4 %s fl %s" % (hashlib.md5(np.ascontiguousarray(heat).tobytes()).hexdigest()[:12], hashlib.md5(fl.cpu().numpy().tobytes()).hexdigest()[:12]))
one = ctx.detect_points(ls[0])
h1, _, _ = ctx.detector_maps(1)
out.append("heat1 %s feat1 %s" % (hashlib.md5(np.ascontiguousarray(h1).tobytes()).hexdigest()[:12], hashlib.md5(one.tobytes()).hexdigest()[:12]))
print(" | ".join(out))
