"""md5 of LightGlue log-assignment matrices (two pair shapes, batch 1 and a batch of 8 through the 112-token passes) computed with whatever libairfe.so is in place:
the A/B scripts run it once per library variant to show that a kernel rewrite kept the bits.    python tools/lib_scores_hash.py"""
import hashlib, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from airslam_amd import api, weights
from planted import normalised, planted_pair
h = hashlib.md5()
ctx = api.Context(lightglue=weights.synthetic_lightglue(1234), max_batch=8, max_keypoints=400)
for n0, n1, seed in ((400, 400, 3), (317, 400, 5)):
    f0, f1 = planted_pair(n0, n1, seed)
    h.update(ctx.lightglue_scores(np.ascontiguousarray(normalised(f0)[:, 1:]), np.ascontiguousarray(normalised(f1)[:, 1:])).tobytes())
B = 8
pairs = [planted_pair(400 - 13 * i, 390 - 7 * i, 40 + i) for i in range(B)]
f0 = torch.zeros((B, 400, 259)); f1 = torch.zeros((B, 400, 259))
for i, (a, b) in enumerate(pairs):
    f0[i, :a.shape[0]] = torch.from_numpy(a); f1[i, :b.shape[0]] = torch.from_numpy(b)
n0 = torch.tensor([p[0].shape[0] for p in pairs], dtype=torch.int32).cuda(); n1 = torch.tensor([p[1].shape[0] for p in pairs], dtype=torch.int32).cuda()
idx = torch.zeros((B, 400, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros((B, 400), device="cuda"); nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
ctx.match_lightglue_batch_dev(f0.cuda(), n0, f1.cuda(), n1, idx, sc, nm)
ctx.sync()
h.update(idx.cpu().numpy().tobytes()); h.update(sc.cpu().numpy().tobytes()); h.update(nm.cpu().numpy().tobytes())
print("scores md5", h.hexdigest(), "matches", nm.cpu().numpy().tolist())
