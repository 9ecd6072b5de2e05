"""Determinism soak of the default keyframe step at the bench size: N steps, every output compared ON THE DEVICE with the first run's
(torch.equal: no host copies, ~10 ms per step).  AIRFE_OVERLAP_LINES=0 puts everything on one stream."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from airslam_amd import api, synth, weights
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B, K = 64, 400
dev = torch.device("cuda", 0)
ctx = api.Context(tuning=tuning_from_env(), superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", lightglue=weights.synthetic_lightglue(1234),
                  max_batch=B, enc_chunk=128, max_keypoints=K)
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
def bufs():
    return dict(fl=z(B, K, 259), fr=z(B, K, 259), nl=z(B, dt=torch.int32), nr=z(B, dt=torch.int32), lines=z(2 * B, 1024, 4, dt=torch.float64),
                nlines=z(2 * B, dt=torch.int32), junc=z(B, 1024, 259), njunc=z(B, dt=torch.int32), idx=z(B, K, 2, dt=torch.int32), sc=z(B, K),
                nm=z(B, dt=torch.int32), found=z(3 * B, dt=torch.int32))
def run(b):
    ctx.stereo_plnet_batch_dev(L, R, b["fl"], b["fr"], b["nl"], b["nr"], b["lines"], b["nlines"], b["junc"], b["njunc"], b["idx"], b["sc"], b["nm"], b["found"])
    ctx.sync()
ref = bufs(); run(ref)
two = [bufs(), bufs()]
bad = {}
for i in range(N):
    b = two[i & 1]; run(b)
    for k in ref:
        if not torch.equal(ref[k], b[k]):
            bad.setdefault(k, []).append(i)
print("soak OVERLAP_LINES=%s: %d steps, outputs that ever differed from the first run: %s" % (os.environ.get("AIRFE_OVERLAP_LINES", "1 (default)"), N, {k: v[:5] for k, v in bad.items()} or "none"))
