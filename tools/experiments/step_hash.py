"""md5 of every output of the default keyframe step (airfe_stereo_plnet_batch_dev, 16 pairs): run once per library variant and compare."""
import hashlib
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from airslam_amd import api, synth, weights
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)

B, K = 16, 400
dev = torch.device("cuda", 0)
ctx = api.Context(tuning=tuning_from_env(), superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", lightglue=weights.synthetic_lightglue(1234),
                  max_batch=B, enc_chunk=32, max_keypoints=K)
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
o = dict(fl=z(B, K, 259), fr=z(B, K, 259), nl=z(B, dt=torch.int32), nr=z(B, dt=torch.int32), lines=z(2 * B, 1024, 4, dt=torch.float64),
         nlines=z(2 * B, dt=torch.int32), junc=z(B, 1024, 259), njunc=z(B, dt=torch.int32), idx=z(B, K, 2, dt=torch.int32), sc=z(B, K),
         nm=z(B, dt=torch.int32), found=z(3 * B, dt=torch.int32))
ctx.stereo_plnet_batch_dev(L, R, o["fl"], o["fr"], o["nl"], o["nr"], o["lines"], o["nlines"], o["junc"], o["njunc"], o["idx"], o["sc"], o["nm"], o["found"])
ctx.sync()
print(" ".join("%s=%s" % (k, hashlib.md5(v.cpu().numpy().tobytes()).hexdigest()[:10]) for k, v in o.items()), "lines", int(o["nlines"].sum()), "matches", int(o["nm"].sum()))
