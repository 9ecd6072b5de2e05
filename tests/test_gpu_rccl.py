"""RCCL under the code that uses it: `airslam_amd.dist` picks backend "nccl" (= RCCL on ROCm) on GPUs, but a 1-GPU box can only ever form
a group of ONE rank (RCCL refuses two ranks on one device), and with one rank the product skips the collective.  These tests form that one-rank "nccl" group
in a child process and FORCE the collectives through it: the library loads, the communicator initialises, `gather` / `all_reduce` accept the packed int32
match buffer straight off the matcher's stream and the side stream of the K = 8 cadence, and return its bytes.  The 8-GPU scaling curve itself is the driver's."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import torch.distributed as dist
from airslam_amd import api, dist as adist, seq, weights
from planted import planted_pair
rank, world, local = adist.init_from_env()                      # backend = "nccl" because a GPU is visible
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B, K = 4, 400
ctx = api.Context(lightglue=weights.synthetic_lightglue(1234), max_batch=B, max_keypoints=K, check_launches=1)
f0 = torch.zeros((B, K, 259)); f1 = torch.zeros((B, K, 259)); n0 = torch.zeros((B,), dtype=torch.int32); n1 = torch.zeros((B,), dtype=torch.int32)
for b in range(B):
    a, c = planted_pair(400 - 13 * b, 380 - 7 * b, 50 + b)
    f0[b, :len(a)] = torch.from_numpy(a); f1[b, :len(c)] = torch.from_numpy(c); n0[b] = len(a); n1[b] = len(c)
f0, f1, n0, n1 = (x.to(dev) for x in (f0, f1, n0, n1))
idx = torch.zeros((B, K, 2), dtype=torch.int32, device=dev); sc = torch.zeros((B, K), device=dev); nm = torch.zeros((B,), dtype=torch.int32, device=dev)
st = torch.cuda.Stream(device=dev)
ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm, stream=st.cuda_stream)
with torch.cuda.stream(st):                                      # the collective rides the matcher's own stream, right behind it
    gi, gs, gn = adist.gather_matches(idx, sc, nm, dst=0, force=True)
st.synchronize()
ok_gather = bool(torch.equal(gi, idx) and torch.equal(gs, sc) and torch.equal(gn, nm))
mx = adist.max_over_ranks(3.25, dev, force=True)
# the K = 8 cadence on a side stream: 16 "frames" of S = B sequences -> two gathers of 8 x B rows
g = seq.MatchGatherer(8, B, K, dev)
handles = []
for t in range(16):
    ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm, stream=st.cuda_stream)
    h = g.add(idx, sc, nm, stream=st)
    if h is not None:
        handles.append(h)
outs = [h.result() for h in handles]
ok_side = len(outs) == 2 and all(o[0].shape == (8 * B, K, 2) and all(torch.equal(o[0][k * B:(k + 1) * B], idx) and torch.equal(o[2][k * B:(k + 1) * B], nm) for k in range(8)) for o in outs)
dist.barrier()
ver = getattr(torch.cuda.nccl, "version", lambda: None)()
print("RCCL_RESULT " + json.dumps(dict(backend=dist.get_backend(), ok_gather=ok_gather, ok_side=ok_side, max=mx, matches=gn.tolist(), gathers=g.gathers, nccl_version=str(ver))))
ctx.close()
dist.destroy_process_group()
'''


def test_one_rank_nccl_group_carries_the_match_gather(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AIRFE_DIST_FORCE_INIT="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_RESULT ")]
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-3000:])
    res = json.loads(line[0][len("RCCL_RESULT "):])
    from gpu_common import diag
    diag("rccl_one_rank", **{k: (str(v) if isinstance(v, list) else v) for k, v in res.items()})
    assert res["backend"] == "nccl" and res["ok_gather"] and res["ok_side"] and res["max"] == 3.25 and res["gathers"] == 2
    assert min(res["matches"]) >= 100, "the gather must carry real match lists"
