import ctypes as C, json, os, subprocess, sys
import numpy as np
os.environ["AIRFE_LG_BLOCK_FORM"] = "1"
sys.path.insert(0, os.getcwd())
subprocess.check_call(["cp", "airslam_amd/libairfe_T.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, synth, weights, _lib
ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=weights.synthetic_lightglue(1234), max_batch=128, enc_chunk=32)
B = 64
rng = np.random.default_rng(0)
f0 = torch.zeros((B, 400, 259)); f1 = torch.zeros((B, 400, 259))
f0[:, :, 1] = torch.rand(B, 400) * 700; f0[:, :, 2] = torch.rand(B, 400) * 400; f0[:, :, 3:] = torch.nn.functional.normalize(torch.randn(B, 400, 256), dim=-1)
f1[:, :, 1] = torch.rand(B, 400) * 700; f1[:, :, 2] = torch.rand(B, 400) * 400; f1[:, :, 3:] = torch.nn.functional.normalize(torch.randn(B, 400, 256), dim=-1)
n = torch.full((B,), 400, dtype=torch.int32)
f0, f1, n0, n1 = f0.cuda(), f1.cuda(), n.cuda(), n.clone().cuda()
idx = torch.zeros((B, 400, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros((B, 400), device="cuda"); nm = torch.zeros((B,), dtype=torch.int32, device="cuda")
for _ in range(3):
    ctx.match_lightglue_batch_dev(f0, n0, f1, n1, idx, sc, nm)
ctx.sync()
l = _lib.lib()
out = (C.c_longlong * (512 * 8 * 12))()
l.airfe_dbg_lf(out)
a = np.array(out[:], dtype=np.int64).reshape(512, 8, 12)[:400, :, :10]
names = ["attn load", "GEMM1+msg", "GEMM2 msg half", "wait x", "GEMM2 x half", "LN stats", "GELU+write h", "GEMM3", "epilogue"]
first = a[:, 0, 0].argsort()
for nme, grp in (("first round", first[:256]), ("second round", first[256:])):
    t = a[grp]                                    # [blocks, waves, stamps]
    t0 = t[:, :, 0].min(1, keepdims=True)
    rel = (t - t0[:, :, None]) / 100.0            # us since the block's first stamp
    print(nme)
    for i in range(1, 10):
        arr = rel[:, :, i]
        print(f"  after {names[i-1]:16s} first wave {arr.min(1).mean():6.2f}  last wave {arr.max(1).mean():6.2f}  skew {(arr.max(1)-arr.min(1)).mean():5.2f}")
print("launch span us", (a[:, :, 9].max() - a[:, :, 0].min()) / 100)
