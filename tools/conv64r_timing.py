"""Per-phase wall-clock breakdown of the fused conv1a+conv1b kernel (conv64r_kernel<POOL, FUSE1A>), per wave.
Needs a build of kernels_conv64r.hip with -DC64R_TIMING linked as airslam_amd/libairfe_T.so.tmp, e.g.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DC64R_TIMING -c airslam_amd/csrc/kernels_conv64r.hip -o /tmp/c64t.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_T.so.tmp /tmp/c64t.o <the other objects of airslam_amd/csrc/build>
    python tools/conv64r_timing.py          (on an MI355X; it copies the variant over libairfe.so of the working copy)"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd())
subprocess.check_call(["cp", "airslam_amd/libairfe_T.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, synth, weights, _lib
ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), max_batch=64, enc_chunk=64)
ls, rs = synth.stereo_batch(4, 480, 752, 3)
imgs = torch.from_numpy(np.tile(ls, (16, 1, 1))).cuda()
feat = torch.zeros((64, 400, 259), device="cuda"); n = torch.zeros((64,), dtype=torch.int32, device="cuda")
for _ in range(2):
    ctx.detect_batch_dev(imgs, feat, n)
ctx.sync()
out = (C.c_longlong * (256 * 8 * 6))()
_lib.lib().airfe_dbg_c64(out)
a = np.array(out[:], dtype=np.float64).reshape(256, 8, 6)[:, :, :5] / 100.0     # us per wave over the launch (128 tiles)
names = ["top (patch issue, acc init, first fragments)", "6 combos (MFMA + conv1a production)", "wait patch/lgkm", "barrier", "epilogue"]
tot = a.sum(2)
print("per wave, whole launch (us): total mean %.1f" % tot.mean())
for i, nme in enumerate(names):
    print(f"  {nme:48s} mean {a[:, :, i].mean():7.1f}  ({100 * a[:, :, i].mean() / tot.mean():4.1f} %)   waves 0-3 {a[:, :4, i].mean():7.1f}  waves 4-7 {a[:, 4:, i].mean():7.1f}")
