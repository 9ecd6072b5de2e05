"""Per-phase wall-clock breakdown of plnet_s1_kernel over one batched PLNet step (wave 0 of every workgroup, summed).
Needs a build of kernels_ext.hip with -DS1_TIMING linked as airslam_amd/libairfe_T.so.tmp:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DS1_TIMING -c airslam_amd/csrc/kernels_ext.hip -o /tmp/ke_T.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_T.so.tmp /tmp/ke_T.o <the other objects of airslam_amd/csrc/build>
    python tools/s1_timing.py [line_precision]   (on an MI355X; it copies the variant over libairfe.so of the working copy; 3 = plnet_s1h_kernel, same slots)"""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd())
subprocess.check_call(["cp", "airslam_amd/libairfe_T.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, synth, weights, _lib
B = 64
ctx = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", lightglue=weights.synthetic_lightglue(1234),
                  max_batch=B, enc_chunk=64, line_precision=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).cuda(), torch.from_numpy(rs).cuda()
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
fl, fr, nl, nr = z(B, 400, 259), z(B, 400, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
idx, sc, nm = z(B, 400, 2, dt=torch.int32), z(B, 400), z(B, dt=torch.int32)
lines, nlines = z(2 * B, 1024, 4, dt=torch.float64), z(2 * B, dt=torch.int32)
junc, njunc = z(B, 1024, 259), z(B, dt=torch.int32)
for _ in range(2):
    ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm)
ctx.sync()
out = (C.c_ulonglong * 16)()
lib = _lib.lib()
lib.airfe_dbg_s1(out, 1)
ctx.stereo_plnet_batch_dev(L, R, fl, fr, nl, nr, lines, nlines, junc, njunc, idx, sc, nm)
ctx.sync()
lib.airfe_dbg_s1(out, 0)
a = np.array(out[:11], dtype=np.float64)
tiles = a[9]
names = ["wait at the tile's first barrier", "header (pairs -> junctions, keep -> proposal) + barrier", "sampling", "barrier after sampling",
         "layer 0 (120 MFMAs + the junction terms)", "residual layer (120 MFMAs)", "barrier (x tile dead)", "h0 write, layer 2, h1 write, layer 4, h0 write (128 MFMAs, 3 barriers)",
         "head + soft-max (2 barriers)"]
print(f"{int(tiles)} tiles of 32 lines; per tile, wave 0 (us):")
for i, nme in enumerate(names):
    print(f"  {nme:90s} {a[i] / tiles / 100.0:7.2f}")
if a[10]:
    print(f"  {'prologue (tables to LDS, the first tile of a workgroup: three dependent header loads), per tile':90s} {a[10] / tiles / 100.0:7.2f}")
print(f"  {'total':90s} {(a[:9].sum() + a[10]) / tiles / 100.0:7.2f}")
