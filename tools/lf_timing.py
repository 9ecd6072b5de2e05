"""Per-phase wall-clock breakdown of lg_blockf_kernel (wave 0 of every workgroup, summed) over LightGlue forwards at a given pair count.
Needs a build of kernels_lgblockf.hip with -DLF_TIMING linked as airslam_amd/libairfe_T.so.tmp:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLF_TIMING -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_T.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_T.so.tmp /tmp/lf_T.o <the other objects of airslam_amd/csrc/build>
    python tools/lf_timing.py [pairs ...]        (on an MI355X; it copies the variant over libairfe.so of the working copy)
With -DLF_TIMING -DLF_TWICE every workgroup runs its pass twice and the second run's timers are printed beside the first's (LF_TWICE=1 in the environment)."""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
subprocess.check_call(["cp", "airslam_amd/libairfe_T.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, weights, _lib
from planted import normalised, planted_pair
# (fold_out_proj, the default: no out-projection phase; the "msg half" is the attention half, under which the x tile lands)
names = ["wait: attn tile + first weights (launch start -> first barrier)", "out-proj (4 slabs) + msg pack + barrier", "ffn.0 msg / attention half (4 slabs) + x-tile DMA issue",
         "wait: x tile + barrier", "ffn.0 x half (4 slabs) + LayerNorm partial sums", "barrier (sums)", "LayerNorm + GELU + pack, residual rows fetched",
         "barrier (h tile)", "ffn.3 (8 slabs) + residual + x stores", "2 barriers + x tile back into LDS", "folded q | k units (4 slabs each) + rotary + stores",
         "folded V unit (4 slabs, transposed) + stores"]
for pairs in [int(v) for v in sys.argv[1:]] or [1, 8, 64]:
    ctx = api.Context(lightglue=weights.synthetic_lightglue(1234), max_batch=2 * pairs, max_keypoints=400)
    f0, f1 = planted_pair(400, 400, 3)
    a = torch.from_numpy(np.repeat(normalised(f0)[None], pairs, 0)).cuda(); b = torch.from_numpy(np.repeat(normalised(f1)[None], pairs, 0)).cuda()
    n = torch.full((pairs,), 400, dtype=torch.int32, device="cuda")
    idx = torch.zeros((pairs, 400, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros((pairs, 400), device="cuda"); nm = torch.zeros((pairs,), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm)
    ctx.sync()
    out = (C.c_ulonglong * 32)()
    lib = _lib.lib()
    lib.airfe_dbg_lf(out, 1)
    reps = 5
    for _ in range(reps):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm)
    ctx.sync()
    lib.airfe_dbg_lf(out, 0)
    t = np.array(out[:12], dtype=np.float64); wgs = float(out[15])
    t2 = np.array(out[16:28], dtype=np.float64); twice = out[31] > 0
    print(f"\n{pairs} pairs ({pairs * 800} tokens): {int(wgs / reps / 18)} workgroups per launch, 18 launches per forward; per workgroup, wave 0 (us)" + (" | the same pass run again by the same workgroup:" if twice else ":"))
    for i, nme in enumerate(names):
        print(f"  {nme:75s} {t[i] / wgs / 100.0:7.2f}" + (f" | {t2[i] / wgs / 100.0:7.2f}" if twice else ""))
    print(f"  {'total':75s} {t.sum() / wgs / 100.0:7.2f}" + (f" | {t2.sum() / wgs / 100.0:7.2f}" if twice else ""))
    ctx.close()
