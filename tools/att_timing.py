"""Per-phase shader-clock breakdown of attention32_kernel (every ACTIVE wave, summed) over LightGlue forwards at a given pair count.
Needs airslam_amd/libairfe_AT.so.tmp (tools/build_timing_variants.sh: kernels_attn.hip with -DATT_TIMING).
    python tools/att_timing.py [pairs ...]        (on an MI355X; it copies the variant over libairfe.so of the working copy)
Per 64-key tile and wave the matrix pipe needs 16 MFMAs x 32 cycles = 512 cycles (8 for QK^T, 8 for PV); three waves share a SIMD."""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
subprocess.check_call(["cp", "airslam_amd/libairfe_AT.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, weights, _lib
from planted import normalised, planted_pair
names = ["prologue: Q fragments + first K / V tile landed + barrier (per wave, once)", "per tile: next tile's DMA issued, K fragments read (ds_read_b128 x 4-8), QK^T chains issued",
         "per tile: QK^T chains complete (wait for the last accumulator)", "per tile: exponentials (16-32 v_exp_f32) + partial row sums (+ first tile: row maximum, re-centring)",
         "per tile: P packed to 2 bytes, V^T fragments read, PV chains issued", "per tile: s_waitcnt vmcnt(0) — this wave's part of the next tile landed",
         "per tile: barrier", "epilogue: row sums exchanged, O scaled, packed, stored (per wave, once)"]
for pairs in [int(v) for v in sys.argv[1:]] or [64]:
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
    lib.airfe_dbg_att(out, 1)
    reps = 5
    for _ in range(reps):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm)
    ctx.sync()
    lib.airfe_dbg_att(out, 0)
    t = np.array(out[:9], dtype=np.float64)
    waves, tiles, retries = float(out[15]), float(out[13]), float(out[14])
    print(f"\n{pairs} pairs (2 x {pairs} sequences x 400 keypoints), {reps} forwards x 18 attention launches: {waves / reps / 18:.0f} active waves per launch, "
          f"{tiles / waves:.2f} tiles per wave, {retries:.0f} re-centred tiles; shader cycles per wave")
    per_wave = [t[0] / waves] + [t[i] / waves for i in range(1, 7)] + [t[7] / waves]
    per_tile = [None] + [t[i] / tiles for i in range(1, 7)] + [None]
    for i, nme in enumerate(names):
        print(f"  {nme:110s} {per_wave[i]:9.0f}" + (f"   = {per_tile[i]:7.0f} per tile" if per_tile[i] is not None else ""))
    tot = t[8] / waves
    print(f"  {'kernel start -> end of the wave':110s} {tot:9.0f}   (sum of the phases {sum(per_wave):.0f}); matrix-pipe time of the wave's own MFMAs: {tiles / waves * 512:.0f} "
          f"= {tiles / waves * 512 / tot:.2f} of its lifetime; three waves share a SIMD")
    ctx.close()
