"""Inside the K loop of the LightGlue block's ffn.0 x half (4 slabs, 64 features per wave, 6-7 token tiles): shader cycles per trip of every wave.
Needs airslam_amd/libairfe_T2.so.tmp (tools/build_timing_variants.sh: kernels_lgblockf.hip with -DLF_TIMING2).
    python tools/lf_timing2.py [pairs ...]        (on an MI355X; it copies the variant over libairfe.so of the working copy)
A trip = one 64-wide K chunk: 8 A fragments prefetched for the next trip, then 2 halves x 2 groups of (4 or 3 B fragments read, 16 or 12 MFMAs of 16 cycles)."""
import ctypes as C, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
subprocess.check_call(["cp", "airslam_amd/libairfe_T2.so.tmp", "airslam_amd/libairfe.so"])
import torch
from airslam_amd import api, weights, _lib
from planted import normalised, planted_pair
for pairs in [int(v) for v in sys.argv[1:]] or [64]:
    ctx = api.Context(lightglue=weights.synthetic_lightglue(1234), max_batch=2 * pairs, max_keypoints=400)
    f0, f1 = planted_pair(400, 400, 3)
    a = torch.from_numpy(np.repeat(normalised(f0)[None], pairs, 0)).cuda(); b = torch.from_numpy(np.repeat(normalised(f1)[None], pairs, 0)).cuda()
    n = torch.full((pairs,), 400, dtype=torch.int32, device="cuda")
    idx = torch.zeros((pairs, 400, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros((pairs, 400), device="cuda"); nm = torch.zeros((pairs,), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm)
    ctx.sync()
    out = (C.c_ulonglong * 16)()
    lib = _lib.lib()
    lib.airfe_dbg_lf2(out, 1)
    reps = 5
    for _ in range(reps):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm)
    ctx.sync()
    lib.airfe_dbg_lf2(out, 0)
    pref, vm, rd, lgkm, mma, mov, trips, groups, mfmas = (float(out[i]) for i in range(9))
    print(f"\n{pairs} pairs ({pairs * 800} tokens), {reps} forwards x 18 block launches: {trips:.0f} K-loop trips timed (every wave of every workgroup: 4 per pass), "
          f"{mfmas / trips:.1f} MFMAs per trip; shader cycles per trip and wave")
    rows = [("weight prefetch for the next trip issued (8 global_load_dwordx4)", pref), ("s_waitcnt vmcnt: this trip's weights landed (requested one trip earlier)", vm),
            ("B fragments requested (ds_read_b128 x 3-4 per group, 4 groups)", rd), ("s_waitcnt lgkmcnt(0): B fragments landed (LDS latency + conflicts + the other wave's reads)", lgkm),
            ("MFMAs issued (16x16x32: 16 cycles of matrix pipe each; two waves share the pipe)", mma), ("end of trip: cur = nxt register moves + loop", mov)]
    tot = sum(v for _, v in rows)
    for nme, v in rows:
        print(f"  {nme:100s} {v / trips:8.0f}   {v / tot:6.1%}")
    print(f"  {'trip':100s} {tot / trips:8.0f}   own matrix-pipe time {mfmas / trips * 16:.0f} = {mfmas * 16 / tot:.2f} of it; with the SIMD's other wave {2 * mfmas * 16 / tot:.2f}")
    print(f"  per group of MFMAs: B read issue {rd / groups:.0f}, wait {lgkm / groups:.0f}, MFMA issue {mma / groups:.0f} cycles ({mfmas / groups:.1f} MFMAs = {mfmas / groups * 16:.0f} pipe cycles)")
    ctx.close()
