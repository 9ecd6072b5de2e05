"""LightGlue forward time at small pair counts, fused block (32- / 64-token passes) against the four separate launches per block:
    python tools/lg_small_batch_sweep.py            (on an MI355X; prints one line per (pairs, form))
Decides GemmArgs/block_min (airfe_match.hip: lg_blockf, lightglue_dev): from which token count the fused kernel is the quicker form."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(pairs):
    import torch
    from airslam_amd import api, weights
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
    from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)
    from planted import normalised, planted_pair
    ctx = api.Context(tuning=tuning_from_env(), lightglue=weights.synthetic_lightglue(1234), max_batch=2 * pairs, max_keypoints=400)
    f0, f1 = planted_pair(400, 400, 3)
    a = torch.from_numpy(np.repeat(normalised(f0)[None], pairs, 0)).cuda(); b = torch.from_numpy(np.repeat(normalised(f1)[None], pairs, 0)).cuda()
    n = torch.full((pairs,), 400, dtype=torch.int32, device="cuda")
    idx = torch.zeros((pairs, 400, 2), dtype=torch.int32, device="cuda"); sc = torch.zeros((pairs, 400), device="cuda")
    nm = torch.zeros((pairs,), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    for _ in range(5):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm, stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.match_lightglue_batch_dev(a, n, b, n, idx, sc, nm, stream=s.cuda_stream)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call, {int(nm[0])} matches")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for pairs in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            for form, env in (("split", {"AIRFE_FUSE_LG_BLOCK": "0"}), ("fused", {"AIRFE_FUSE_LG_BLOCK": "1"}),
                              ("fused64", {"AIRFE_FUSE_LG_BLOCK": "1", "AIRFE_LGB_TOKENS": "64"}), ("fused112", {"AIRFE_FUSE_LG_BLOCK": "1", "AIRFE_LGB_TOKENS": "112"})):
                r = subprocess.run([sys.executable, __file__, str(pairs)], env=dict(os.environ, **env), capture_output=True, text=True)
                print(f"pairs {pairs:3d} tokens {pairs * 800:6d} {form:9s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
