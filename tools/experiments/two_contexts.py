"""Does the matcher of one half-batch overlap the encoder of the other?  One context with 64 pairs against two contexts (own stream, own
arena) with 32 pairs each, issued back to back from one host thread; same total work per step.
    python tools/experiments/two_contexts.py [plnet|superpoint]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from airslam_amd import api, synth, weights
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..' if 'experiments' in _os.path.abspath(__file__) else '.'))
from tuning_env import tuning_from_env      # (tools/tuning_env.py: AIRFE_* environment -> airfe_tuning; the library itself reads no environment)

mode = sys.argv[1] if len(sys.argv) > 1 else "plnet"
B, K, H, W = 64, 400, 480, 752
dev = torch.device("cuda", 0)
ls, rs = synth.stereo_batch(B, H, W, 1000)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)


def make(nb, lo):
    if mode == "plnet":
        ctx = api.Context(tuning=tuning_from_env(), superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", lightglue=weights.synthetic_lightglue(1234),
                          max_batch=nb, enc_chunk=64, max_keypoints=K)
    else:
        ctx = api.Context(tuning=tuning_from_env(), superpoint=weights.synthetic_superpoint(1234), lightglue=weights.synthetic_lightglue(1234), max_batch=nb, enc_chunk=64, max_keypoints=K)
    L, R = torch.from_numpy(ls[lo:lo + nb]).to(dev), torch.from_numpy(rs[lo:lo + nb]).to(dev)
    bufs = dict(fl=z(nb, K, 259), fr=z(nb, K, 259), nl=z(nb, dt=torch.int32), nr=z(nb, dt=torch.int32), idx=z(nb, K, 2, dt=torch.int32), sc=z(nb, K),
                nm=z(nb, dt=torch.int32), lines=z(2 * nb, 1024, 4, dt=torch.float64), nlines=z(2 * nb, dt=torch.int32), junc=z(nb, 1024, 259),
                njunc=z(nb, dt=torch.int32))
    st = torch.cuda.Stream(device=dev)

    def step():
        b = bufs
        if mode == "plnet":
            ctx.stereo_plnet_batch_dev(L, R, b["fl"], b["fr"], b["nl"], b["nr"], b["lines"], b["nlines"], b["junc"], b["njunc"], b["idx"], b["sc"], b["nm"],
                                       stream=st.cuda_stream)
        else:
            ctx.stereo_batch_dev(L, R, b["fl"], b["fr"], b["nl"], b["nr"], b["idx"], b["sc"], b["nm"], stream=st.cuda_stream)
    return ctx, step, bufs


def run(steps_fns, n=60):
    for _ in range(5):
        for f in steps_fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for f in steps_fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


c1, s1, b1 = make(64, 0)
t_one = run([s1])
ca, sa, ba = make(32, 0)
cb, sb, bb = make(32, 32)
t_two = run([sa, sb])
same = torch.equal(b1["nm"][:32], ba["nm"]) and torch.equal(b1["nm"][32:], bb["nm"]) and torch.equal(b1["idx"][:32], ba["idx"])
print(f"{mode}: one context x 64 pairs {t_one:.3f} ms/step = {64 / t_one * 1e3:.0f} pairs/s;  two contexts x 32 pairs {t_two:.3f} ms/step = {64 / t_two * 1e3:.0f} pairs/s;"
      f"  same matches: {same}")
