"""Is the matcher reproducible with an UNRELATED kernel stream beside it?  Fixed features (one detector pass), then N matcher passes, each
with torch work queued on another stream; scores of every pass against the first."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from airslam_amd import api, synth, weights

B, K, N = 64, 400, int(sys.argv[1]) if len(sys.argv) > 1 else 80
KIND = sys.argv[2] if len(sys.argv) > 2 else "mm"          # mm | elementwise | detector | plnet (a SECOND context working on its own stream) | none
dev = torch.device("cuda", 0)
ctx = api.Context(superpoint=weights.synthetic_superpoint(1234), lightglue=weights.synthetic_lightglue(1234), max_batch=B, enc_chunk=64, max_keypoints=K)
ls, rs = synth.stereo_batch(B, 480, 752, 1000)
L, R = torch.from_numpy(ls).to(dev), torch.from_numpy(rs).to(dev)
z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)
fl, fr, nl, nr = z(B, K, 259), z(B, K, 259), z(B, dt=torch.int32), z(B, dt=torch.int32)
ctx.detect_batch_dev(L, fl, nl); ctx.detect_batch_dev(R, fr, nr); ctx.sync()
side = torch.cuda.Stream(device=dev)
ctx2 = None
if KIND in ("detector", "plnet"):
    ctx2 = api.Context(superpoint=weights.synthetic_plnet_s0(1234), plnet_s1="tests/golden/plnet_s1.airfe", max_batch=B, enc_chunk=64, max_keypoints=K)
    f2, n2 = z(B, K, 259), z(B, dt=torch.int32)
    lines2, nlines2 = z(B, 1024, 4, dt=torch.float64), z(B, dt=torch.int32)
a = torch.randn(4096, 4096, device=dev, dtype=torch.float16); b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
big = torch.randn(64 << 20, device=dev)
def run():
    idx, sc, nm = z(B, K, 2, dt=torch.int32), z(B, K), z(B, dt=torch.int32)
    if KIND == "detector":
        ctx2.detect_batch_dev(L, f2, n2, stream=side.cuda_stream)
    elif KIND == "plnet":
        ctx2.detect_plnet_batch_dev(L, f2, n2, lines2, nlines2, stream=side.cuda_stream)
    elif KIND != "none":
        with torch.cuda.stream(side):
            for _ in range(6):
                if KIND == "mm": (a @ b)
                else: big.mul_(1.0001)
    outs = []
    for rep in range(3 if ctx2 is not None else 1):          # (three passes cover the second context's whole step: encoder, then line path)
        idx, sc, nm = z(B, K, 2, dt=torch.int32), z(B, K), z(B, dt=torch.int32)
        ctx.match_lightglue_batch_dev(fl, nl, fr, nr, idx, sc, nm)
        outs.append((idx, sc, nm))
    ctx.sync(); torch.cuda.synchronize()
    return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), torch.stack([o[2] for o in outs])
ref = run()
bad = 0
for i in range(N):
    o = run()
    if not (torch.equal(ref[0], o[0]) and torch.equal(ref[1], o[1]) and torch.equal(ref[2], o[2])):
        bad += 1
        if bad <= 3:
            dd = (ref[1] - o[1]).abs().flatten(0, 1); rows = (dd.max(1).values > 0).nonzero().flatten().tolist()
            print("  pass", i, "pairs", rows, "max abs diff %.3e" % float(dd.max()))
print(f"matcher beside '{KIND}': {bad} of {N} passes differ from the first")
