"""Constants for the SYNTHETIC PLNet stage-0 line head (airslam_amd/weights.py: PLNET_S0_LOI_BIAS / _THIN_BIAS / _AUX_BIAS).

plnet_s0.onnx is absent upstream, plnet_s1.onnx (the line-verification head) is real.  A He-uniform stage-0 line head feeds the real
stage-1 head features it has never seen: every candidate line scores < 0.5 and the line filter / junction path downstream carries
nothing (1 line of ~1100 candidates).  This script finds, by gradient ascent on the restated stage-1 head (oracle/ref_nets.py, the real
weights in tests/golden/plnet_s1.airfe), a constant LOI / thin / aux feature vector for which the head is confident (logit margin ~ +2.5)
— used as the BIAS of the synthetic head's feature channels, with its weights scaled down, the features a line samples are that vector
plus an image-dependent perturbation: most candidates pass 0.5, a good part pass the reference's 0.75, some fail.
    python tools/plnet_s0_calibrate.py        -> prints the three arrays"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from airslam_amd import weights  # noqa: E402

w = {k: torch.from_numpy(v) for k, v in weights.load_pack(os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")).items()}


def head(f, thin, aux):
    x = torch.cat([f, f, thin.repeat_interleave(30), aux.repeat_interleave(30)])[None]
    lin = lambda n, v: torch.nn.functional.linear(v, w[n + ".weight"], w[n + ".bias"])
    h = lin("fc2.4", torch.relu(lin("fc2.2", torch.relu(lin("fc2.0", x)))))
    h = h + torch.relu(lin("fc2_res.0", x[:, 256:]))
    lg = lin("fc2_head", h)[0]
    return lg[1] - lg[0]


torch.manual_seed(0)
p = torch.zeros(136, requires_grad=True)
opt = torch.optim.Adam([p], lr=0.02)
for it in range(3000):
    opt.zero_grad()
    m = head(p[:128], p[128:132], p[132:136])
    loss = (m - 2.5) ** 2 + 0.02 * (p ** 2).sum()          # a confident but finite margin with the smallest features that give it
    loss.backward()
    opt.step()
with torch.no_grad():
    m = head(p[:128], p[128:132], p[132:136])
    print("margin", float(m), "score", float(torch.sigmoid(m)), "|f| rms", float(p[:128].pow(2).mean().sqrt()), file=sys.stderr)
    # sensitivity: margin under N(0, s) perturbations of the features
    for s in (0.05, 0.1, 0.2, 0.4):
        ms = torch.stack([head(p[:128] + s * torch.randn(128), p[128:132] + s * torch.randn(4), p[132:136] + s * torch.randn(4)) for _ in range(400)])
        sc = torch.sigmoid(ms)
        print(f"noise {s}: margin mean {float(ms.mean()):.2f} sd {float(ms.std()):.2f}; score > 0.5: {float((sc > 0.5).float().mean()):.2f}, > 0.75: {float((sc > 0.75).float().mean()):.2f}", file=sys.stderr)
np.set_printoptions(precision=4, suppress=True, linewidth=140)
v = p.detach().numpy().astype(np.float32)
print("PLNET_S0_LOI_BIAS = np.array(%s, dtype=np.float32)" % np.array2string(v[:128], separator=", "))
print("PLNET_S0_THIN_BIAS = np.array(%s, dtype=np.float32)" % np.array2string(v[128:132], separator=", "))
print("PLNET_S0_AUX_BIAS = np.array(%s, dtype=np.float32)" % np.array2string(v[132:136], separator=", "))
