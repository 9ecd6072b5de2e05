"""Shared helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "gpurun_out", "diag")


def diag(name: str, **kv):
    """Record numbers for post-mortem reading (gpurun only returns the tail of stdout)."""
    os.makedirs(DIAG, exist_ok=True)
    out = {}
    for k, v in kv.items():
        if isinstance(v, np.ndarray):
            v = v.tolist()
        elif isinstance(v, (np.floating, np.integer)):
            v = v.item()
        out[k] = v
    with open(os.path.join(DIAG, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print(f"[diag:{name}] " + " ".join(f"{k}={out[k]}" for k in out if not isinstance(out[k], list)))


def to_2byte(x: np.ndarray, prec: int = 0) -> np.ndarray:
    """Round fp32 to the storage type the kernels use (bf16 / fp16), back in fp32."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    t = t.to(torch.float16 if prec == 1 else torch.bfloat16).float()
    return t.numpy()


def cosine_dist(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    num = (a * b).sum(-1)
    den = np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1)
    return 1.0 - num / np.maximum(den, 1e-30)


_CTX = {}


def context(kind: str, tuning=None, **cfg):
    """Cached airfe contexts (weight packing + arena allocation is not free).  `tuning` = {airfe_tuning field: value}: the kernel-selection
    overrides airfe_create reads (include/airfe.h; the library reads no environment variables)."""
    from airslam_amd import api, weights
    key = (kind, tuple(sorted((tuning or {}).items())), tuple(sorted(cfg.items())))
    if key not in _CTX:
        sp = weights.synthetic_superpoint(1234) if "sp" in kind else None
        lg = weights.synthetic_lightglue(1234) if "lg" in kind else None
        _CTX[key] = (api.Context(superpoint=sp, lightglue=lg, tuning=tuning, check_launches=1, **cfg), sp, lg)
    return _CTX[key]
