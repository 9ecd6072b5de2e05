import os, sys, numpy as np, hashlib, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpu_common import context
from test_gpu_lightglue import _pair
_, _, a, b = _pair(400, 400, 1600)
for fold in ("0", "1"):
    ctx, _, lg = context("lg", tuning={"fuse_lg_block": 1, "fold_qkv": fold}, max_batch=4)
    c = collections.Counter(hashlib.md5(ctx.lightglue_scores(a, b).tobytes()).hexdigest()[:6] for _ in range(400))
    print("fold", fold, "distinct results in 400 runs:", len(c), c.most_common(3))
