"""world_size-2 gloo test of the only exchange on the path: the match gather to rank 0."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _payload(rank, b=3, cap=400):
    """What a rank holds after a step: the oracle's match lists of `b` planted pairs, in the device's [b][cap] buffer layout
    (~200 matches each) — the real size and shape of the gather, not a toy."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from airslam_amd import weights
    from oracle import ref_nets, ref_post
    from planted import normalised, planted_pair
    lg = weights.synthetic_lightglue(1234, n_layers=2)
    idx = np.zeros((b, cap, 2), np.int32); sc = np.zeros((b, cap), np.float32); nm = np.zeros((b,), np.int32)
    for i in range(b):
        f0, f1 = planted_pair(400 - 17 * i, 400 - 9 * rank, 1000 * rank + i)
        a, c = normalised(f0)[:, 1:], normalised(f1)[:, 1:]
        m, s = ref_post.filter_matches(ref_nets.lightglue_forward(lg, a[:, :2], a[:, 2:], c[:, :2], c[:, 2:], n_layers=2), 0.1)
        nm[i] = len(m); idx[i, :len(m)] = m; sc[i, :len(m)] = s
    return idx, sc, nm


def _worker(rank, world, port, q, payload):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from airslam_amd import dist as adist
    r, w, _ = adist.init_from_env("gloo")
    idx, sc, nm = (torch.from_numpy(x) for x in payload)
    out = adist.gather_matches(idx, sc, nm, dst=0)
    lo, hi = adist.shard_range(10, r, w)
    mx = adist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    if r == 0:
        gi, gs, gn = out
        q.put((gi.numpy(), gs.numpy(), gn.numpy(), (lo, hi), mx))
    else:
        assert out is None
        q.put(((lo, hi), mx))
    torch.distributed.destroy_process_group()


def test_gather_matches_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    payloads = [_payload(r) for r in range(2)]
    assert min(int(p[2].min()) for p in payloads) >= 100, "the gather must carry real match lists"
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q, payloads[r])) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = [r for r in res if len(r) == 5][0]
    other = [r for r in res if len(r) == 2][0]
    gi, gs, gn, rng0, mx = full
    assert gi.shape == (6, 400, 2) and gs.shape == (6, 400) and gn.shape == (6,)
    for rank in range(2):
        idx, sc, nm = payloads[rank]
        np.testing.assert_array_equal(gn[rank * 3:(rank + 1) * 3], nm)
        np.testing.assert_array_equal(gi[rank * 3:(rank + 1) * 3], idx)
        np.testing.assert_array_equal(gs[rank * 3:(rank + 1) * 3], sc)
    assert rng0 == (0, 5) and other[0] == (5, 10) and mx == 2.0 and other[1] == 2.0


def test_shard_range_covers_everything():
    from airslam_amd import dist as adist
    for n in (1, 7, 8, 64, 200):
        for w in (1, 2, 4, 8):
            spans = [adist.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_bench_gpus_n_starts_n_ranks_and_checks_the_world_size():
    """bench.py --gpus N: (a) with a launcher that started a different number of ranks it refuses instead of printing a mislabelled line;
    (b) with no launcher it starts N ranks itself (torch.distributed.run) — without a GPU each of them then stops at the product's
    "needs a GPU" exit, which is enough to see that two ranks came up (the GPU form of this test: tests/test_gpu_bench_contract.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 1 rank(s)" in r.stderr
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, AIRFE_DIST_BACKEND="gloo"), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") >= 2, r.stderr[-1500:]
