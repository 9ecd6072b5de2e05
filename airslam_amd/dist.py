"""Multi-GPU harness: stereo frames shard embarrassingly across ranks (one process per GPU); the only exchange
on the path is the variable-length gather of matches to rank 0 — fixed-capacity padded buffers
plus the per-pair counts, one collective per step.  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests.  The reference has no multi-GPU code at all (README.md:143), so this is new design, not a port."""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """-> (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("AIRFE_DIST_FORCE_INIT")) and not dist.is_initialized():      # (FORCE_INIT: a process group of one rank, tests only)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":      # the rank's own GPU must be current BEFORE the communicator exists (else every rank's barrier lands on device 0)
            dev = local % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n_items for `rank` (remainder spread over the first ranks)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_matches(idx: torch.Tensor, score: torch.Tensor, nmatch: torch.Tensor, dst: int = 0, force: bool = False):
    """idx [B,cap,2] int32, score [B,cap] f32, nmatch [B] int32 on every rank -> on `dst`:
    (idx [W*B,cap,2], score [W*B,cap], nmatch [W*B]) in rank order; None elsewhere.
    One packed buffer per rank => a single gather collective per step (latency-bound payload).
    force: go through the collective even in a group of ONE rank (tests/test_gpu_rccl.py: the only way to put RCCL under this code on a 1-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return idx, score, nmatch
    b, cap, _ = idx.shape
    packed = torch.cat([idx.reshape(b, cap * 2).to(torch.int32), score.view(torch.int32).reshape(b, cap),
                        nmatch.reshape(b, 1).to(torch.int32)], dim=1).contiguous()
    world, rank = dist.get_world_size(), dist.get_rank()
    outs: Optional[List[torch.Tensor]] = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, outs, dst=dst)
    if rank != dst:
        return None
    allp = torch.cat(outs, dim=0)
    gi = allp[:, :cap * 2].reshape(world * b, cap, 2)
    gs = allp[:, cap * 2:cap * 3].contiguous().view(torch.float32)
    gn = allp[:, cap * 3].contiguous()
    return gi, gs, gn


def max_over_ranks(x: float, device, force: bool = False) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_over_ranks(x: float, device, force: bool = False) -> List[float]:
    """every rank's value of `x`, in rank order, on every rank (one all_gather of a double)"""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return [float(x)]
    t = torch.tensor([x], dtype=torch.float64, device=device)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def pin_rank_to_cores(local: int, n_local: int) -> Optional[List[int]]:
    """One process per GPU queues hundreds of launches per step from its own host thread: give local rank `local` of `n_local` its own contiguous share of the
    cores this process may run on (os.sched_setaffinity), so that eight ranks do not migrate over each other's cores.  -> the cores chosen, or None where the
    platform has no affinity call / there are fewer cores than ranks (nothing is changed then)."""
    if not hasattr(os, "sched_getaffinity") or n_local <= 1:
        return None
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < n_local:
        return None
    per = len(cores) // n_local
    mine = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
