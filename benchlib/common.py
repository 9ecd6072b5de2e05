"""What every bench.py workload shares: peaks, the contract's line builder, barriers, the stage table and the counter-profile rule."""
import glob
import json
import os
import re
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# which roof a stage is priced against: matrix stages by their algorithmic FLOPs against the dense 2-byte MFMA peak, the rest by
# their algorithmic bytes against the HBM peak; plnet_stage1 is a chain of gathers and a small MLP on the 2-byte MFMA (cfg.line_precision = 3): latency-bound, no single roof
STAGE_BOUND = {"conv1_fused": "mfma", "conv3x3_cin64": "mfma", "conv3x3_cin128": "mfma", "head_gemm": "mfma", "lg_gemm": "mfma", "lg_attention": "mfma"}
DOMINANT_STAGE = "conv1_fused"   # the dominant KERNEL (conv64r_kernel<POOL, FUSE1A>: conv1a + conv1b + pool, ~21 % of a step) is a stage of
                                # its own: its launches keep their HIP events inside the timed region
PEAK_MFMA_TFLOPS = 2500.0       # dense bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0
S1_PACK = os.path.join(ROOT, "tests", "golden", "plnet_s1.airfe")      # the REAL stage-1 weights (output/plnet_s1.onnx of the reference)


def latest_profile(suffix):
    """newest committed profiles/rNN_<suffix> (counter passes are separate runs: tools/gpu_profile.sh), or None"""
    c = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)) if re.match(r"r\d\d_" + re.escape(suffix) + "$", os.path.basename(f))]
    return sorted(c)[-1] if c else None


def counter_profile(suffix):
    """(document, path, age) of the newest committed profiles/rNN_<suffix>: age = {"profile", "profile_csrc_sha", "tree_csrc_sha", "stale"} — counters measured on other
    kernel sources than the ones this process runs are NOT reported (document = None): a number from an older kernel must never ride on a changed one."""
    from airslam_amd.build import csrc_sha
    pf = latest_profile(suffix)
    if not pf:
        return None, None, None
    with open(pf) as fh:
        doc = json.load(fh)
    now = csrc_sha()
    age = {"profile": "profiles/" + os.path.basename(pf), "profile_csrc_sha": doc.get("csrc_sha"), "tree_csrc_sha": now, "stale": doc.get("csrc_sha") != now}
    return (None if age["stale"] else doc), pf, age


def encoder_mfma_util(pdoc, pf):
    """counter-derived MFMA utilisation of the encoder kernels (separate --pmc pass, tools/pmc_summary.py) -> (dict or None, source text or None)"""
    if not pdoc:
        return None, None
    enc = {k: v for k, v in pdoc["kernels"].items() if ("conv64r_kernel" in k or "conv128r_kernel" in k) and "mfma_util" in v}
    wsum = sum(v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"] for v in enc.values())
    if wsum <= 0:
        return None, None
    util = {"encoder_time_weighted": sum(v["mfma_util"] * v["counters_per_launch"]["GRBM_GUI_ACTIVE"] * v["launches_sampled"] for v in enc.values()) / wsum,
            "per_kernel": {k.split("(")[0].replace("void airfe::", ""): round(v["mfma_util"], 3) for k, v in enc.items()}}
    return util, "profiles/" + os.path.basename(pf) + ": SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)"


def barrier(dev, world):
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)


def dtype_label(args):
    return args.dtype if args.dtype == args.matcher_dtype else f"{args.dtype} (encoder) + {args.matcher_dtype} (matcher), fp32 accumulate"


def line(args, *, metric, value, unit, world, steps, warmup, ms_per_step, config, **extra):
    """The driver's contract, in one place: every workload's JSON line starts from this dict (extra keys are appended in the order given)."""
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_label(args), "data": "synthetic", "config": config}
    out.update(extra)
    for k in ("roofline", "cpu_baseline", "collective"):
        out.setdefault(k, None)
    return out


def stage_table(stages, n_steps):
    """ctx.profile_read() of n_steps bracketed steps -> ({stage: {ms_per_step, share, tflops, algo_gbs, bound, frac}}, algorithmic FLOPs per step):
    each stage against its own roof (algorithmic FLOPs or bytes of the stage / its event time / the peak)."""
    tot = sum(s["ms"] for s in stages.values())
    tab = {k: {"ms_per_step": v["ms"] / n_steps, "share": v["ms"] / tot if tot else 0,
               "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None,
               "algo_gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None}
           for k, v in stages.items() if v["launches"]}
    for k, v in tab.items():
        if STAGE_BOUND.get(k) == "mfma" and v["tflops"]:
            v["bound"], v["frac"] = "mfma", v["tflops"] / PEAK_MFMA_TFLOPS
        elif k == "plnet_stage1":
            v["bound"], v["frac"] = "latency (gather chains + a 4-layer MLP per 32-line tile: DESIGN.md 3)", None
        elif v["algo_gbs"]:
            v["bound"], v["frac"] = "hbm", v["algo_gbs"] / PEAK_HBM_GBS
    return tab, sum(v["flops"] for v in stages.values()) / n_steps


def merge_stages(dicts):
    """sum the profile_read() dicts of several contexts (the sequence workload runs two or four)"""
    out = {}
    for d in dicts:
        for k, v in d.items():
            o = out.setdefault(k, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "launches": 0})
            for f in o:
                o[f] += v[f]
    return out


class HostClock:
    """wall time the host spends inside a block (queueing device work), summed: `with clock: step()`"""

    def __init__(self):
        self.s, self.cpu, self.n = 0.0, 0.0, 0

    def __enter__(self):
        self._t, self._c = time.perf_counter(), time.thread_time()
        return self

    def __exit__(self, *a):
        self.s += time.perf_counter() - self._t
        self.cpu += time.thread_time() - self._c
        self.n += 1
        return False

    def ms(self):
        return self.s / max(self.n, 1) * 1e3

    def cpu_ms(self):
        """CPU time of the calling thread inside the block (a launch call that blocks on a full queue sleeps: wall time, not CPU time)"""
        return self.cpu / max(self.n, 1) * 1e3
