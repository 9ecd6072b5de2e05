"""In-tree build of libairfe.so (hipcc, gfx950 only).  `python -m airslam_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libairfe.so")
SOURCES = ["airfe.hip", "airfe_seq.hip", "airfe_load.hip", "airfe_detect.hip", "airfe_match.hip", "kernels_mm.hip", "kernels_conv64r.hip", "kernels_conv128r.hip", "kernels_gemm8.hip", "kernels_gemmr.hip", "kernels_img.hip", "kernels_sel.hip", "kernels_nms512.hip", "kernels_lg.hip", "kernels_attn.hip", "kernels_lgblockf.hip", "kernels_ext.hip", "kernels_s0.hip", "kernels_f32.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Per-file flags.  -fno-slp-vectorize: the SLP vectoriser packs incidental scalar fp32 arithmetic of these (latency- / HBM-bound) kernels into
# v_pk_*_f32 with cross-half `op_sel` selections — the instruction form behind round 2's irreproducible rotary element (common.h, rotate_pairs;
# tests/test_no_scratch_cpu.py::test_no_packed_f32_cross_half_selects keeps every translation unit free of it).
EXTRA_FLAGS = {f: ["-fno-slp-vectorize"] for f in ("kernels_lg.hip", "kernels_img.hip", "kernels_s0.hip", "kernels_f32.hip")}


def csrc_sha() -> str:
    """sha256 over the kernel / host sources of libairfe.so (csrc/*.hip, csrc/*.h, include/airfe*.h), in name order: the identity of the code a profile was
    taken on.  tools/pmc_summary.py and tools/pmc_traffic.py stamp it into profiles/rNN_*.json; bench.py refuses counter-derived numbers whose stamp differs from
    the tree it runs on (`roofline.counters_age`) — the GPU box has no .git to ask."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files += [os.path.join(HERE, "..", "include", h) for h in ("airfe.h", "airfe_debug.h", "airfe_seq.h")]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    headers += [os.path.join(HERE, "..", "include", h) for h in ("airfe.h", "airfe_debug.h", "airfe_seq.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, os.path.abspath(__file__)] + headers):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
