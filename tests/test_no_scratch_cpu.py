"""Build-time guard: the hot kernels must not touch scratch memory.  (gemm8_kernel is deliberately absent: it spills ~20 loop-invariant
epilogue pointers once per workgroup in its prologue and reloads each once — outside the K loop, not a cost.)  A register array that hipcc cannot keep in registers (a pointer
select between two accumulator arrays, a spill at the occupancy bound) silently moves to scratch and the kernel runs 3-8x slower with
every parity test still green — it happened twice in round 2 (attention32_kernel: 46 -> 123 us with 45 spilled registers; later a
masked-tail if/else put both score accumulators into 192 B of scratch per lane: 41 -> 330 us)."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "airslam_amd", "csrc")
# file -> kernel-name fragments that must compile without scratch
HOT = {
    "kernels_attn.hip": ["attention32_kernel"],
    "kernels_lgblockf.hip": ["lg_blockf_kernel"],
    "kernels_conv64r.hip": ["conv64r_kernel"],
    "kernels_conv128r.hip": ["conv128r_kernel"],
    "kernels_gemmr.hip": ["gemmr_kernel", "gemmr_pair_kernel"],
    "kernels_ext.hip": ["sg_sinkhorn_fused_kernel", "sg_sinkhorn_reg_kernel", "plnet_s1_kernel", "s1_junc_proj_kernel"],
    "kernels_s0.hip": ["s0_j2l_grid_kernel", "s0_decode_kernel"],
    "kernels_nms512.hip": ["nms512_kernel"],
}


def _usage(src):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", os.devnull],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
        for key in ("ScratchSize [bytes/lane]", "VGPRs Spill", "VGPRs"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                out[name][key] = int(m.group(1))
    return out


def test_hot_kernels_use_no_scratch():
    with ThreadPoolExecutor(max_workers=min(len(HOT), os.cpu_count() or 1)) as ex:
        usages = dict(zip(HOT, ex.map(_usage, HOT)))
    seen = 0
    for src, frags in HOT.items():
        for name, u in usages[src].items():
            if any(f in name for f in frags):
                seen += 1
                assert u.get("ScratchSize [bytes/lane]", 0) == 0 and u.get("VGPRs Spill", 0) == 0, (src, name, u)
    assert seen >= 10          # the template instantiations were actually found
