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
    "kernels_lgblockf.hip": ["lg_blockf_kernel", "lg_blockf_mixed_kernel"],
    "kernels_conv64r.hip": ["conv64r_kernel"],
    "kernels_conv128r.hip": ["conv128r_kernel"],
    "kernels_gemmr.hip": ["gemmr_kernel", "gemmr_pair_kernel", "gemmr_gather_kernel"],
    "kernels_ext.hip": ["sg_sinkhorn_reg_kernel", "plnet_s1_kernel", "s1_junc_proj_kernel", "s1h_junc_proj_kernel"],
    "kernels_s0.hip": ["s0_j2l_grid_kernel", "s0_decode_kernel"],
    "kernels_nms512.hip": ["nms512_kernel"],
    "kernels_lg.hip": ["lg_sim_lse_kernel", "lg_sim_arg_kernel"],
}


# every translation unit of the library: also scanned for packed fp32 math with cross-half selections (see test_no_packed_f32_cross_half_selects)
from airslam_amd import build as _build
SCANNED = sorted(_build.SOURCES)
_CACHE = {}


def _compile(src):
    """(resource usage per kernel, ISA text) of one translation unit; compiled once per session"""
    if src not in _CACHE:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc"] + _build.FLAGS + _build.EXTRA_FLAGS.get(src, []) +        # the build's own flags
                               ["-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-o", out],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            _CACHE[src] = (r.stderr, open(out).read())
    return _CACHE[src]


def _compile_all():
    with ThreadPoolExecutor(max_workers=min(len(SCANNED), os.cpu_count() or 1)) as ex:
        list(ex.map(_compile, SCANNED))


def _usage(src):
    class R: pass
    r = R()
    r.stderr = _compile(src)[0]
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
    _compile_all()
    usages = {src: _usage(src) for src in HOT}
    seen = 0
    for src, frags in HOT.items():
        for name, u in usages[src].items():
            if any(f in name for f in frags):
                seen += 1
                assert u.get("ScratchSize [bytes/lane]", 0) == 0 and u.get("VGPRs Spill", 0) == 0, (src, name, u)
    assert seen >= 10          # the template instantiations were actually found


def test_split_stage1_kernel_spills_only_outside_its_matrix_loops():
    """plnet_s1h_kernel is capped at 128 registers (four workgroups per CU) and keeps a handful of tile-loop-INVARIANT values in scratch — stored once in the prologue,
    reloaded between phases.  That is by design (kernels_ext.hip); what must not happen is a scratch access inside a layer's MFMA run."""
    _compile_all()
    u = [v for k, v in _usage("kernels_ext.hip").items() if "plnet_s1h_kernel" in k]
    assert len(u) == 1 and u[0].get("ScratchSize [bytes/lane]", 0) <= 64, u
    isa = _compile("kernels_ext.hip")[1]
    body = isa[isa.index("plnet_s1h_kernel"):]
    body = body[:body.index("s_endpgm")].splitlines()
    mf = [i for i, l in enumerate(body) if "v_mfma_f32_32x32x16_f16" in l]
    sc = [i for i, l in enumerate(body) if re.search(r"\bscratch_(load|store)", l)]
    assert len(mf) == 138 and len(sc) <= 16, (len(mf), len(sc))
    # the four layers are runs of MFMAs separated by barriers: no scratch access between the first and last MFMA of a run
    barriers = [i for i, l in enumerate(body) if "s_barrier" in l]
    for a, b in zip([0] + barriers, barriers + [len(body)]):
        run = [i for i in mf if a <= i < b]
        if run:
            assert not [i for i in sc if run[0] <= i <= run[-1]], ("scratch access inside an MFMA run", a, b)


def test_no_packed_f32_cross_half_selects():
    """No v_pk_{mul,fma,add}_f32 whose LOW result selects the HIGH half of a source (a non-zero `op_sel`).  hipcc's vectoriser made
    exactly that of the rotary epilogue (pair 1 of a lane's four (even, odd) pairs has its cos / sin in the high half of a register pair):
    `v_pk_mul_f32 .. op_sel:[1,1] op_sel_hi:[0,1]` + `v_pk_fma_f32 .. op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]` — and that one element
    (lanes 48-63, the even element of pair 1) came out wrong in one 16-token tile of a launch in 0.1-4 % of the 64-pair steps: round 2's
    "matcher race" (tools/experiments/matcher_trace.py, profiles/r03_matcher_trace_probe1.txt, _probe2.txt).  The broadcast forms (`op_sel_hi` only), thousands
    in the GELU / LayerNorm code, never deviated in ~10 000 traced steps.  rotate_pairs() is written in single instructions since; this
    keeps the pattern from coming back through the vectoriser anywhere else."""
    _compile_all()
    pat = re.compile(r"v_pk_(?:mul|fma|add)_f32 .*\bop_sel:\[[0-9,]*1")
    hits = []
    n_pk = 0
    for src in SCANNED:
        for line in _compile(src)[1].splitlines():
            if "v_pk_" in line and "_f32" in line:
                n_pk += 1
                if pat.search(line):
                    hits.append((src, line.strip()))
    assert n_pk > 1000          # the scan saw the packed code there is (GELU, LayerNorm, bias adds)
    assert not hits, hits[:8]
