"""No-GPU checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol
include/airfe.h declares; the product path fails loudly (no CPU fallback) when no device is visible."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, gpu_available


def _header_symbols(name=None):
    """every function include/*.h declares (the boundary airfe.h + the test hooks airfe_debug.h)"""
    out = set()
    for h in ([name] if name else sorted(os.listdir(os.path.join(ROOT, "include")))):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(airfe_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_debug_hooks_are_not_in_the_product_header():
    assert not [n for n in _header_symbols("airfe.h") + _header_symbols("airfe_seq.h") if n.startswith("airfe_debug_")]
    assert all(n.startswith("airfe_seq_") for n in _header_symbols("airfe_seq.h"))
    assert all(n.startswith("airfe_debug_") for n in _header_symbols("airfe_debug.h"))


def test_library_exports_every_declared_symbol(libpath):
    from airslam_amd import _lib
    lib = C.CDLL(libpath)
    names = _header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libairfe.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and include/airfe.h disagree"


def test_default_cfg_matches_reference_yaml(libpath):
    from airslam_amd import _lib
    c = _lib.Cfg()
    _lib.lib().airfe_default_cfg(C.byref(c))
    # configs/visual_odometry/vo_euroc.yaml:1-14
    assert (c.max_keypoints, c.remove_borders, c.matcher, c.image_width, c.image_height) == (400, 4, 0, 752, 480)
    assert abs(c.keypoint_threshold - 0.004) < 1e-9 and abs(c.line_threshold - 0.75) < 1e-9
    assert c.line_length_threshold == 50.0


@pytest.mark.skipif(gpu_available(), reason="only meaningful without a GPU")
def test_create_fails_loudly_without_gpu(libpath):
    from airslam_amd import api
    with pytest.raises(api.AirfeError) as e:
        api.Context()
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "airslam_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f"{f} reaches into oracle/"


def test_weight_pack_roundtrip(tmp_path):
    from airslam_amd import weights
    w = weights.synthetic_superpoint(7)
    weights.check_spec(w, weights.superpoint_spec())
    p = str(tmp_path / "sp.airfe")
    weights.save_pack(p, w)
    r = weights.load_pack(p)
    assert list(r) == list(w)
    for k in w:
        np.testing.assert_array_equal(r[k], w[k])
    assert sum(v.size for v in w.values()) == 1300865
    lg = weights.synthetic_lightglue(7)
    weights.check_spec(lg, weights.lightglue_spec())


def test_library_reads_no_environment():
    """kernel selection of the shipped library must not depend on the process environment — the switches live in airfe_tuning."""
    csrc = os.path.join(ROOT, "airslam_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f} reads the environment"


def test_every_c_entry_catches_exceptions():
    """include/airfe.h and include/airfe_seq.h promise "never throws": every extern "C" function that can reach host-side C++ (std::vector / std::string growth) is
    a function-try-block ending in AIRFE_CATCH (csrc/airfe.hip) / SEQ_CATCH or a catch-all (csrc/airfe_seq.hip)."""
    trivial = {"airfe_profile_stages", "airfe_has_line_branch", "airfe_debug_trace_slots"}      # one expression on plain ints / pointers
    found = set()
    for fname, catch, least in (("airfe.hip", "} AIRFE_CATCH(", 45), ("airfe_seq.hip", "} SEQ_CATCH(", 5)):
        src = open(os.path.join(ROOT, "airslam_amd", "csrc", fname)).read()
        body = src[src.index('extern "C" {'):]
        defs = re.findall(r"^int (airfe_[a-z0-9_]+)\([^;{]*?\)\s*(try)?\s*\{", body, flags=re.M | re.S)
        assert len(defs) >= least
        missing = [n for n, t in defs if not t and n not in trivial]
        assert not missing, f"{fname}: no function-try-block: {missing}"
        n_catch = body.count(catch) + body.count("} catch (...) { return -1; }") + (body.count("} AIRFE_CATCH(") if fname != "airfe.hip" else 0)
        assert n_catch == sum(1 for _, t in defs if t), fname
        found |= {n for n, _ in defs}
    not_int = ("airfe_default_cfg", "airfe_default_tuning", "airfe_destroy", "airfe_last_error", "airfe_profile_stage_name", "airfe_seq_default_policy", "airfe_seq_destroy",
               "airfe_seq_last_error", "airfe_seq_stream")
    declared = {n for n in _header_symbols() if n not in not_int}
    assert declared == found, declared ^ found


def test_default_tuning_is_all_minus_one(libpath):
    from airslam_amd import _lib
    t = _lib.Tuning()
    _lib.lib().airfe_default_tuning(C.byref(t))
    assert all(getattr(t, n) == -1 for n, _ in _lib.Tuning._fields_ if n != "reserved") and list(t.reserved) == [-1] * 5
    assert C.sizeof(_lib.Tuning) == 4 * 23


def test_committed_counter_profiles_were_taken_on_these_kernel_sources():
    """`roofline.traffic` / `mfma_util_counters` on the bench line come from committed rocprofv3 PMC passes — they must be passes over THIS tree's
    kernels (sha256 over csrc/*.hip, csrc/*.h, include/airfe*.h, stamped by tools/pmc_summary.py / tools/pmc_traffic.py).  bench.py drops stale ones (null +
    counters_age.stale); this test keeps a commit from shipping them at all: change a kernel -> re-run tools/gpu_profile.sh."""
    import glob
    import json
    from airslam_amd.build import csrc_sha
    for suffix in ("pmc_summary.json", "hbm_traffic.json"):
        newest = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)) if re.match(r"r\d\d_" + re.escape(suffix) + "$", os.path.basename(f)))[-1]
        assert json.load(open(newest)).get("csrc_sha") == csrc_sha(), f"{os.path.basename(newest)} was measured on other kernel sources than this tree's"
