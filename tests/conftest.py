import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def libpath():
    """libairfe.so, built in-tree with hipcc if it is not there yet (cross-compiles without a GPU)."""
    from airslam_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.LIB_PATH


def gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


_EXPF_PROBE = {}


def host_expf_is_the_restated_glibc_routine() -> bool:
    """The bit-exact pins of match scores compare the device (which restates glibc >= 2.27's FMA build of expf, common.h expf_like_glibc) with the HOST's libm —
    through oracle/ref_post._expf and through the compiled reference.  On a host whose libm runs another expf (no FMA build: older x86, aarch64, musl, an older
    glibc) those pins would fail although the device is unchanged: this probe — tools/expf_glibc_check.c on every 4099th negative float, a fraction
    of a second — lets them skip there with the reason.  No gcc: assume the image's libm (the probe is the CPU suite's test_glibc_expf_restatement... in full)."""
    if "ok" not in _EXPF_PROBE:
        import shutil
        import subprocess
        import tempfile
        gcc = shutil.which("gcc")
        ok = True
        if gcc is not None:
            with tempfile.TemporaryDirectory() as td:
                exe = os.path.join(td, "expf_check")
                try:
                    subprocess.run([gcc, "-O2", "-mfma", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools", "expf_glibc_check.c"), "-lm"], check=True,
                                   capture_output=True)
                    out = subprocess.run([exe, "0", "4099"], check=True, capture_output=True, text=True).stdout
                    ok = "mismatches=0" in out
                except Exception:
                    ok = True
        _EXPF_PROBE["ok"] = ok
    return _EXPF_PROBE["ok"]


def skip_unless_host_expf_is_glibc():
    if not host_expf_is_the_restated_glibc_routine():
        pytest.skip("the host's libm expf is not the routine the device restates (glibc >= 2.27, FMA build): the bit-exact score pins compare against this host's libm")
