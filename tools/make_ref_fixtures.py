"""Writes tests/golden/ref_pin.npz: the outputs of the REFERENCE'S OWN CODE (oracle/_ref/libairslam_ref.so, compiled from /root/reference by
oracle/Makefile) on the seeded cases of tests/ref_cases.py.  The inputs are regenerated from their seeds, so the file holds outputs only.

    make -C oracle && python tools/make_ref_fixtures.py

tests/test_ref_pin_cpu.py holds oracle/ref_post.py to these fixtures (always) and to the live library (when it is built), and checks that this
script regenerates the committed file bit for bit; tests/test_gpu_ref_pin.py holds the HIP kernels to them."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_cases as rc  # noqa: E402
from oracle import ref_lib  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "ref_pin.npz")


def generate() -> dict:
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for fam, (table, gen) in rc.FAMILIES.items():
            for name in table:
                for k, v in rc.run_ref(fam, gen(name), tmp).items():
                    if k == "fed_input":           # the pre-processed image the engine was fed: 1 MB each, and cv::resize is a stand-in anyway
                        continue
                    out[f"{fam}/{name}/{k}"] = np.asarray(v)
    out["_sources"] = np.array(ref_lib.sources())
    return out


if __name__ == "__main__":
    assert ref_lib.available(), "build oracle/_ref first: make -C oracle (needs /root/reference)"
    d = generate()
    np.savez_compressed(OUT, **d)
    print(f"wrote {OUT}: {len(d)} arrays, {os.path.getsize(OUT) / 1e6:.2f} MB")
