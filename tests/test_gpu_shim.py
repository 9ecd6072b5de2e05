"""The C++ drop-in wrappers (shim/) EXECUTED on the GPU: SuperPoint::infer, PLNet::infer, SuperPointLightGlue::infer and
SuperGlue::infer, built against the stand-in Eigen / OpenCV headers, must return byte for byte what the ctypes path returns through
the same C ABI with the same packs and configuration ("the C++ shim is compile/link-checked only and never executed")."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from airslam_amd import api, synth, weights
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_shim_wrappers_on_the_gpu(libpath, tmp_path):
    md, od = tmp_path / "models", tmp_path / "out"
    md.mkdir(); od.mkdir()
    sp, s0 = weights.synthetic_superpoint(1234), weights.synthetic_plnet_s0(1234)
    lg, sg = weights.synthetic_lightglue(1234), weights.synthetic_superglue(1234)
    # packs sit next to where the reference keeps its ONNX files, same stem (shim/include/airfe_shim_common.h::pack_path)
    weights.save_pack(str(md / "superpoint_v1_sim_int32.airfe"), sp)
    weights.save_pack(str(md / "plnet_s0.airfe"), s0)
    shutil.copy(os.path.join(GOLDEN, "plnet_s1.airfe"), str(md / "plnet_s1.airfe"))
    weights.save_pack(str(md / "superpoint_lightglue.airfe"), lg)
    weights.save_pack(str(md / "superglue_outdoor_sim_int32.airfe"), sg)
    left, right = synth.stereo_pair(480, 752, 4)
    left.tofile(str(tmp_path / "l.raw")); right.tofile(str(tmp_path / "r.raw"))
    exe = str(tmp_path / "shim_gpu")
    srcs = [os.path.join(ROOT, "shim", "src", f) for f in ("plnet.cpp", "super_point.cpp", "light_glue.cpp", "super_glue.cpp")]
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", f"-I{ROOT}/shim/stubs", f"-I{ROOT}/shim/stubs/noref", f"-I{ROOT}/shim/include", f"-I{ROOT}/include", *srcs,
                        os.path.join(ROOT, "shim", "gpu_main.cpp"), "-o", exe, f"-L{os.path.dirname(libpath)}", "-lairfe",
                        f"-Wl,-rpath,{os.path.dirname(libpath)}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(md), str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "480", "752", str(od)], capture_output=True, text=True)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stdout}\n{r.stderr}"
    print(r.stdout)

    def rd(name, dt, cols=None):
        a = np.fromfile(str(od / name), dtype=dt)
        return a.reshape(-1, cols) if cols else a

    # ---- SuperPoint::infer (the wrapper's context: max_batch 1, enc_chunk 1, config defaults)
    c = api.Context(superpoint=sp, max_batch=1, enc_chunk=1)
    f0, f1 = c.detect_points(left), c.detect_points(right)
    np.testing.assert_array_equal(rd("sp_f0.bin", np.float32, 259), f0)     # Eigen 259 x N column-major == N rows of 259
    np.testing.assert_array_equal(rd("sp_f1.bin", np.float32, 259), f1)
    c.close()
    # ---- PLNet::infer: points + on-device line branch + stage 1 + junctions, nothing supplied by the host
    c = api.Context(superpoint=s0, plnet_s1=os.path.join(GOLDEN, "plnet_s1.airfe"), max_batch=1, enc_chunk=1, line_threshold=0.5,
                    line_length_threshold=4.0)
    pf, pl, pj = c.detect_plnet(left, None, want_junctions=True)
    np.testing.assert_array_equal(rd("pl_feat.bin", np.float32, 259), pf)
    np.testing.assert_array_equal(rd("pl_junc.bin", np.float32, 259), pj)
    lines = rd("pl_lines.bin", np.float64, 4)
    np.testing.assert_array_equal(lines[0], [1.0, 2.0, 3.0, 4.0])           # what was in the vector stays in front
    np.testing.assert_array_equal(lines[1:], pl)
    c.close()
    # ---- SuperPointLightGlue::infer on NormalizeKeypoints'ed features (profile maximum 1024 keypoints, like the wrapper)
    c = api.Context(lightglue=lg, max_batch=1, max_keypoints=1024)
    pm = api.PointMatcher(c, 752, 480, 0)
    cnt, matches = pm.MatchingPoints(np.asfortranarray(f0.T), np.asfortranarray(f1.T))
    idx = rd("lg_idx.bin", np.int32, 2); sc = rd("lg_score.bin", np.float32)
    assert cnt >= 60 and [tuple(p) for p in idx] == [(m[0], m[1]) for m in matches]
    np.testing.assert_array_equal(np.float32(1.0) - sc, np.array([m[2] for m in matches], np.float32))
    c.close()
    # ---- SuperGlue::infer
    c = api.Context(superglue=sg, matcher=1, max_batch=1, max_keypoints=1024)
    n0 = api.PointMatcher.NormalizeKeypoints(np.asfortranarray(f0.T), 752, 480, 0.7)
    n1 = api.PointMatcher.NormalizeKeypoints(np.asfortranarray(f1.T), 752, 480, 0.7)
    i0, i1, m0, m1 = c.match_superglue(np.ascontiguousarray(n0.T), np.ascontiguousarray(n1.T))
    np.testing.assert_array_equal(rd("sg_i0.bin", np.int32), i0)
    np.testing.assert_array_equal(rd("sg_i1.bin", np.int32), i1)
    np.testing.assert_array_equal(rd("sg_m0.bin", np.float64), m0)
    np.testing.assert_array_equal(rd("sg_m1.bin", np.float64), m1)
    assert (i0 >= 0).sum() >= 50
    c.close()
