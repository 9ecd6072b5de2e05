"""the feature / line records of AirSLAM's map files (include/utils.h:184-222) without Boost."""
import os
import struct
import subprocess

import numpy as np

from airslam_amd import mapfile
from planted import features


def _frame_bytes(rng, frame_id, feat, lines, junc):
    """The neighbourhood a feature record has inside Frame::serialize (include/frame.h:149-183), as a binary archive lays it out:
    frame id (int), timestamp (double), pose-fixed flag (bool, one byte), the 4x4 pose (16 doubles), THEN SerializeFeatures; some
    vectors of ints / doubles (keypoints, grids, depths ...) stand in for the members between the feature matrix and the line lists."""
    b = struct.pack("<id?", frame_id, 1.4e9 + frame_id * 0.05, bool(frame_id % 2)) + rng.normal(size=16).astype("<f8").tobytes()
    b += mapfile.pack_features(feat)
    b += struct.pack("<q", feat.shape[0]) + rng.uniform(0, 752, size=feat.shape[0] * 7).astype("<f4").tobytes()     # keypoints
    b += struct.pack("<q", 40) + rng.integers(-1, 400, size=40).astype("<i4").tobytes()
    b += mapfile.pack_lines(lines) + mapfile.pack_lines(lines[: len(lines) // 2])
    b += mapfile.pack_features(junc)
    return b


def test_record_round_trip_and_eigen_layout():
    f = features(37, 1)
    rec = mapfile.pack_features(f)
    assert struct.unpack_from("<ii", rec) == (37, 259) and len(rec) == 8 + 37 * 259 * 4
    # column-major Eigen 259 x N: element (row r, col c) sits at float index c * 259 + r
    eig = np.asfortranarray(f.T)                                     # the [259, N] matrix the reference holds
    assert rec[8:] == eig.tobytes(order="F")
    g, off = mapfile.unpack_features(rec)
    assert off == len(rec)
    np.testing.assert_array_equal(g, f)
    lines = np.arange(20, dtype=np.float64).reshape(5, 4) + 0.25
    lr = mapfile.pack_lines(lines)
    assert len(lr) == 4 + 5 * 32
    np.testing.assert_array_equal(mapfile.unpack_lines(lr)[0], lines)
    assert mapfile.unpack_features(mapfile.pack_features(np.zeros((0, 259), np.float32)))[0].shape == (0, 259)


def test_container_round_trip(tmp_path):
    frames = [features(n, 10 + n) for n in (400, 1, 123, 0)]
    p = str(tmp_path / "frames.airfemap")
    mapfile.write_records(p, frames)
    back = mapfile.read_records(p)
    assert len(back) == 4
    for a, b in zip(frames, back):
        np.testing.assert_array_equal(a, b)


def test_scanner_finds_every_record_in_an_archive_like_stream():
    rng = np.random.default_rng(0)
    blob = b"\x16\x00\x00\x00\x00\x00\x00\x00serialization::archive\x11\x00\x04\x08\x04\x08\x01\x00\x00\x00"     # a Boost header's shape
    want = []
    for i in range(6):
        feat = features(int(rng.integers(50, 400)), 100 + i)
        junc = features(int(rng.integers(1, 30)), 200 + i)
        want += [feat, junc]
        blob += _frame_bytes(rng, i, feat, rng.uniform(0, 700, size=(int(rng.integers(2, 40)), 4)), junc)
    got = mapfile.scan_boost_archive(blob)
    assert len(got) == len(want)
    for (off, a), b in zip(got, want):
        np.testing.assert_array_equal(a, b)
        assert struct.unpack_from("<ii", blob, off) == (b.shape[0], 259)


def test_loop_closure_pairs_shape():
    pairs = mapfile.loop_closure_pairs(40)
    assert len(pairs) == 200 and all(q != c for q, c in pairs)
    assert mapfile.loop_closure_pairs(3) == [(0, 1), (0, 2), (1, 2), (1, 0), (2, 0), (2, 1)]


def test_cxx_header_writes_the_same_bytes(tmp_path):
    """shim/include/airfe_mapfile.h (the reference-side codec) against the Python one, both directions."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "rt.cpp"
    src.write_text('''#include <fstream>
#include <iostream>
#include "airfe_mapfile.h"
int main(int argc, char** argv) {
  std::ifstream in(argv[1], std::ios::binary); std::ofstream out(argv[2], std::ios::binary);
  std::vector<float> rows; std::vector<double> ln; int32_t n = 0, l = 0;
  if (!airfe_mapfile::read_features(in, rows, n) || !airfe_mapfile::read_lines(in, ln, l)) return 2;
  airfe_mapfile::write_features(out, rows.data(), n); airfe_mapfile::write_lines(out, ln.data(), l);
  std::cout << n << " " << l << std::endl; return 0; }''')
    exe = tmp_path / "rt"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "shim", "include"), str(src), "-o", str(exe)], check=True)
    f = features(77, 3)
    lines = np.random.default_rng(4).uniform(0, 700, size=(9, 4))
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    a.write_bytes(mapfile.pack_features(f) + mapfile.pack_lines(lines))
    r = subprocess.run([str(exe), str(a), str(b)], check=True, capture_output=True, text=True)
    assert r.stdout.split() == ["77", "9"]
    assert a.read_bytes() == b.read_bytes()
