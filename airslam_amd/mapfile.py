"""AirSLAM's on-disk feature / line records without Boost.

The reference saves its map as a Boost *binary* archive of the whole object graph (`src/map_builder.cc:564`, read back at
`src/map_refiner.cc:42`, `src/map_user.cc:88`).  Inside it every frame's feature matrix goes through `SerializeFeatures`
(`include/utils.h:205-222`): `int cols; int rows;` followed by `make_array(features.data(), features.size())` — and a binary
archive writes primitives and arrays as their raw little-endian bytes, so on disk one record is

    int32 cols | int32 rows (= 259) | rows*cols float32, COLUMN-major (= cols rows of [score, x, y, d0..d255])

which is byte-identical to the `[N][259]` float rows of `include/airfe.h`.  Line lists (`SerializeEigenVector4dList`,
`include/utils.h:184-202`) are `int32 l | l * 4 float64`.  This module packs / unpacks those records, keeps a small container
of them (for the matcher-only loop-closure benchmark, `src/map_refiner.cc:213-230`: a query frame against its best <= 5
candidates), and can pull the feature records out of a real AirSLAM map file by scanning for their header pattern — the
rest of the archive (tracked pointers, class versions, the other members of `Frame::serialize`, `include/frame.h:149-183`)
is skipped, not interpreted.  `shim/include/airfe_mapfile.h` is the same codec for the reference's C++ side."""
from __future__ import annotations

import struct
from typing import Iterable, List, Tuple

import numpy as np

ROWS = 259
MAGIC = b"AIRFEMAP1\0"


def pack_features(feat_rows: np.ndarray) -> bytes:
    """[N, 259] float32 rows (= the Eigen 259 x N matrix, column-major) -> one SerializeFeatures record."""
    f = np.ascontiguousarray(feat_rows, dtype="<f4").reshape(-1, ROWS)
    return struct.pack("<ii", f.shape[0], ROWS) + f.tobytes()


def unpack_features(buf: bytes, off: int = 0) -> Tuple[np.ndarray, int]:
    cols, rows = struct.unpack_from("<ii", buf, off)
    if rows != ROWS or cols < 0:
        raise ValueError(f"not a feature record at byte {off}: cols={cols} rows={rows}")
    n = cols * rows
    a = np.frombuffer(buf, dtype="<f4", count=n, offset=off + 8).reshape(cols, rows).copy()
    return a, off + 8 + 4 * n


def pack_lines(lines: np.ndarray) -> bytes:
    """[L, 4] float64 (x1, y1, x2, y2) -> one SerializeEigenVector4dList record."""
    l = np.ascontiguousarray(lines, dtype="<f8").reshape(-1, 4)
    return struct.pack("<i", l.shape[0]) + l.tobytes()


def unpack_lines(buf: bytes, off: int = 0) -> Tuple[np.ndarray, int]:
    (l,) = struct.unpack_from("<i", buf, off)
    if l < 0:
        raise ValueError("negative line count")
    a = np.frombuffer(buf, dtype="<f8", count=4 * l, offset=off + 4).reshape(l, 4).copy()
    return a, off + 4 + 32 * l


def write_records(path: str, frames: Iterable[np.ndarray]) -> None:
    """A flat container of feature records (what the benchmark replays): magic, int32 count, then the records back to back."""
    frames = list(frames)
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<i", len(frames)))
        for fr in frames:
            f.write(pack_features(fr))


def read_records(path: str) -> List[np.ndarray]:
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:len(MAGIC)] != MAGIC:
        raise ValueError(f"{path}: not an airfe feature container")
    (n,) = struct.unpack_from("<i", buf, len(MAGIC))
    off, out = len(MAGIC) + 4, []
    for _ in range(n):
        a, off = unpack_features(buf, off)
        out.append(a)
    return out


def scan_boost_archive(buf: bytes, max_cols: int = 4096) -> List[Tuple[int, np.ndarray]]:
    """Feature records inside a Boost binary archive written by the reference (map_builder.cc:564): every position where
    `int32 cols in [1, max_cols]` is followed by `int32 259` and by cols * 259 finite floats whose scores (element 0 of each
    column) lie in (0, 1] and whose descriptors (elements 3..258) have unit norm — the invariants extract_descriptors
    (src/plnet.cpp:369-417) guarantees.  Returns (byte offset, [cols, 259] array) in file order; junction matrices
    (`_junctions`, same record type) are found the same way."""
    out: List[Tuple[int, np.ndarray]] = []
    pat = struct.pack("<i", ROWS)
    n, pos = len(buf), 4
    while True:
        pos = buf.find(pat, pos)
        if pos < 0 or pos + 4 > n:
            break
        (cols,) = struct.unpack_from("<i", buf, pos - 4)
        end = pos + 4 + 4 * cols * ROWS
        if 1 <= cols <= max_cols and end <= n:
            a = np.frombuffer(buf, dtype="<f4", count=cols * ROWS, offset=pos + 4).reshape(cols, ROWS)
            if np.isfinite(a).all() and (a[:, 0] > 0).all() and (a[:, 0] <= 1).all() and \
                    np.allclose(np.linalg.norm(a[:, 3:], axis=1), 1.0, atol=1e-3):
                out.append((pos - 4, a.copy()))
                pos = end
                continue
        pos += 1
    return out


def loop_closure_pairs(n_frames: int, n_candidates: int = 5, stride: int = 7) -> List[Tuple[int, int]]:
    """The matcher workload of loop closure (map_refiner.cc:213-230): every query frame against its GoodCandidateNum <= 5
    best group candidates.  Candidates here are a fixed pseudo-random choice of other frames (the BoW ranking is not on this path)."""
    pairs = []
    for q in range(n_frames):
        for k in range(min(n_candidates, n_frames - 1)):
            c = (q + 1 + k * stride) % n_frames
            if c == q:
                c = (c + 1) % n_frames
            pairs.append((q, c))
    return pairs
