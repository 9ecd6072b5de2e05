"""HBM bytes per launch of every kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-profile
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-profile
    python tools/pmc_traffic.py gpurun_out/pmc_f/f_results.db gpurun_out/pmc_w/w_results.db profiles/rNN_hbm_traffic.json
Units as MI355X_MICROARCH.md 'HBM' prescribes: both counters are KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B,
so it is doubled; WRITE_SIZE is taken as is (it matched the algorithmic output bytes of the fused conv1b kernel to 1 %)."""
import collections
import json
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airslam_amd.build import csrc_sha  # noqa: E402


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if "counters_collection" in x][0]
    rows = cur.execute(f"select kernel_name, sum(value), count(distinct dispatch_id) from {t} where counter_name = ? group by kernel_name",
                       (counter,)).fetchall()
    return {re.sub(r"\s+", " ", k): (v, n) for k, v, n in rows}


def main(db_f, db_w, out, images_per_launch=128):
    f, w = per_kernel(db_f, "FETCH_SIZE"), per_kernel(db_w, "WRITE_SIZE")
    kernels = collections.OrderedDict()
    for k in sorted(set(f) | set(w)):
        fv, fn = f.get(k, (0.0, 0))
        wv, wn = w.get(k, (0.0, 0))
        kernels[k] = {"launches_sampled": max(fn, wn),
                      "read_bytes_per_launch": 2.0 * 1024.0 * fv / fn if fn else None,
                      "write_bytes_per_launch": 1024.0 * wv / wn if wn else None}
    # the bench's dominant stage = every conv64r_kernel launch (conv1a+conv1b+pool fused, conv2a, conv2b+pool, conv3a)
    c64 = [v for k, v in kernels.items() if "conv64r_kernel" in k]
    n = sum(v["launches_sampled"] for v in c64)
    tot = sum((v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_sampled"] for v in c64)
    dom = [v for k, v in kernels.items() if "conv64r_kernel" in k and re.search(r"conv64r_kernel<[^>]*,\s*true,\s*true", k)]   # <POOL, FUSE1A>
    nd = sum(v["launches_sampled"] for v in dom)
    totd = sum((v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches_sampled"] for v in dom)
    doc = {"csrc_sha": csrc_sha(),       # the sources these counters were measured on (bench.py: roofline.counters_age)
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over "
                     "`python bench.py --steps 2 --warmup 1 --pairs 64 --cpu-pairs 0 --no-profile`; tools/pmc_traffic.py",
           "units": "FETCH_SIZE/WRITE_SIZE are KiB; bytes = KiB*1024; FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B, "
                    "MI355X_MICROARCH.md 'HBM'); WRITE_SIZE taken as is",
           "kernels": kernels,
           "conv3x3_cin64_stage": {"hbm_bytes_per_launch": tot / n if n else None, "launches_sampled": n},
           # the bench's dominant kernel: conv64r_kernel<POOL, FUSE1A> = conv1a + conv1b + pool (its launch covers the images of one encoder chunk)
           "conv1_fused": {"hbm_bytes_per_launch": totd / nd if nd else None, "launches_sampled": nd,
                           "images_per_launch": int(images_per_launch)}}   # (bench.py --chunk of the profiled command: 128 by default)
    with open(out, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, v in kernels.items():
        if v["read_bytes_per_launch"] and v["read_bytes_per_launch"] + (v["write_bytes_per_launch"] or 0) > 20e6:
            print(f"{k[:100]:100s} read {v['read_bytes_per_launch'] / 1e6:8.1f} MB  write {(v['write_bytes_per_launch'] or 0) / 1e6:8.1f} MB  x{v['launches_sampled']}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
