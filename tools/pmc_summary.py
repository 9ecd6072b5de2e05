#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters + the MFMA utilisation they imply.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES ... GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_a -o a -- <cmd>
    python tools/pmc_summary.py out.json gpurun_out/pmc_a/a_results.db [more.db ...]

Every counter is summed over a kernel's dispatches and divided by the dispatch count.  Derived columns (MI355X_MICROARCH.md,
'Per-instruction cycle constants': SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles of matrix-pipe occupancy summed over all SIMDs;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves):
    mfma_util      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE_per_xcd * 1024 SIMDs)
    wave split     = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY as fractions of SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; it is divided by 8 here (checked against kernel durations x clock)."""
import collections
import json
import re
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from airslam_amd.build import csrc_sha  # noqa: E402

SIMDS = 1024
XCDS = 8


def per_kernel(db):
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if "counters_collection" in x][0]
    rows = cur.execute(f"select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from {t} group by kernel_name, counter_name").fetchall()
    out = collections.defaultdict(dict)
    for k, cn, v, n in rows:
        out[re.sub(r"\s+", " ", k)][cn] = (v, n)
    dur = {}
    try:
        kt = [x for x in tabs if x.startswith("kernels") or "kernel_dispatch" in x]
        for cand in kt:
            cols = [r[1] for r in cur.execute(f"pragma table_info({cand})")]
            if "start" in cols and "end" in cols and ("name" in cols or "kernel_name" in cols):
                nm = "name" if "name" in cols else "kernel_name"
                for k, s, n in cur.execute(f"select {nm}, sum(end - start), count(*) from {cand} group by {nm}"):
                    dur[re.sub(r"\s+", " ", k)] = (s / n / 1e3, n)
                break
    except Exception:
        pass
    return out, dur


def main(out_path, *dbs):
    kernels = collections.OrderedDict()
    for db in dbs:
        pk, dur = per_kernel(db)
        for k, cs in pk.items():
            e = kernels.setdefault(k, {"counters_per_launch": {}})
            for cn, (v, n) in cs.items():
                e["counters_per_launch"][cn] = v / n
                e["launches_sampled"] = max(e.get("launches_sampled", 0), n)
            if k in dur:
                e["avg_us_under_pmc"] = dur[k][0]
    for k, e in kernels.items():
        c = e["counters_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            e["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / XCDS * SIMDS)
        if c.get("SQ_WAVE_CYCLES"):
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if nm in c:
                    e["frac_" + nm] = c[nm] / c["SQ_WAVE_CYCLES"]
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_INSTS_VALU"):
            e["valu_per_mfma"] = (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"]
    doc = {"csrc_sha": csrc_sha(),       # the sources these counters were measured on (bench.py: roofline.counters_age)
           "source": "rocprofv3 --pmc ... --kernel-trace (PMC passes only, never with other tracing); tools/pmc_summary.py",
           "derived": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); frac_* = share of SQ_WAVE_CYCLES",
           "kernels": kernels}
    with open(out_path, "w") as fh:
        json.dump(doc, fh, indent=1)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1]["counters_per_launch"].get("GRBM_GUI_ACTIVE", 0)):
        c = e["counters_per_launch"]
        if c.get("GRBM_GUI_ACTIVE", 0) < 2e4 * XCDS:
            continue
        print(f"{k[:70]:70s} x{e.get('launches_sampled', 0):3d} gui/xcd {c.get('GRBM_GUI_ACTIVE', 0) / XCDS:9.0f} mfma_util {e.get('mfma_util', float('nan')):.3f} "
              f"wait {e.get('frac_SQ_WAIT_ANY', float('nan')):.2f} stall {e.get('frac_SQ_WAIT_INST_ANY', float('nan')):.2f} issue {e.get('frac_SQ_ACTIVE_INST_ANY', float('nan')):.2f} "
              f"valu/mfma {e.get('valu_per_mfma', float('nan')):.1f}")


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
