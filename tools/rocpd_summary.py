"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel table rocprofv3 --stats prints.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db profiles/r01_x_kernel_stats.csv"""
import csv
import re
import sqlite3
import sys


def main(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "AGPRs", "LDS"])
        for r in rows:
            name = re.sub(r"\s+", " ", r[0])
            w.writerow([name, r[1], int(r[2]), round(r[3], 1), round(100.0 * r[2] / tot, 3), r[4], r[5], r[6], r[7], r[8]])
    print(f"{len(rows)} kernels, {tot / 1e6:.3f} ms total -> {out_csv}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
