#!/bin/bash
# Round 5: two small A/Bs at the kernel level (rocprofv3 kernel durations of the point-only step): pre-process with 8 rows per workgroup (libairfe_pre8.so.tmp)
# against 4, and the fused block's tokens per workgroup (112 against 128) now that the out-projection phase is gone.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05s; mkdir -p $OUT
export TMPDIR=/tmp
cp airslam_amd/libairfe.so /tmp/main.so
run() {   # $1 = label, $2 = tuning
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 ${2:+--tuning $2} > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  echo "== $1"
  python - "$OUT/ks_$1.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("preprocess", "lg_blockf", "attention32")):
        print("  %-64s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n.split("(")[0][-64:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
{ run main ""; cp airslam_amd/libairfe_pre8.so.tmp airslam_amd/libairfe.so; run pre8 ""; cp /tmp/main.so airslam_amd/libairfe.so; run tokens128 "lgb_tokens=128"; } 2>&1 | tee $OUT/small_ab.txt
