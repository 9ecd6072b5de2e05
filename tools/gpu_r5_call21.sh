#!/bin/bash
# Round 5: attention32_kernel with tile 0's DMA issued before the Q loads (one memory round trip in the prologue instead of two): the same bits (md5), kernel durations
# attnq (= Q first, the previous commit) / main / attnq / main.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05y; mkdir -p $OUT
export TMPDIR=/tmp
cp airslam_amd/libairfe.so /tmp/main.so
{ echo -n "main:  "; python tools/lib_scores_hash.py 2>/dev/null | tail -1; cp airslam_amd/libairfe_attnq.so.tmp airslam_amd/libairfe.so; echo -n "attnq: "; python tools/lib_scores_hash.py 2>/dev/null | tail -1; cp /tmp/main.so airslam_amd/libairfe.so; } | tee $OUT/hash.txt
run() {   # $1 = label
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  python - "$OUT/ks_$1.csv" "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "attention32" in n:
        print("  %-6s %-40s calls %4s avg %9.2f us min %9.2f max %9.2f" % (sys.argv[2], n.split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for v in attnq main attnq main; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  run $v
done 2>&1 | tee $OUT/attn_prologue_ab.txt
cp /tmp/main.so airslam_amd/libairfe.so
