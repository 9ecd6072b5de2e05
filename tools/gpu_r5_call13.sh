#!/bin/bash
# Round 5: where do the ~200 us of the descriptor head's gather GEMM go?  Kernel durations of the point-only step under rocprofv3 with the streaming gather
# (default in this tree) and with the tiled kernel (desc_gather_stream=0).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05q; mkdir -p $OUT
export TMPDIR=/tmp
for t in "desc_gather_stream=1" "desc_gather_stream=0"; do
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 --tuning $t > /dev/null 2> $OUT/err_$t.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$t.csv > /dev/null 2>&1
  echo "== $t"
  python - "$OUT/ks_$t.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("gemm8_kernel", "gemmr_gather", "sample_desc", "desc_cells", "head_softmax", "select_list", "conv128r_kernel<airfe::PF16, false>")):
        print("  %-60s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done 2>&1 | tee $OUT/desc_gather_kernels.txt
