#!/bin/bash
# Round 5: pre-process with four output rows per workgroup: bit-exactness (seven sizes, strided rows, the batch paths), then its kernel time under rocprofv3.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05r; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_detector.py tests/test_gpu_stereo.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > /dev/null 2> $OUT/err.txt
python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks.csv > /dev/null 2>&1
python - "$OUT/ks.csv" <<'PY' | tee $OUT/preprocess_kernel.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("preprocess", "lg_prepare", "select_list", "nms512", "gemmr_gather")):
        print("  %-60s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
