#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel trace + stats, then PMC passes in their OWN runs (never with other tracing).
#   tools/gpu_profile.sh <tag>        (on the MI355X box, from the repo root; writes under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-profile ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python tools/rocpd_summary.py $OUT/kt/kt_results.db $OUT/kernel_stats.csv
PC="python bench.py --steps 2 --warmup 1 --cpu-pairs 0 --no-profile ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pa -o a -- $PC > /dev/null 2> $OUT/pa.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 --kernel-trace -d $OUT/pb -o b -- $PC > /dev/null 2> $OUT/pb.err
python tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pa/a_results.db $OUT/pb/b_results.db > $OUT/pmc_summary.txt 2>&1
if [ "${TRAFFIC:-1}" = "1" ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pf -o f -- $PC > /dev/null 2> $OUT/pf.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pw -o w -- $PC > /dev/null 2> $OUT/pw.err
  python tools/pmc_traffic.py $OUT/pf/f_results.db $OUT/pw/w_results.db $OUT/hbm_traffic.json > $OUT/hbm_traffic.txt 2>&1
fi
# keep the merge-back small: the sqlite traces stay on the box
rm -rf $OUT/kt $OUT/pa $OUT/pb $OUT/pf $OUT/pw
head -40 $OUT/pmc_summary.txt
