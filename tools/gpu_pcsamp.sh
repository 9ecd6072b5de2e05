#!/bin/bash
# Instruction-level evidence for the matcher's two kernels.
#   1. rocprofv3 --att (thread trace): needs the trace decoder library, which this image does not ship — the attempt and its message are recorded.
#   2. rocprofv3 PC sampling (beta), stochastic (hardware) sampling where the GPU offers it, else host-trap: PCs with the hardware's own stall reason per sample.
# Output: gpurun_out/<tag>/{att_attempt.txt, pcsamp_*}.  Summaries: tools/pcsamp_summary.py.
cd "$(dirname "$0")/.."
TAG=${1:-r06pc}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --detector superpoint --pairs 64 --steps 3 --warmup 1 --cpu-pairs 0 --no-profile --stage-steps 0 --io-steps 0"
{ echo "== rocprofv3 --att"; timeout 300 rocprofv3 --att --kernel-include-regex "lg_blockf|attention32" --att-target-cu 1 -d /tmp/att -o att -- $CMD 2>&1 | tail -25; echo "rc=$?"; ls -R /tmp/att 2>/dev/null | head -30; } > $OUT/att_attempt.txt 2>&1
tail -12 $OUT/att_attempt.txt
{ echo "== rocprofv3 -L (pc sampling configurations)"; timeout 120 rocprofv3 -L 2>&1 | grep -i -A12 "pc.sampl" | head -60; } > $OUT/pcsamp_avail.txt 2>&1
cat $OUT/pcsamp_avail.txt | head -40
for METHOD in stochastic host_trap; do
  UNIT=cycles; INT=${PCS_INTERVAL:-4096}
  if [ $METHOD = host_trap ]; then UNIT=time; INT=${PCS_INTERVAL_US:-10}; fi
  rm -rf /tmp/pcs_$METHOD
  ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT --pc-sampling-interval $INT \
      --kernel-trace --output-format csv -d /tmp/pcs_$METHOD -o pcs -- $CMD > $OUT/pcsamp_${METHOD}_bench.json 2> $OUT/pcsamp_${METHOD}.err
  echo "pc sampling $METHOD rc=$?"; tail -5 $OUT/pcsamp_${METHOD}.err
  find /tmp/pcs_$METHOD -type f | head -20
  for f in $(find /tmp/pcs_$METHOD -name "*pc_sampling*.csv"); do
    echo "-- $f: $(wc -l < $f) lines"; head -3 $f | cut -c1-600
    python tools/pcsamp_summary.py $f $(find /tmp/pcs_$METHOD -name "*kernel_trace.csv" | head -1) > $OUT/pcsamp_${METHOD}_summary.txt 2>&1
    head -c 40000000 $f | gzip > $OUT/pcsamp_${METHOD}_head.csv.gz      # (the first 40 MB: enough to re-summarise offline)
  done
  [ -s $OUT/pcsamp_${METHOD}_summary.txt ] && head -60 $OUT/pcsamp_${METHOD}_summary.txt
done
