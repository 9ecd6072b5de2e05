#!/bin/bash
# One GPU visit = a list of steps, run in order on the MI355X box from the repo root; everything lands under gpurun_out/<tag>/.
#
#   gpurun --timeout 1800 -- 'tools/gpu_visit.sh <tag> <step> [args] [-- <step> [args]] ...'
#
# steps
#   tests [pytest args]                 -m gpu suite (default: all of tests/, not -x)                  -> pytest[_N].log
#   bench <name> [bench.py args]        one bench line                                                  -> bench_<name>.json (+ .err)
#   kstats <name> [bench.py args]       rocprofv3 --kernel-trace --stats of a short bench run           -> kstats_<name>.csv + the kernels matching $KFILTER
#   ab <name> <label[=tuning]>...       kernel-level A/B: kstats per label, `--tuning` from the label    -> ab_<name>.txt   (extra bench args: $AB_ARGS)
#   pmc [bench.py args]                 tools/gpu_profile.sh (kernel stats + PMC passes in their own runs + HBM traffic)
#   run <name> <command...>             anything else, output captured                                  -> run_<name>.txt
#
# env: KFILTER = substrings of kernel names to print (default: the matcher's and the encoder's dominant kernels)
cd "$(dirname "$0")/.."
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
KFILTER=${KFILTER:-conv64r,conv128r,lg_blockf,attention32,gemm8,gemmr}

kernel_lines() {   # $1 = csv
  python - "$1" "$KFILTER" <<'PY'
import csv, sys
keys = [k for k in sys.argv[2].split(",") if k]
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in keys):
        print("  %-72s calls %5s avg %9.1f us min %9.1f max %9.1f" % (n.split("(")[0][-72:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}

kstats() {         # $1 = name, rest = bench args
  local name=$1; shift
  rm -rf /tmp/kt_$name
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o kt -- python bench.py --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 "$@" > "$OUT/kstats_$name.json" 2> "$OUT/kstats_$name.err"
  python tools/rocpd_summary.py /tmp/kt_$name/kt_results.db "$OUT/kstats_$name.csv" > /dev/null 2>&1
  rm -rf /tmp/kt_$name
  echo "== kstats $name $*"; kernel_lines "$OUT/kstats_$name.csv"
}

step() {
  local kind=$1; shift
  case $kind in
    tests)
      N=$((${N:-0} + 1)); local log=$OUT/pytest_$N.log
      if [ $# -eq 0 ]; then set -- tests; fi
      timeout ${PYTEST_TIMEOUT:-1800} python -m pytest "$@" -q -m gpu > "$log" 2>&1; echo "pytest rc=$? ($log)"; tail -${PYTEST_TAIL:-15} "$log" | cut -c1-300 ;;
    bench)
      local name=$1; shift
      timeout ${BENCH_TIMEOUT:-600} python bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "bench $name rc=$?"
      python tools/bench_brief.py "$OUT/bench_$name.json" ;;
    kstats) kstats "$@" ;;
    ab)
      local name=$1; shift
      { for lab in "$@"; do
          local tun=""; case $lab in *=*) tun=${lab#*=}; lab=${lab%%=*};; esac
          kstats ${name}_$lab ${tun:+--tuning $tun} ${AB_ARGS:-}
        done; } 2>&1 | tee "$OUT/ab_$name.txt" ;;
    pmc) BENCH_ARGS="$*" tools/gpu_profile.sh $TAG; mkdir -p "$OUT/prof"; cp -r gpurun_out/prof_$TAG/. "$OUT/prof/" ;;
    run)
      local name=$1; shift
      timeout ${RUN_TIMEOUT:-900} "$@" > "$OUT/run_$name.txt" 2>&1; echo "run $name rc=$?"; tail -${RUN_TAIL:-25} "$OUT/run_$name.txt" | cut -c1-300 ;;
    *) echo "unknown step $kind"; exit 2 ;;
  esac
}

args=()
for a in "$@" --; do
  if [ "$a" = "--" ]; then
    [ ${#args[@]} -gt 0 ] && step "${args[@]}"
    args=()
  else
    args+=("$a")
  fi
done
