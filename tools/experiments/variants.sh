# kernel times of the line path per library variant (airslam_amd/libairfe_<name>.so.tmp), one box
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export AIRFE_OVERLAP_LINES=0
cp airslam_amd/libairfe.so /tmp/main.so
for v in main "$@"; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-profile --cpu-pairs 0 > /dev/null 2>&1
  python tools/rocpd_summary.py /tmp/kt/kt_results.db /tmp/ks.csv > /dev/null
  python - "$v" <<'PY'
import csv, sys
out=[]
for r in csv.DictReader(open('/tmp/ks.csv')):
    if 'plnet_s1_kernel' in r['Name'] or 's0_j2l_grid' in r['Name'] or 's0_head_decode' in r['Name']: out.append((r['Name'].split('(')[0][-28:], round(float(r['AverageNs'])/1e3,1)))
print(sys.argv[1], out)
PY
done
cp /tmp/main.so airslam_amd/libairfe.so
