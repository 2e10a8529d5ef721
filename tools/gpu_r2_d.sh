#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
for cfg in "768 64 0" "512 64 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2d/$tag -o t -- python3 tools/prof_stereo_r2.py $cfg > gpurun_out/r2d/$tag.log 2>&1
  rc=$?; echo "cfg $cfg rc=$rc"; tail -n 2 gpurun_out/r2d/$tag.log
  if [ $rc -ge 124 ]; then exit $rc; fi
  f=$(find gpurun_out/r2d/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_us={float(r['TotalDurationNs'])/1e3:9.1f}")
PY
done
