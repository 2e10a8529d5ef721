#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/r2h_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 8 "gpurun_out/r2h_$name.log"; if [ $rc -ge 124 ]; then exit $rc; fi; }
step pytest 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "resampl or spec or mono_pipeline or cli"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h/m -o t -- python3 tools/mode_bench.py 63 3 > gpurun_out/r2h/m.log 2>&1
grep '"channels": 1' gpurun_out/r2h/m.log | cut -c1-200
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2h/m/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"  {r['Name'][:80]:80s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
