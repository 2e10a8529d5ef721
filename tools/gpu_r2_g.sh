#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/r2g_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 12 "gpurun_out/r2g_$name.log"; if [ $rc -ge 124 ]; then exit $rc; fi; }
step pytest 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "stereo or pll or state or ragged or cli or spec"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2g/st -o t -- python3 tools/prof_stereo_r2.py > gpurun_out/r2g/st.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r2g/st/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(f"  {r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
step envelope 300 python tests/tools/stereo_envelope.py
