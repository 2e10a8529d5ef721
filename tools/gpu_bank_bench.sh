#!/bin/bash
# round 3: bank timing + per-kernel split; args are passed to tools/bank_bench.py
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/bank_bench.py "$@" > gpurun_out/bank_bench.log 2>&1; rc=$?
cat gpurun_out/bank_bench.log
if [ $rc -ne 0 ]; then exit $rc; fi
rm -rf gpurun_out/prof_bank; mkdir -p gpurun_out/prof_bank
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bank -o bank -- python3 tools/bank_bench.py --channels 16384 --calls 3 > gpurun_out/prof_bank/run.log 2>&1 < /dev/null; echo rc=$?
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_bank/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
