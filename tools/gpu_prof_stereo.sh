#!/bin/bash
set +e
export TMPDIR=/tmp
OUT=gpurun_out/profs
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/prof_target_stereo.py 0 > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/profs/trace/**/*kernel_stats.csv', recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 12: print("  {Name:.70s} calls={Calls} avg_ns={AverageNs} total_ns={TotalDurationNs} pct={Percentage}".format(**r))
PY
