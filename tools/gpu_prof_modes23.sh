#!/bin/bash
# kernel stats of the modes 2 / 3 mono pipelines (tools/prof_modes23_r2.py) -> gpurun_out/prof23
export TMPDIR=/tmp
rm -rf gpurun_out/prof23; mkdir -p gpurun_out/prof23
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof23 -o m23 -- python3 tools/prof_modes23_r2.py > gpurun_out/prof23/run.log 2>&1 < /dev/null; rc=$?; echo rc=$rc
tail -3 gpurun_out/prof23/run.log
[ $rc -eq 0 ] || exit $rc
f=$(find gpurun_out/prof23 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-8 "$f" | cut -c1-220 | head -20
