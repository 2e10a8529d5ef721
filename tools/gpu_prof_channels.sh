#!/bin/bash
export TMPDIR=/tmp
rm -rf gpurun_out/prof_ch; mkdir -p gpurun_out/prof_ch
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ch -o ch -- python3 tools/prof_channels_r2.py > gpurun_out/prof_ch/run.log 2>&1 < /dev/null; echo rc=$?
tail -2 gpurun_out/prof_ch/run.log
