#!/bin/bash
# option overlap_calls: parity test, then the step-time comparison
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "overlapped or stereo" > gpurun_out/ov_tests.txt 2>&1 < /dev/null || { tail -40 gpurun_out/ov_tests.txt; exit 1; }
tail -3 gpurun_out/ov_tests.txt
timeout -k 10 300 python tools/stereo_overlap_bench.py 12 18 24 36 48 > gpurun_out/ov_bench.txt 2>&1 < /dev/null || { tail -30 gpurun_out/ov_bench.txt; exit 1; }
cat gpurun_out/ov_bench.txt
