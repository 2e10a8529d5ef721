#!/bin/bash
# resampler check: parity tests that touch it + modes 2/3 throughput
set +e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "resampl or pipeline or mono or stereo" > gpurun_out/pytest_rs.log 2>&1; rc=$?
tail -2 gpurun_out/pytest_rs.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python tools/mode_bench.py 63 3 2>/dev/null | grep '"mode": [23], "channels": 1'
