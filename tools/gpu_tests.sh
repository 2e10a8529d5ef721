#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 300 -rA > gpurun_out/pytest_gpu.log 2>&1
rc=$?
grep -E "passed|failed|FAILED|ERROR|rms err|pll max" gpurun_out/pytest_gpu.log | tail -60
exit $rc
