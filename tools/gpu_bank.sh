#!/bin/bash
# round 3: the receiver banks (tests/test_gpu_channels.py); $1 = pytest -k expression (optional)
set +e
mkdir -p gpurun_out
K="${1:-not 256}"
timeout -k 10 900 python -m pytest tests/test_gpu_channels.py tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "$K" > gpurun_out/pytest_bank.log 2>&1; rc=$?
tail -40 gpurun_out/pytest_bank.log
exit $rc
