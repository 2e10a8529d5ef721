#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rds.py -m gpu -q --timeout 300 -p no:cacheprovider -s > gpurun_out/r2j_pytest.log 2>&1; echo "rc=$?"; tail -n 25 gpurun_out/r2j_pytest.log
