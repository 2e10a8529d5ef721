#!/bin/bash
# resampler kernel: parity subset, chain-count A/B, kernel stats
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "resampl or mode2 or mode3 or modes or mode_2 or mode_3 or stereo or state or pipeline" > gpurun_out/rs_tests.txt 2>&1 < /dev/null || { tail -30 gpurun_out/rs_tests.txt; exit 1; }
tail -3 gpurun_out/rs_tests.txt
timeout -k 10 300 python tools/resample_chains_ab.py > gpurun_out/rs_chains.txt 2>&1 < /dev/null || { tail -30 gpurun_out/rs_chains.txt; exit 1; }
cat gpurun_out/rs_chains.txt
bash tools/gpu_prof_modes23.sh
