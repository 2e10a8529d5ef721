#!/bin/bash
set +e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "stereo or pll or state or ragged or cli" > gpurun_out/pytest_st.log 2>&1; rc=$?
tail -2 gpurun_out/pytest_st.log
if [ $rc -ge 124 ]; then exit $rc; fi
for m in 0 3; do timeout -k 10 200 python3 tools/prof_target_stereo.py $m 2>/dev/null | tail -1; done
