#!/bin/bash
# quick GPU check: parity tests, bench (no CPU baseline), one PMC pass
set +e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench.log 2>&1; rc=$?
if [ $rc -ge 124 ]; then exit $rc; fi
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench.log') if l.startswith('{')][-1])
r=d['roofline']; print('value',d['value'],'ms/step',d['ms_per_step'],'fe_ms',r['avg_launch_ms'],'frac',r['frac'],'stages',r['stage_ms'])
PY
export TMPDIR=/tmp
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/pmc_q; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/prof/pmc_q -o pmc_q -- python3 tools/prof_target.py 3 256 > gpurun_out/prof/pmc_q.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof 2>/dev/null | grep -A9 "pmc_q" | head -24
