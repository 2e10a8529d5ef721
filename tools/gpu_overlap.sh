#!/bin/bash
# option overlap_calls: parity tests, the stereo step-time comparison, the bench line's legs
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "overlapped or stereo or resampl" > gpurun_out/ov_tests.txt 2>&1 < /dev/null || { tail -40 gpurun_out/ov_tests.txt; exit 1; }
tail -3 gpurun_out/ov_tests.txt
timeout -k 10 300 python tools/stereo_overlap_bench.py 12 > gpurun_out/ov_bench.txt 2>&1 < /dev/null || { tail -30 gpurun_out/ov_bench.txt; exit 1; }
cat gpurun_out/ov_bench.txt
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/ov_benchline.txt 2>&1 < /dev/null || { tail -30 gpurun_out/ov_benchline.txt; exit 1; }
python3 - <<'PY'
import json
for l in open('gpurun_out/ov_benchline.txt'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['roofline']['frac'])
        for k, v in d['legs'].items():
            if isinstance(v, dict) and 'ms_per_step' in v: print(k, v['ms_per_step'], v['value'], v['frac'], v.get('pll_repaired_segments'))
PY
