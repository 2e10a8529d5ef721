#!/bin/bash
# A/B of two builds of the library on one box: the headline bench command, interleaved (FMRX_LIB selects the library)
#   software-defined-radio_amd/lib/libfmrx_base.so = the build to compare with (made by hand from the commit in question)
set -e
mkdir -p gpurun_out
B=software-defined-radio_amd/lib/libfmrx_base.so
[ -f "$B" ] || { echo "no $B"; exit 1; }
for r in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export FMRX_LIB=$PWD/$B; else unset FMRX_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-side-legs --steps 60 --warmup 10 > gpurun_out/ab_$which.txt 2>&1 < /dev/null
    python3 - $which <<'PY'
import json, sys
for l in open(f'gpurun_out/ab_{sys.argv[1]}.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['frac'], flush=True)
PY
  done
done
