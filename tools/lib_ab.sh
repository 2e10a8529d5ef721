#!/bin/bash
# same-box A/B of two builds of libfmrx.so (tools/ab/libfmrx_old.so vs tools/ab/libfmrx_new.so), alternating
set +e
mkdir -p gpurun_out
: > gpurun_out/lib_ab.log
for rep in 1 2 3; do
  for v in old new; do
    FMRX_LIB=$PWD/tools/ab/libfmrx_$v.so timeout -k 10 120 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1; rc=$?
    if [ $rc -ge 124 ]; then echo "timeout $v"; exit $rc; fi
    python3 - $v <<'PY' | tee -a gpurun_out/lib_ab.log
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_%s.log'%sys.argv[1]) if l.startswith('{')][-1])
r=d['roofline']; print(sys.argv[1],'value',d['value'],'ms/step',d['ms_per_step'],'stages',r['stage_ms'])
PY
  done
done
