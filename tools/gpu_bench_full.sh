#!/bin/bash
# the default bench run, as the driver runs it, timed
mkdir -p gpurun_out
S=$(date +%s)
timeout -k 10 900 python3 bench.py "$@" > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; rc=$?
E=$(date +%s)
echo "rc=$rc wall=$((E-S))s"
tail -3 gpurun_out/bench_full.err
python3 - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_full.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print('value',d['value'],'ms/step',d['ms_per_step'],'frac',d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
    print('library', d['config']['library'])
    for k,v in d.get('legs',{}).items():
        print(' ',k, {kk:vv for kk,vv in v.items() if kk in ('value','frac','ms_per_step','error','x_real_time','channels_at_real_time','ms_per_call')})
    c=d.get('cpu_baseline',{})
    print('cpu', c.get('value'), {k:v['value'] for k,v in c.get('legs',{}).items()})
PY
exit $rc
