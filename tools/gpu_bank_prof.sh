#!/bin/bash
# per-kernel split of a bank configuration: args are passed to tools/bank_bench.py
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bank; mkdir -p gpurun_out/prof_bank
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bank -o bank -- python3 tools/bank_bench.py "$@" > gpurun_out/prof_bank/run.log 2>&1 < /dev/null; echo rc=$?
cat gpurun_out/prof_bank/run.log | grep channels
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_bank/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
rows=[]
for f in glob.glob('gpurun_out/prof_bank/**/*kernel_trace.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
ks=[r for r in rows if 'chs_' in r['Kernel_Name'] or 'pll_channels' in r['Kernel_Name'] or 'fe_mfma_bank' in r['Kernel_Name']]
if ks:
    n=len(ks); last=ks[-(n//5 if n>=5 else n):]
    t0=int(last[0]['Start_Timestamp'])
    for r in last[:40]:
        nm=r['Kernel_Name']; nm=nm[nm.find('::',25)+2:][:30] if 'anonymous' in nm else nm[:30]
        print(f"{nm:32s} q={r['Queue_Id']} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}")
PY
