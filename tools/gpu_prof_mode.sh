#!/bin/bash
# kernel stats + SQ / MFMA / LDS / FETCH / WRITE counters (separate passes, counters never combined with tracing domains other than
# kernel-trace) of one mono mode's bench-leg workload: bash tools/gpu_prof_mode.sh <mode> [blocks]
set +e
export TMPDIR=/tmp
M=${1:-1}; B=${2:-256}
OUT=gpurun_out/prof_mode$M
rm -rf $OUT; mkdir -p $OUT
T="python3 tools/prof_mode.py $M $B"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $T 30 > $OUT/trace.log 2>&1 < /dev/null; echo "trace rc=$?"
pmc() { local name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- $T 4 > $OUT/$name.log 2>&1 < /dev/null; echo "$name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc mfma SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
python3 - "$M" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
m = sys.argv[1]
out = f'gpurun_out/prof_mode{m}'
print(f"== mode {m}: rocprofv3 --kernel-trace --stats -- python3 tools/prof_mode.py {m} (30 steps)")
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    for i, r in enumerate(csv.DictReader(open(f))):
        if i < 6: print("  {Name:.90s} calls={Calls} avg_ns={AverageNs} min_ns={MinNs} max_ns={MaxNs} pct={Percentage}".format(**r))
print("== counters per dispatch (separate --pmc passes, 4 steps each; averages over the dispatches of each kernel)")
for name in ("fetch", "write", "sq1", "sq2", "mfma"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/{name}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(s in k for s in ("mono_fused", "fe_mfma", "resample", "audio_fir")):
                agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(f"  [{name}] {k}: " + ", ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in d.items()))
PY
