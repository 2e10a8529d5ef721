#!/bin/bash
# HBM-side traffic and SQ/LDS counters of the modes 2 / 3 kernels (tools/prof_modes23_r2.py), one counter set per pass
export TMPDIR=/tmp
OUT=gpurun_out/pmc23; rm -rf $OUT; mkdir -p $OUT
pmc() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- python3 tools/prof_modes23_r2.py > $OUT/$name.log 2>&1 < /dev/null; rc=$?; echo "$name rc=$rc"; [ $rc -eq 0 ] || exit $rc; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pmc sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU
pmc mfma SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8
python3 - <<'PY'
import csv, glob, collections
for d in ("fetch", "write", "sq1", "sq2", "mfma"):
    fs = glob.glob(f"gpurun_out/pmc23/{d}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "fmrx" not in k: continue
        k = k.split("(anonymous namespace)::")[-1][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(d, k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, flush=True)
PY
