#!/bin/bash
# SQ counters of the bank's kernels (separate --pmc passes): args are passed to tools/bank_bench.py
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bank
rm -rf $OUT; mkdir -p $OUT
pmc() { local name=$1; shift; local ctr="$1"; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/$name -o pmc -- python3 tools/bank_bench.py "$@" > $OUT/$name.log 2>&1 < /dev/null; echo "$name rc=$?"; }
pmc sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "$@"
pmc sq2 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "$@"
pmc mem "FETCH_SIZE" "$@"
pmc memw "WRITE_SIZE" "$@"
python3 - <<'PY'
import csv, glob, collections
out='gpurun_out/pmc_bank'
for name in ("sq1","sq2","mem","memw"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/{name}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if any(s in k for s in ("chs_","pll_channels","fe_mfma_bank")):
                k=k[k.find('::',25)+2:][:34] if 'anonymous' in k else k[:34]
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        print(f"[{name}] {k}: "+", ".join(f"{c}={sum(v)/len(v):.4g}" for c,v in d.items()))
PY
