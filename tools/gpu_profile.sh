#!/bin/bash
# rocprofv3 on the bench workload: (1) kernel trace + stats, (2..) PMC passes
# (counters only, never combined with tracing domains other than kernel-trace).
set +e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/prof_target.py 10 256 > $OUT/trace.log 2>&1
echo "trace rc=$?"
pmc() { # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o $name -- python3 tools/prof_target.py 3 256 > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
pmc pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc pmc_sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.csv" | head -40
python3 tools/prof_summary.py $OUT > gpurun_out/prof_summary.txt 2>&1
cat gpurun_out/prof_summary.txt | head -80
rocprofv3 -L > gpurun_out/counters.txt 2>&1 || true
