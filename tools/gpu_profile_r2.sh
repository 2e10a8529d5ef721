#!/bin/bash
# round-2 committed profiles: the bench command under rocprofv3 (kernel trace + stats), FETCH/WRITE and SQ/LDS PMC
# passes of the same command (counters only, never combined with tracing domains other than kernel-trace), the
# stereo step's kernel stats, and a plain default bench run.  Output: gpurun_out/profr2/ -> tools/save_profile_r2.py
set +e
export TMPDIR=/tmp
OUT=gpurun_out/profr2
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-side-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 30 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1; echo "trace rc=$?"
pmc() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- $B --steps 5 --warmup 2 --settle-ms 0 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc pmc_sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc pmc_mfma SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stereo -o stereo -- python3 tools/prof_stereo_r2.py > $OUT/stereo.log 2>&1; echo "stereo rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/modes23 -o m23 -- python3 tools/prof_modes23_r2.py > $OUT/modes23.log 2>&1 < /dev/null; echo "modes23 rc=$?"
timeout -k 10 400 python3 bench.py > $OUT/bench_default.log 2>&1; echo "bench default rc=$?"
tail -c 600 $OUT/bench_default.log
