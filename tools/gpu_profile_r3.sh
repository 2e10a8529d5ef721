#!/bin/bash
# round-3 committed profiles of the bench command itself (default: 1024 blocks resident): kernel trace + stats, FETCH / WRITE PMC passes
# (counters only, never combined with tracing domains other than kernel-trace), and a plain default bench run.
# Output: gpurun_out/profr3/ -> tools/save_profile_r3.py
set +e
export TMPDIR=/tmp
OUT=gpurun_out/profr3
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-side-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $B --steps 30 --warmup 5 > $OUT/bench_under_rocprof.log 2>&1 < /dev/null; echo "trace rc=$?"
pmc() { local name=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$name -o pmc -- $B --steps 5 --warmup 2 --settle-ms 0 > $OUT/$name.log 2>&1 < /dev/null; echo "$name rc=$?"; }
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
timeout -k 10 600 python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench default rc=$?"
python3 tools/save_profile_r3.py
