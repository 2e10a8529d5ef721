#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/r2i_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 6 "gpurun_out/r2i_$name.log" | cut -c1-900; if [ $rc -ge 124 ]; then exit $rc; fi; }
step pytest 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "fused or mono or block_split or many_channels or random_block or bench"
B="--steps 50 --warmup 5 --no-cpu-baseline --no-side-legs"
for i in 1 2; do
step bench_new$i 200 python bench.py $B
FMRX_LIB=$PWD/software-defined-radio_amd/lib/libfmrx_base.so step bench_base$i 200 python bench.py $B
done
grep -h -o '"avg_launch_ms": [0-9.]*' gpurun_out/r2i_bench_new*.log gpurun_out/r2i_bench_base*.log
