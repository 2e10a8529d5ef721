#!/bin/bash
set -u
mkdir -p gpurun_out
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/r2f_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 12 "gpurun_out/r2f_$name.log"; if [ $rc -ge 124 ]; then exit $rc; fi; }
step pytest 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider
step bench 400 python bench.py --steps 20 --warmup 5
