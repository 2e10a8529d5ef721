#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/r2e_$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 14 "gpurun_out/r2e_$name.log"; if [ $rc -ge 124 ]; then exit $rc; fi; }
step pytest 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -k "stereo or pll or state or ragged or cli or spec or pipeline"
step stereo_bench 200 python tools/stereo_bench.py
step envelope 300 python tests/tools/stereo_envelope.py
