#!/bin/bash
# round 2 acceptance pass on the GPU box: smoke, the whole -m gpu suite, the default bench line, and a 2-rank rehearsal of
# `bench.py --gpus 2` (both ranks on GPU 0: the launcher, rank environment and gloo control plane on real hardware)
set -u
mkdir -p gpurun_out/final
step() { local name=$1 t=$2; shift 2; echo "== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/final/$name.log" 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 6 "gpurun_out/final/$name.log" | cut -c1-600; if [ $rc -ge 124 ]; then exit $rc; fi; }
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step pytest 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider
step bench_default 500 python bench.py
step bench_two_ranks 300 python bench.py --gpus 2 --all-on-device0 --steps 20 --warmup 5
