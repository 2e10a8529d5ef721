#!/bin/bash
# round 2, first GPU pass: whole -m gpu suite, stereo error-vs-time table (cause by cause), a short bench line.
# A step that is KILLED (timeout) stops the script; a step that merely fails (assertion) does not.
set -u
mkdir -p gpurun_out
step() {   # name, timeout, command...
  local name=$1 t=$2; shift 2
  echo "== $name" | tee -a gpurun_out/r2a_steps.log
  timeout -k 10 "$t" "$@" > "gpurun_out/r2a_$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/r2a_steps.log
  tail -n 15 "gpurun_out/r2a_$name.log"
  if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
}
: > gpurun_out/r2a_steps.log
step pytest 800 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider
step envelope 300 python tests/tools/stereo_envelope.py
step bench 400 python bench.py --steps 20 --warmup 5
