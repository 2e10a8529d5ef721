#!/bin/bash
# the -m gpu suite three times in a row (one process each, one after the other): run-to-run stability of the multi-stream paths
set -e
mkdir -p gpurun_out
for r in 1 2 3; do
  timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/suite_$r.txt 2>&1 < /dev/null || { tail -30 gpurun_out/suite_$r.txt; exit 1; }
  tail -1 gpurun_out/suite_$r.txt
done
