#!/bin/bash
# receiver banks in every mode, both flavours, stereo and mono (tools/bank_bench.py; inputs resident in HBM)
for m in 0 1 2 3; do
  for e in 1 0; do
    echo "== mode $m, exact=$e, stereo"; timeout -k 10 200 python3 tools/bank_bench.py --mode $m --exact $e --channels 16384 --blocks-per-call 4 --calls 3 2>&1 | grep channels
  done
done
for m in 0 1 2 3; do
  echo "== mode $m, exact=1, mono"; timeout -k 10 200 python3 tools/bank_bench.py --mode $m --exact 1 --audio-channels 1 --channels 16384 --blocks-per-call 4 --calls 3 2>&1 | grep channels
done
for m in 2 3; do
  echo "== mode $m, exact=0, mono"; timeout -k 10 200 python3 tools/bank_bench.py --mode $m --exact 0 --audio-channels 1 --channels 16384 --blocks-per-call 4 --calls 3 2>&1 | grep channels
done
