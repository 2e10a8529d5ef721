#!/bin/bash
# bank throughput against the number of receivers (one reference block each per call, and four): exact and fast
for e in 1 0; do
  echo "== exact=$e, 51200 samples per receiver and call"
  timeout -k 10 300 python3 tools/bank_bench.py --exact $e --channels 64,256,1024,4096,16384,65536,131072 --calls 3 2>&1 | grep channels
  echo "== exact=$e, 204800 samples per receiver and call"
  timeout -k 10 300 python3 tools/bank_bench.py --exact $e --channels 1024,4096,16384,32768 --blocks-per-call 4 --calls 3 2>&1 | grep channels
done
for m in 1; do echo "== exact=1, mode $m"; timeout -k 10 300 python3 tools/bank_bench.py --exact 1 --mode $m --channels 16384,65536 --calls 3 2>&1 | grep channels; done
