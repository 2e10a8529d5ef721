#!/bin/bash
# round 3, parity at scale on the final build: every kernel shape on 8-27 M-sample blocks against the oracle, and the receiver banks over
# long streams in every mode (exact: bit patterns; fast: envelope).  Output: gpurun_out/evid3/ -> profiles/round3/10_*, 11_*
set -u
F=gpurun_out/evid3; rm -rf $F; mkdir -p $F
timeout -k 10 500 python3 tests/tools/big_shape_sweep.py > $F/10_big_shape_sweep.txt 2>&1; echo "sweep rc=$?"; tail -n 3 $F/10_big_shape_sweep.txt
for m in 1 2 3; do
  timeout -k 10 400 python3 tests/tools/bank_long_streams.py 10 8 $m > $F/11_bank_long_streams_mode$m.txt 2>&1; echo "mode $m rc=$?"; tail -n 4 $F/11_bank_long_streams_mode$m.txt | cut -c1-300
done
