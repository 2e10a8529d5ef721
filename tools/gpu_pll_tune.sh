#!/bin/bash
set +e
for LW in "512 512" "1024 512" "512 768" "256 768"; do
  set -- $LW
  echo "== L=$1 W=$2"
  FMRX_PLL_SEGMENT=$1 FMRX_PLL_WARMUP=$2 timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "stereo or pll or state or ragged or cli" 2>&1 | grep -E "^E +Assertion|passed|failed" | head -4
done
