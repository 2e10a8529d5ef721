#!/bin/bash
set +e
for H in 4096 2048 1024 512 0; do
  echo "== head $H"
  FMRX_PLL_HEAD=$H timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "stereo or pll or state or ragged or cli" 2>&1 | grep -E "^E +Assertion|passed|failed" | head -4
  FMRX_PLL_HEAD=$H timeout -k 10 100 python3 tools/mode_bench.py 3 12 2>/dev/null | grep '"mode": 0, "channels": 2' | cut -c1-330
done
