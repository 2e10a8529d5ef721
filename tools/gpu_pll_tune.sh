#!/bin/bash
set +e
for W in 3072 1024 512; do
  FMRX_PLL_WARMUP=$W timeout -k 10 100 python3 tools/prof_target_stereo.py 0 2>/dev/null | tail -1
done
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "stereo or pll or state or ragged or cli" 2>&1 | tail -2
