#!/bin/bash
set +e
timeout -k 10 300 python -m pytest tests -m gpu -q --timeout 120 -k "front_end" 2>&1 | tail -2
timeout -k 10 200 python3 tools/fe_ab.py 256 8 2>/dev/null | tail -3
