#!/bin/bash
# pll_segments_kernel's duration by lane-start method and warm-up length (kernel stats of tools/prof_stereo_r2.py)
export TMPDIR=/tmp
for cfg in "1 64" "0 64" "1 0" "1 128"; do set -- $cfg
rm -rf gpurun_out/prof_seg; mkdir -p gpurun_out/prof_seg
FMRX_PLL_START=$1 FMRX_PLL_WARMUP=$2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_seg -o st -- python3 tools/prof_stereo_r2.py > gpurun_out/prof_seg/run.log 2>&1 < /dev/null
python3 - "$1" "$2" <<'PY'
import csv, sys
for r in csv.DictReader(open('gpurun_out/prof_seg/st_kernel_stats.csv')):
    if 'pll_segments' in r['Name'] or 'pll_lti' in r['Name']:
        print(f"start={sys.argv[1]} W={sys.argv[2]}: {r['Name'][30:60]} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f}")
PY
done
