export TMPDIR=/tmp
rm -rf gpurun_out/prof_lti; mkdir -p gpurun_out/prof_lti
FMRX_PLL_START=1 FMRX_PLL_WARMUP=64 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lti -o st -- python3 tools/prof_stereo_r2.py > gpurun_out/prof_lti/run.log 2>&1 < /dev/null; echo rc=$?
tail -2 gpurun_out/prof_lti/run.log
