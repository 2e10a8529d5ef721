#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, microbench, bench.  A step that
# times out stops the script (no further GPU work after a hang).
set +e
mkdir -p gpurun_out
step() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a gpurun_out/steps.log
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/steps.log
    if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping"; exit $rc; fi
    return $rc
}
rocminfo | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/device.log 2>&1
nproc >> gpurun_out/device.log
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step pytest_gpu 600 python -m pytest tests -m gpu -q --timeout 300 -rA
step valu 120 ./tools/microbench/valu_rates
step bench 420 python bench.py --steps 20 --warmup 3
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench.log
