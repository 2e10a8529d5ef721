#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, A/B, bench.  A step that
# times out stops the script (no further GPU work after a hang).
set +e
mkdir -p gpurun_out
step() {  # name, timeout, command...
    local name=$1 to=$2; shift 2
    echo "== $name" | tee -a gpurun_out/steps.log
    timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a gpurun_out/steps.log
    if [ $rc -ge 124 ]; then echo "step $name timed out / was killed: stopping"; tail -20 "gpurun_out/$name.log"; exit $rc; fi
    return $rc
}
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step pytest_gpu 400 python -m pytest tests -m gpu -q --timeout 120 -rA
step bench 300 python bench.py
step mode_bench 400 python tools/mode_bench.py 63 12
step cli_bench 300 python tools/cli_bench.py
tail -3 gpurun_out/smoke.log; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_gpu.log | tail -15; tail -1 gpurun_out/bench.log; tail -30 gpurun_out/mode_bench.log; tail -12 gpurun_out/cli_bench.log
