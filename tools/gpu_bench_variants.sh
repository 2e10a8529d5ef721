#!/bin/bash
set +e
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --fe-variant valu --no-cpu-baseline --steps 30 > gpurun_out/bench_valu.log 2>&1; echo "valu rc=$?"; tail -1 gpurun_out/bench_valu.log | cut -c1-1800
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --dist-backend gloo --all-on-device0 > gpurun_out/bench_n2.log 2>&1; echo "n2 rc=$?"; grep '^{' gpurun_out/bench_n2.log | cut -c1-700
