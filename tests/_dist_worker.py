"""Worker for tests/test_multiproc_cpu.py: the bench's N>1 control plane on CPU
(gloo): per-rank channel seed, barrier + max-over-ranks timing, job throughput."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rank, local_rank, world, dist = bench.init_ranks("gloo")
assert dist is not None and world == 2
seed = bench.channel_seed(rank)
calls = []


def step():
    calls.append(1)
    time.sleep(0.02 * (rank + 1))      # rank 1 is the slow channel


elapsed = bench.timed_region(step, lambda: None, steps=5, warmup=2, dist=dist)
mine = 0.02 * (rank + 1) * 5
import torch  # noqa: E402
seeds = [None, None]
dist.all_gather_object(seeds, seed)
if rank == 0:
    print(json.dumps({"elapsed_max": elapsed, "rank0_own": mine, "seeds": seeds, "calls": len(calls),
                      "value": bench.job_throughput(world, 1000, 5, elapsed)}))
dist.barrier()
dist.destroy_process_group()
