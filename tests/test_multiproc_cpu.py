"""The N>1 path of bench.py on CPU: 2 ranks over gloo.  Channels are independent
(no data-path collective); the process group only carries the barrier and the
max-over-ranks of the elapsed time, and the job figure is all ranks' samples over
the slowest rank's time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_control_plane():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "tests", "_dist_worker.py")],
                       capture_output=True, text=True, env=env, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["seeds"] == [0x3D74, 0x3D74 + 1]                 # one independent channel per rank
    assert d["calls"] == 7                                      # 2 warm-up + exactly 5 timed steps
    assert d["elapsed_max"] >= 0.2 - 1e-3                       # the slow rank (5 x 40 ms) sets the time
    assert d["elapsed_max"] < 0.2 + 0.15
    assert abs(d["value"] - 2 * 1000 * 5 / d["elapsed_max"] / 1e6) < 1e-9


def test_single_rank_helpers():
    sys.path.insert(0, ROOT)
    import bench
    os.environ.pop("WORLD_SIZE", None)
    rank, local_rank, world, dist = bench.init_ranks("gloo")
    assert (rank, world, dist) == (0, 1, None)
    n = []
    t = bench.timed_region(lambda: n.append(1), lambda: None, steps=3, warmup=1, dist=None)
    assert len(n) == 4 and t >= 0
    assert bench.job_throughput(8, 1_000_000, 10, 1.0) == 80.0
