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


def _bench(*args, env=None, timeout=240):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                       env=env or {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (the form the driver runs at N=1, and what a user types for N>1) must start
    two ranks by itself: n_gpus == 2 in the one JSON line, one channel seed per rank, max-over-ranks timing.
    The GPU step is replaced by a sleep of stub_ms x (rank+1) (bench.py's documented test hook), so this runs
    on the CPU: launcher, rank environment, gloo control plane, barrier, aggregation end to end."""
    rc, d, err = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--stub-ms", "30")
    assert rc == 0, err[-2000:]
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2
    assert d["seeds"] == [0x3D74, 0x3D74 + 1] and d["local_rank"] == 0
    el = d["ms_per_step"] * 5e-3
    assert 0.3 - 1e-3 <= el < 0.3 + 0.2                       # the slow rank (5 x 60 ms) sets the time
    assert abs(d["value"] - 2 * 1000 * 5 / el / 1e6) < 1e-9     # all ranks' units / the slowest rank's time
    # one rank, no launcher
    rc, d, err = _bench("--gpus", "1", "--steps", "3", "--warmup", "1", "--stub-ms", "10")
    assert rc == 0 and d["n_gpus"] == 1 and d["seeds"] == [0x3D74], err[-1000:]


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    """Under a launcher (WORLD_SIZE set) --gpus must agree with it: a 1-rank job must not claim 8 GPUs."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    rc, d, err = _bench("--gpus", "8", "--steps", "2", "--stub-ms", "5", env=env)
    assert rc != 0 and d and "error" in d


def test_bench_under_torch_distributed_run():
    """The driver's own N>1 form: torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 (stub step)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "4", "--warmup", "1", "--stub-ms", "20"],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["seeds"] == [0x3D74, 0x3D74 + 1]


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
