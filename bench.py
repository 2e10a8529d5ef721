#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FM-receiver hot path on MI355X.

Metric (BASELINE.json): complex I/Q mega-samples per second through the
receiver chain, per job (all GPUs), and the dominant kernel's fraction of the
HBM roofline.

Workload at N=1 = BASELINE.json configs[1]: mode 0 mono, 101-tap front-end
FIR + decimate(10) + FM discriminator + 101-tap audio FIR + decimate(5) + s16
pack, synthetic 2.4 MS/s FM I/Q, blocks of 1,024,000 complex samples.  One
"step" = one pass of the whole chain over a device-resident batch of
`--blocks` (default 256) consecutive blocks of one stream (0.5 GB of u8 I/Q),
submitted as one block-parallel call (the mono chain is a sliding-window map of
its input, SURVEY A.4, so this equals block-by-block streaming: bit for bit in the
IF / discriminator stages and the carried state, to float32 summation order
(<= 2e-6) in the audio FIR of the fused kernel --
tests/test_gpu_parity.py::test_fused_mono_kernel, ::test_block_split_invariance_on_device).  Inputs are
resident in HBM when the timed region starts; outputs (f32 audio + s16 PCM) are
written to HBM.  N>1: one process per GPU, one independent channel per GPU
(seed + rank), no data-path collective (RCCL is used only for the barrier and
the max-over-ranks of the elapsed time): weak scaling.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BLOCK_SAMPLES = 1_024_000          # complex samples per block (= 20 reference blocks)
PROF_EVERY = 4                     # HIP events around every 4th launch of the timed region
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic HBM bytes per complex input sample of the dominant kernel = the whole mono chain
# fused (S3): 2 B of u8 I/Q in + (4 B f32 audio + 2 B s16 PCM) per rf_decim*audio_decim
# = 50 input samples out.  (SURVEY's S3 figures are 2.04 B for s16 only, 2.08 B for f32 only; this
# workload writes both.)  S2 (front end + discriminator to HBM) would be 2.4 B, S1 (IF only) 2.8 B.
FE_BYTES_PER_SAMPLE = 2.0 + (4.0 + 2.0) / 50.0
FE_FLOP_PER_SAMPLE = 2 * 2 * 101 / 10.0 + 9 / 10.0 + 2 * 101 / 50.0     # useful FIR + discriminator flops


def cpu_baseline(seconds: float = 10.0) -> dict:
    """The oracle (C restatement of the reference, oracle/fm_oracle.c, -O3 like
    src/Makefile:4) timed on this host: same chain, same taps, same signal.
    Single thread = the reference-equivalent figure; then one independent
    channel per available core."""
    from concurrent.futures import ThreadPoolExecutor

    from _oracle import Oracle  # checker library: allowed here (cpu_baseline leg) only
    synth = importlib.import_module("software-defined-radio_amd.synth")
    o = Oracle()
    nblk_ref = 40                                        # 40 reference blocks = 2,048,000 samples per pass
    iq = synth.synth_fm_u8(51200 * nblk_ref)

    def run(budget_s: float, seed_off: int = 0) -> tuple[int, float]:
        pl = o.pipeline(0, 1)
        data = iq if seed_off == 0 else np.roll(iq, 2 * seed_off)
        t0 = time.perf_counter()
        done = 0
        while True:
            for b in range(nblk_ref):
                pl.process(data[b * 102400:(b + 1) * 102400])
            done += 51200 * nblk_ref
            if time.perf_counter() - t0 >= budget_s:
                break
        return done, time.perf_counter() - t0

    n1, t1 = run(seconds)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    with ThreadPoolExecutor(cores) as ex:                # ctypes releases the GIL inside the C call
        t0 = time.perf_counter()
        res = list(ex.map(lambda c: run(seconds * 0.6, c + 1), range(cores)))
        wall = time.perf_counter() - t0
    return {
        "value": round(n1 / t1 / 1e6, 2), "unit": "MS/s", "cores": 1, "kind": "port",
        "sample": f"mode-0 mono chain (101/101 taps), {n1} complex samples in {t1:.1f} s, single thread, "
                  "oracle/fm_oracle.c built -O3 (reference flags)",
        "all_cores": {"value": round(sum(r[0] for r in res) / wall / 1e6, 2), "unit": "MS/s", "cores": cores,
                      "sample": f"{cores} independent channels, one per core, {wall:.1f} s"},
    }


def init_ranks(backend: str | None):
    """One process per GPU (torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE).
    Returns (rank, local_rank, world, dist-or-None).  The process group is CONTROL
    PLANE only (barrier + max of the elapsed time): channels never exchange data."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend or "gloo", rank=rank, world_size=world)
    return rank, local_rank, world, dist


def channel_seed(rank: int) -> int:
    """Channel c of the job = the synthetic stream with seed 0x3D74 + c."""
    return 0x3D74 + rank


def timed_region(step, sync, steps: int, warmup: int, dist, device=None) -> float:
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds."""
    import torch

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed


def job_throughput(world: int, samples_per_step_per_rank: int, steps: int, elapsed_max: float) -> float:
    """Whole-job MS/s: units all ranks processed / the slowest rank's time."""
    return world * samples_per_step_per_rank * steps / elapsed_max / 1e6


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=256, help="1,024,000-sample blocks resident per step")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed steps run for this long before the warm-up, so that the clocks the chip holds under "
                         "this load are reached (a step is ~0.13 ms; the first ~300 after idle run up to 30 %% slower)")
    ap.add_argument("--fe-variant", default="mfma", choices=["mfma", "valu"],
                    help="mfma (default): matrix-core kernels, whole mono chain fused; valu: the vector-ALU kernels "
                         "(front end + discriminator, then the audio kernel) -- the 'no MFMA' form of the north star")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="control-plane backend for N>1 (nccl = RCCL; gloo only to rehearse N>1 on a one-GPU box)")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()

    if args.fe_variant == "valu":
        os.environ["FMRX_FE_VARIANT"] = "valu"     # read per call by libfmrx
    import torch

    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; libfmrx has no CPU fallback"}))
        return 2
    dev_index = 0 if args.all_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev_index)
    rank, local_rank, world, dist = init_ranks(args.dist_backend)
    local_rank = dev_index

    fmrx = importlib.import_module("software-defined-radio_amd")
    synth = importlib.import_module("software-defined-radio_amd.synth")

    # ---- device-resident synthetic stream: this rank's channel ----
    B = args.blocks
    base_blocks = 4 if B % 4 == 0 else 1
    iq_host = synth.synth_fm_u8(BLOCK_SAMPLES * base_blocks, 2.4e6, seed=channel_seed(rank))
    d_iq = torch.from_numpy(iq_host).cuda().repeat(B // base_blocks)       # [2 * B * 1,024,000] u8
    n_bytes = d_iq.numel()
    n_samples = n_bytes // 2
    pl = fmrx.Pipeline(0, 1, rf_taps=101, base_audio_taps=101, max_block_bytes=n_bytes, device=local_rank)
    n_audio = pl.n_audio(n_bytes)
    d_audio = torch.empty(n_audio, dtype=torch.float32, device="cuda")
    d_pcm = torch.empty(n_audio, dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        pl.process_dev(d_iq.data_ptr(), n_bytes, d_audio.data_ptr(), d_pcm.data_ptr(), wrap=True, stream=stream)

    # untimed: bring the device from idle to the clocks it sustains under this load, then W warm-up steps
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    # HIP events around the dominant kernel, on the launch stream, every PROF_EVERY-th step of the timed
    # region (an event record costs ~5 us of stream time, 4 % of a step)
    pl.set_profiling(PROF_EVERY)
    elapsed = timed_region(step, torch.cuda.synchronize, args.steps, 0, dist,
                           device="cuda" if args.dist_backend == "nccl" else "cpu")

    tsum, cnt = pl.timing_sum((args.steps + PROF_EVERY - 1) // PROF_EVERY)
    # The fused kernel is the whole step: the handle's other two event pairs then bracket nothing and
    # measure what a pair of event records itself takes on the stream (~4-5 us); the kernel's launch
    # duration is the bracketed interval minus that (it then agrees with rocprofv3's kernel trace).
    # (with --fe-variant valu the audio kernel sits in the second pair; the third is still empty.)
    pair_ms = min(tsum["audio_ms"], tsum["rest_ms"]) / cnt
    fused = args.fe_variant == "mfma" and tsum["audio_ms"] / cnt < 0.02
    fe_ms = tsum["front_end_ms"] / cnt - pair_ms
    pl.set_profiling(False)

    # ---- side legs, after the timed region, N=1 only: (1) the same workload through the vector-ALU
    #      kernels (the north star's "no MFMA" form), (2) what a plain device-to-device copy reaches on
    #      this box (report the fraction of measured copy bandwidth next to the 8 TB/s peak) ----
    north_star_form = copy_gbs = None
    if world == 1 and args.fe_variant == "mfma":
        os.environ["FMRX_FE_VARIANT"] = "valu"
        for _ in range(200):
            step()
        torch.cuda.synchronize()
        pl.set_profiling(PROF_EVERY)
        k2 = max(20, min(args.steps, 50))
        t0 = time.perf_counter()
        for _ in range(k2):
            step()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t0
        t2, c2 = pl.timing_sum((k2 + PROF_EVERY - 1) // PROF_EVERY)
        pl.set_profiling(False)
        os.environ.pop("FMRX_FE_VARIANT")
        fe2 = (t2["front_end_ms"] - t2["rest_ms"]) / c2
        north_star_form = {
            "fe_variant": "valu", "value": round(n_samples * k2 / dt2 / 1e6, 1), "unit": "MS/s",
            "kernel": "fe_demod_kernel<101,10,8> (v_pk_fma_f32 FIR + discriminator, S2: 2.4 B/sample) then audio_fir_kernel",
            "avg_launch_ms": round(fe2, 4), "achieved": round(2.4 * n_samples / (fe2 * 1e-3) / 1e9, 1),
            "frac": round(2.4 * n_samples / (fe2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": k2,
        }
    if world == 1:
        src = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
        dst = torch.empty_like(src)
        for _ in range(5):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 2 * src.numel() * 4 * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e9     # bytes read + written
        del src, dst

    out = None
    if rank == 0:
        value = job_throughput(world, n_samples, args.steps, elapsed)
        # dominant kernel: the fused mono kernel (S3: 2.12 B/sample) or, for --fe-variant valu, the
        # vector-ALU front end + discriminator (S2: 2 B in + 4/10 B of f32 demod out)
        bytes_per_sample = FE_BYTES_PER_SAMPLE if fused else 2.0 + 4.0 / 10.0
        fe_bytes = bytes_per_sample * n_samples
        achieved = fe_bytes / (fe_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "fe_traffic.json")   # PMC-derived bytes per launch, if collected
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if int(tj.get("blocks", -1)) == B:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "I/Q MS/s through front-end FIR+decimate+demod(+audio) per job; % HBM roofline",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8+f32" if fused else "f32", "data": "synthetic",
            "config": {
                "workload": "configs[1]: mode 0 mono, 101-tap FE FIR+decimate(10) + FM discriminator + 101-tap audio "
                            "FIR+decimate(5) + s16 pack; synthetic 2.4 MS/s FM I/Q (u8), 1,024,000-sample blocks",
                "blocks_per_step": B, "samples_per_step_per_gpu": n_samples,
                "sharding": f"{world} independent channel(s), one per GPU, no collective",
                "realtime_channels_equiv": round(value / 2.4, 0), "settle_ms": args.settle_ms,
            },
            "roofline": {
                "kernel": ("mono_fused_kernel<101,10,101,5> (u8 I/Q -> 101-tap FIR, decimate 10 (int8 MFMA) -> FM "
                           "discriminator -> 101-tap audio FIR, decimate 5 (f32 MFMA) -> f32 audio + s16 PCM)") if fused else
                          "fe_demod_kernel<101,10,8> (u8 I/Q -> 101-tap FIR, decimate 10 (v_pk_fma_f32) -> FM discriminator -> f32 demod)",
                "algorithmic_bytes_per_sample": round(bytes_per_sample, 2), "fe_variant": args.fe_variant,
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic if fused else None,
                "algorithmic_bytes_per_launch": int(fe_bytes), "avg_launch_ms": round(fe_ms, 4),
                "launches_timed": cnt, "event_pair_overhead_ms": round(pair_ms, 4),
                "useful_tflops": round(FE_FLOP_PER_SAMPLE * n_samples / (fe_ms * 1e-3) / 1e12, 2),
                "stage_ms": {k: round(v / cnt, 4) for k, v in tsum.items()},
                "measured_copy": None if copy_gbs is None else {
                    "what": "torch d2d copy of 512 MiB f32, bytes read + written per second", "GB/s": round(copy_gbs, 1),
                    "frac_of_copy": round(achieved / copy_gbs, 4)},
            },
        }
        if north_star_form is not None:
            out["north_star_form"] = north_star_form
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
