#!/usr/bin/env python3
"""bench.py -- headline benchmark of the FM-receiver hot path on MI355X.

Metric (BASELINE.json): complex I/Q mega-samples per second through the receiver chain, per job (all
GPUs), and the dominant kernel's fraction of the HBM roofline.

Workload at N=1 = BASELINE.json configs[1]: mode 0 mono, 101-tap front-end FIR + decimate(10) + FM
discriminator + 101-tap audio FIR + decimate(5) + s16 pack (the reference's output format,
src/threadMonoOnly.cpp:185-191), synthetic 2.4 MS/s FM I/Q, blocks of 1,024,000 complex samples.  One
"step" = one pass of the whole chain over a device-resident batch of `--blocks` (default 1024)
consecutive blocks of ONE stream (2.1 GB of u8 I/Q: 8 x the 256 MiB Infinity Cache), submitted as one block-parallel call: an offline,
one-long-stream figure (a live 2.4 MS/s channel delivers 51,200-sample blocks; the per-block latency of
that regime is reported separately as `small_block`).  The mono chain is a sliding-window map of its
input, so the block-parallel call equals block-by-block streaming
(tests/test_gpu_parity.py::test_fused_mono_kernel, ::test_block_split_invariance_on_device).  Inputs are
resident in HBM when the timed region starts; the s16 PCM is written to HBM.

N>1: one process per GPU, one independent channel per GPU (seed + rank), no data-path collective: weak
scaling.  `python bench.py --gpus N` starts the N ranks itself (child processes, control plane over
gloo: the north star leaves RCCL unused); under `torch.distributed.run` (RANK / WORLD_SIZE set) it is
one of the ranks.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

BLOCK_SAMPLES = 1_024_000          # complex samples per block (= 20 reference blocks)
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# Algorithmic HBM bytes per complex input sample:
S3_BYTES = 2.0 + 2.0 / 50.0        # whole mono chain fused, s16 out: 2 B of u8 I/Q in + 2 B per rf_decim*audio_decim = 50 samples
S2_BYTES = 2.0 + 4.0 / 10.0        # front end + discriminator to HBM (f32 demod out)
S1_BYTES = 2.0 + 8.0 / 10.0        # front end only: f32 IF I,Q out
FE_FLOP_PER_SAMPLE = 2 * 2 * 101 / 10.0 + 9 / 10.0 + 2 * 101 / 50.0     # useful FIR + discriminator flops


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = the reference's algorithm restated; checker library, this leg only)
# ------------------------------------------------------------------------------------------------
def cpu_baseline(seconds: float = 10.0) -> dict:
    """oracle/fm_oracle.c (-O3 like src/Makefile:4) timed on this host: same chain, same taps, same
    signal.  (i) single thread = the figure to quote; (ii) the reference's own 2-thread shape
    (src/project.cpp:470-496: front-end thread -> 6-deep queue -> audio thread); (iii) one independent
    channel per available core."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np
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

    def run_two_threads(budget_s: float) -> tuple[int, float]:
        h_rf, h_au = o.impulse_response_lpf(2.4e6, 100e3, 101), o.impulse_response_lpf(240e3, 16e3, 101)
        q: queue.Queue = queue.Queue(maxsize=6)           # QUEUE_ELEMS, include/dy4.h:30
        done = [0]
        t0 = time.perf_counter()

        def front_end():
            si = sq = np.zeros(100, np.float32)
            pi = pq = 0.0
            while time.perf_counter() - t0 < budget_s:
                for b in range(nblk_ref):
                    f = o.u8_to_f32(iq[b * 102400:(b + 1) * 102400])
                    fi, si = o.convolve_block_fast_fir(f[0::2], h_rf, si, 10)
                    fq, sq = o.convolve_block_fast_fir(f[1::2], h_rf, sq, 10)
                    d, pi, pq = o.fm_demod(fi, fq, pi, pq)
                    q.put(d)
            q.put(None)

        def audio():
            st = np.zeros(100, np.float32)
            while True:
                d = q.get()
                if d is None:
                    return
                y, st = o.convolve_block_fast_fir(d, h_au, st, 5)
                o.pcm16(y)
                done[0] += 51200

        ta, tb = threading.Thread(target=front_end), threading.Thread(target=audio)
        ta.start(); tb.start(); ta.join(); tb.join()
        return done[0], time.perf_counter() - t0

    def run_mode(mode: int, channels: int, budget_s: float) -> dict:
        """single thread, another mode / the stereo chain: the figure that stands beside that mode's GPU leg"""
        p = o.mode_params(mode, 101, 101, 101)
        nb = 20
        data = synth.synth_fm_u8(p.block_bytes // 2 * nb, float(p.rf_Fs), seed=0x3D74 + 10 + mode)
        pl = o.pipeline(mode, channels)
        t0 = time.perf_counter()
        done = 0
        while True:
            for b in range(nb):
                pl.process(data[b * p.block_bytes:(b + 1) * p.block_bytes])
            done += p.block_bytes // 2 * nb
            if time.perf_counter() - t0 >= budget_s:
                break
        dt = time.perf_counter() - t0
        return {"value": round(done / dt / 1e6, 2), "unit": "MS/s", "cores": 1, "kind": "port",
                "sample": f"mode {mode} {'stereo' if channels == 2 else 'mono'} chain (101/101/101 taps), {done} complex samples in {dt:.1f} s, "
                          "single thread, reference-size blocks"}

    n1, t1 = run(seconds)
    n2, t2 = run_two_threads(seconds * 0.3)
    per_mode = {"mode1_mono": run_mode(1, 1, seconds * 0.3), "mode2_mono": run_mode(2, 1, seconds * 0.3),
                "mode3_mono": run_mode(3, 1, seconds * 0.3), "mode0_stereo": run_mode(0, 2, seconds * 0.3),
                "mode1_stereo": run_mode(1, 2, seconds * 0.2), "mode2_stereo": run_mode(2, 2, seconds * 0.2)}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    with ThreadPoolExecutor(cores) as ex:                # ctypes releases the GIL inside the C call
        t0 = time.perf_counter()
        res = list(ex.map(lambda c: run(seconds * 0.5, c + 1), range(cores)))
        wall = time.perf_counter() - t0
    return {
        "value": round(n1 / t1 / 1e6, 2), "unit": "MS/s", "cores": 1, "kind": "port",
        "sample": f"mode-0 mono chain (101/101 taps), {n1} complex samples in {t1:.1f} s, single thread, "
                  "oracle/fm_oracle.c built -O3 (reference flags)",
        "two_thread_pipeline": {"value": round(n2 / t2 / 1e6, 2), "unit": "MS/s", "cores": 2,
                                "sample": f"front-end thread -> 6-deep queue -> audio thread (src/project.cpp:470-496), {t2:.1f} s; "
                                          "stage calls through ctypes/numpy, so an upper bound on the reference binary's overhead"},
        "all_cores": {"value": round(sum(r[0] for r in res) / wall / 1e6, 2), "unit": "MS/s", "cores": cores,
                      "sample": f"{cores} independent channels, one per core, {wall:.1f} s"},
        "legs": per_mode,
    }


# ------------------------------------------------------------------------------------------------
# ranks
# ------------------------------------------------------------------------------------------------
def init_ranks(backend: str | None):
    """One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Returns
    (rank, local_rank, world, dist-or-None).  The process group is CONTROL PLANE only (barrier + max of
    the elapsed time): channels never exchange data."""
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


def timed_region(step, sync, steps: int, warmup: int, dist, device=None, on_start=None, on_stop=None) -> float:
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync on both
    sides; returns the MAX over ranks of the elapsed seconds.  on_start / on_stop: hooks inside the
    bracket (device events on the launch stream)."""
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
    if on_start:
        on_start()
    for _ in range(steps):
        step()
    if on_stop:
        on_stop()
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


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv: list[str]) -> int:
    """`bench.py --gpus N` without a launcher: start N child processes of this script, one rank per GPU
    (LOCAL_RANK = device ordinal), rendezvous on 127.0.0.1, and relay rank 0's JSON line.  The parent
    never touches the GPU (children are started, not exec'ed into)."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = next((l for l in reversed((out0 or "").splitlines()) if l.startswith("{")), None)
    if line:
        print(line, flush=True)
    if any(rcs) or not line:
        sys.stderr.write(f"bench.py: ranks exited with {rcs}\n")
        return next((rc for rc in rcs if rc), 1)
    return 0


# ------------------------------------------------------------------------------------------------
# device-side measurement helpers
# ------------------------------------------------------------------------------------------------
def event_ms(torch, step, k: int, warm: int = 3, drain: bool = True) -> float:
    """Average device time of one step: HIP events on the launch stream (torch's current stream, the one
    every step is submitted to) around k back-to-back steps.  Includes the dispatch gap between
    consecutive kernels (1-2 us), subtracts nothing.  drain = False: the timed steps follow the warm-up steps
    without a device-wide wait in between (the few-step stereo legs: a software pipeline over consecutive calls
    would otherwise be timed with its fill)."""
    for _ in range(warm):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if drain:
        torch.cuda.synchronize()
    e0.record()
    for _ in range(k):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


def leg(name, what, n_samples, ms, bytes_per_sample, extra=None):
    gbs = bytes_per_sample * n_samples / (ms * 1e-3) / 1e9
    d = {"what": what, "samples_per_step": n_samples, "ms_per_step": round(ms, 4), "value": round(n_samples / (ms * 1e-3) / 1e6, 1),
         "unit": "MS/s", "algorithmic_bytes_per_sample": round(bytes_per_sample, 3), "achieved_GBs": round(gbs, 1),
         "frac": round(gbs / HBM_PEAK_GBS, 4)}
    if extra:
        d.update(extra)
    return name, d


def side_legs(torch, fmrx, synth, args, pl, step, d_iq, n_bytes, stream) -> dict:
    """N=1 only, after the timed region: the other configurations and kernels of the path, each with its
    algorithmic bytes per sample, device time per step and fraction of the HBM peak."""
    import numpy as np
    legs = {}
    n_samples = n_bytes // 2
    k = max(10, min(args.steps, 30))
    # (0) the same step with 256 blocks resident (0.5 GB: BASELINE's minimum, the headline configuration of rounds 1 and 2): a shorter
    #     call amortises the kernel's ramp-up and tail over fewer tiles per wave (tools/working_set_probe.py: 0.53-0.60 at 256 blocks,
    #     0.64-0.65 at 1024-2048, pure streaming reads flat at 6.1-6.5 TB/s: no Infinity Cache effect either way).  Measured first,
    #     in the thermal state of the headline (the vector-ALU legs below pull the clocks down for a while)
    try:
        nb256 = 256 * BLOCK_SAMPLES * 2
        if n_bytes >= nb256:
            q = fmrx.Pipeline(0, 1, rf_taps=101, base_audio_taps=101, max_block_bytes=nb256, device=torch.cuda.current_device())
            d_pcm_s = torch.empty(q.n_audio(nb256), dtype=torch.int16, device="cuda")
            ms = event_ms(torch, lambda: q.process_dev(d_iq.data_ptr(), nb256, None, d_pcm_s.data_ptr(), wrap=True, stream=stream), 30, warm=200)
            name, d = leg("mono_256_blocks", "the headline step with 256 x 1,024,000-sample blocks (0.5 GB) resident per step: "
                          "mono_fused_kernel<101,10,101,5>, s16 out", nb256 // 2, ms, S3_BYTES)
            legs[name] = d
            q.close()
            del q, d_pcm_s
    except Exception as e:
        legs["mono_256_blocks"] = {"error": str(e)}
    # (1) the same workload through the vector-ALU kernels: the north star's "no MFMA" form (S2 + audio kernel)
    pl.set_option("fe_variant", "valu")
    ms = event_ms(torch, step, k, warm=100)
    pl.set_profiling(4)
    for _ in range(16):
        step()
    torch.cuda.synchronize()
    t4, c4 = pl.timing_sum(4)
    pl.set_profiling(False)
    fe_ms = max(ms - t4["audio_ms"] / c4, 1e-6)
    legs["north_star_form"] = {
        "fe_variant": "valu", "value": round(n_samples / (ms * 1e-3) / 1e6, 1), "unit": "MS/s", "ms_per_step": round(ms, 4),
        "kernel": "fe_demod_kernel<101,10,8> (v_pk_fma_f32 FIR + discriminator, S2: 2.4 B/sample) then audio_fir_kernel",
        "avg_launch_ms": round(fe_ms, 4), "achieved": round(S2_BYTES * n_samples / (fe_ms * 1e-3) / 1e9, 1),
        "frac": round(S2_BYTES * n_samples / (fe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "steps": k,
        "note": "avg_launch_ms = step time minus the audio kernel's event-bracketed time"}
    pl.set_option("fe_variant", "mfma")
    # (2) S2: matrix-core front end + discriminator to HBM, then the audio kernel (the two-kernel path)
    pl.set_option("fused_min_audio", 10**12)
    ms = event_ms(torch, step, k)
    name, d = leg("two_kernel_path", "fe_mfma_kernel<101,10> (S2, f32 demod to HBM) + audio_fir_kernel, mode 0 mono", n_samples, ms,
                  S2_BYTES + 4.0 / 10.0 + 2.0 / 50.0)
    legs[name] = d
    pl.set_option("fused_min_audio", 65536)
    # (3) S1: the north star's literal kernel, IF only (u8 I/Q -> f32 IF I,Q)
    h = fmrx.impulseResponseLPF(2.4e6, 100e3, 101)
    plan = fmrx.FrontEndPlan(h, 10)
    d_if = torch.empty(2 * (n_samples // 10), dtype=torch.float32, device="cuda")
    for var in ("mfma", "valu"):
        fmrx.set_option("fe_variant", var)
        ms = event_ms(torch, lambda: plan.run_dev(d_iq.data_ptr(), n_samples, None, d_if.data_ptr(), stream=stream), k,
                      warm=100 if var == "valu" else 3)
        name, d = leg(f"s1_if_only_{var}", f"front-end FIR + decimate only (S1), {var} kernel, fmrx_fe_run_dev", n_samples, ms, S1_BYTES)
        legs[name] = d
    fmrx.set_option("fe_variant", "mfma")
    del d_if, plan
    # (4) the other modes / stereo, streaming (state carried), device-resident
    def mode_leg(name, mode, channels, blocks, what, bytes_per_sample, bytes_unit, base_blocks=1, options=None):
        # base_blocks blocks are synthesised and repeated; stereo needs a seamless stream (a splice is a pilot phase
        # jump: the PLL would un-lock and be repaired): 3 x 1,024,000 samples = 1280 periods of the 1 ms multiplex
        p = fmrx.modeParams(mode)
        nb = blocks * bytes_unit
        iq = synth.synth_fm_u8(base_blocks * bytes_unit // 2, float(p.rf_Fs), seed=0x3D74 + 10 + mode)
        d_in = torch.from_numpy(iq).cuda().repeat(blocks // base_blocks)
        q = fmrx.Pipeline(mode, channels, max_block_bytes=nb, device=torch.cuda.current_device())
        for key, val in (options or {}).items():
            q.set_option(key, val)
        na = q.n_audio(nb)
        d_pcm = torch.empty(channels * na, dtype=torch.int16, device="cuda")
        fn = lambda: q.process_dev(d_in.data_ptr(), nb, None, d_pcm.data_ptr(), wrap=True, stream=stream)
        if channels == 2:
            # the reference's float32 trigArg is down to a resolution of 0.5 rad 8.4 M IF samples (35 s) into a stream, where the PLL
            # lanes fall back to long warm-ups, and its trigOffset stops counting at 2^24 (70 s): the whole
            # measurement (warm-up + timed steps) is a fresh stream's first 31 s
            q.reset()
            n_if = nb // 2 // int(p.rf_decim)
            warm = 2 if 6 * n_if < 7_500_000 else 1
            ms = event_ms(torch, fn, max(2, min(8, 7_500_000 // n_if - warm)), warm=warm, drain=False)
        else:
            ms = event_ms(torch, fn, k, warm=5)
        nm, d = leg(name, what, nb // 2, ms, bytes_per_sample)
        if channels == 2:
            d["pll_repaired_segments"] = q.pll_diagnostics()[0]
            d["tolerance"] = TOL_FAST
        legs[nm] = d
        q.close()

    TOL_FAST = ("audio RMS error vs the reference <= 1e-4 for the first 0.13 s of a stream, <= 0.06 ulp(trigArg(t)) after (2.2e-4 at 1 s, "
                "5.5e-4 at 2.1 s: the reference's own float32 phase grid, DESIGN.md section 2); mono sum (L+R)/2 <= 2e-6 throughout")
    TOL_EXACT = "bit-exact: left / right equal the compiled reference's for any stream length (SHA-256 over 2.13 s, tests/golden/stereo_long_mode0.npz)"
    mode_leg("mode1_mono", 1, 1, 256, "mode 1 mono (1.44 MS/s, decim 5 x 6), fused kernel, 256 blocks of 614,400 samples", 2.0 + 2.0 / 30.0, 1228800)
    mode_leg("mode2_mono", 2, 1, 63, "mode 2 mono (U/D = 147/800 resampler), 63 blocks of 1,008,000 samples", 2.0 + 2.0 * 147 / 8000.0, 2016000)
    mode_leg("mode3_mono", 3, 1, 63, "mode 3 mono (960 kS/s, U/D = 441/3200), 63 blocks of 1,008,000 samples", 2.0 + 2.0 * 441 / 9600.0, 2016000)
    mode_leg("mode0_stereo", 0, 2, 12, "mode 0 stereo (pilot PLL + 38 kHz mixer + L/R), s16 L,R out, 12 x 1,024,000-sample blocks per step, stream continued (its first 31 s)",
             2.0 + 4.0 / 50.0, 2048000, base_blocks=3)
    mode_leg("mode0_stereo_24_blocks", 0, 2, 24, "mode 0 stereo, 24 x 1,024,000-sample blocks per step (the PLL's 128-step lanes are a fixed cost per "
             "call up to ~40 blocks), stream continued", 2.0 + 4.0 / 50.0, 2048000, base_blocks=3)
    # the same with option overlap_calls = 1: the caller vouches that a call's input is complete when the call is made (it is:
    # resident in HBM), and the next call's front end + band-pass pair + PLL chunk records run on an internal stream under
    # this call's PLL lanes and output stage; same kernels, same results (tests: bit-identical PCM and state)
    mode_leg("mode0_stereo_overlapped", 0, 2, 12, "mode 0 stereo, 12 x 1,024,000-sample blocks per step, option overlap_calls = 1 (consecutive calls "
             "software-pipelined over two streams), stream continued", 2.0 + 4.0 / 50.0, 2048000, base_blocks=3, options={"overlap_calls": 1})
    mode_leg("mode0_stereo_18_blocks_overlapped", 0, 2, 18, "mode 0 stereo, 18 x 1,024,000-sample blocks per step, option overlap_calls = 1, stream continued",
             2.0 + 4.0 / 50.0, 2048000, base_blocks=3, options={"overlap_calls": 1})
    # (4b) the CONFORMING stereo paths (within the north star's 1e-4 for any stream length = bit-exact):
    #      single stream: the bit-exact mode (every stage in the reference's order, fmPLL walked by one lane)
    try:
        nb = 2048000
        p = fmrx.modeParams(0)
        iq = synth.synth_fm_u8(nb // 2, float(p.rf_Fs), seed=0x3D74 + 10)
        d_in = torch.from_numpy(iq).cuda()
        q = fmrx.Pipeline(0, 2, max_block_bytes=nb, device=torch.cuda.current_device())
        q.set_force_generic(True)
        d_pcm = torch.empty(2 * q.n_audio(nb), dtype=torch.int16, device="cuda")
        ms = event_ms(torch, lambda: q.process_dev(d_in.data_ptr(), nb, None, d_pcm.data_ptr(), wrap=True, stream=stream), 2, warm=1)
        nm, d = leg("mode0_stereo_exact", "mode 0 stereo, ONE stream, bit-exact mode (set_force_generic / CLI --exact): reference evaluation order "
                    "everywhere, fmPLL's recurrence walked by one lane; one 1,024,000-sample block per step", nb // 2, ms, 2.0 + 4.0 / 50.0)
        d["tolerance"] = TOL_EXACT
        d["x_real_time"] = round(nb / 2 / 2.4e6 / (ms * 1e-3), 2)
        legs[nm] = d
        q.close()
        del d_in, d_pcm
    except Exception as e:
        legs["mode0_stereo_exact"] = {"error": str(e)}
    #      many streams: the receiver bank, one lane per channel walks the exact recurrence (fmrx_channels_create_ex, exact = 1)
    def bank_leg(name, mode, nch, blocks_per_call, calls, exact=True, audio_channels=2):
        p = fmrx.modeParams(mode)
        bb = int(p.block_bytes) * blocks_per_call
        ns = bb // 2
        distinct = 64                                          # distinct signals dealt round-robin over the channels: every lane of a wave differs
        base = torch.stack([torch.from_numpy(synth.synth_fm_u8(ns, float(p.rf_Fs), seed=0x3D74 + c, start=7919 * c)) for c in range(distinct)]).cuda()
        chs = fmrx.Channels(mode, nch, audio_channels=audio_channels, exact=exact, block_bytes=bb, device=torch.cuda.current_device())
        src = base.repeat((nch + distinct - 1) // distinct, 1)[:nch].contiguous()
        chs.load_dev(src.data_ptr(), stream)
        del src, base
        d_pcm_all = torch.empty(nch * chs.n_audio * audio_channels, dtype=torch.int16, device="cuda")
        ms = event_ms(torch, lambda: chs.process_dev(None, d_pcm_all.data_ptr(), wrap=True, stream=stream), calls, warm=2)
        how = ("fmrx_channels_create_ex(exact = 1): reference evaluation order in every stage, fmPLL one lane per channel with glibc's sinf/cosf/atan2f"
               if exact else
               "fmrx_channels_create_ex(exact = 0): matrix-core front end, one fma per tap in the band-pass pair and the audio FIRs, the PLL's fast "
               "recurrence one lane per channel, three internal streams")
        out_rate = (p.audio_upsamp / float(p.audio_decim) if p.audio_upsamp else 1.0 / p.audio_decim) / p.rf_decim   # audio samples per input sample
        if audio_channels == 1 and not exact:
            how = "fmrx_channels_create_ex(audio_channels = 1, exact = 0), resampling mode: matrix-core front end + the batched lane-per-channel resampler"
        nm, d = leg(name, f"{nch} independent mode-{mode} {'STEREO' if audio_channels == 2 else 'MONO'} receivers, {ns:,} samples ({blocks_per_call} reference "
                    f"block(s)) each per call, {how}; s16 {'L,R ' if audio_channels == 2 else ''}out; inputs resident in HBM, 64 distinct signals dealt over "
                    f"the channels", nch * ns, ms, 2.0 + 2.0 * audio_channels * out_rate)
        d["tolerance"] = TOL_EXACT if exact else (TOL_FAST if audio_channels == 2 else "audio RMS error vs the reference <= 2e-6 (north star: 1e-4)")
        d["channels"] = nch
        d["channels_at_real_time"] = int(nch * (ns / float(p.rf_Fs)) / (ms * 1e-3))
        d["bound"] = (("vector ALU, not HBM: the reference's order is 2 separately rounded vector operations per tap and output (nothing for the matrix "
                       "cores), ~83 lane-instructions per input sample in total; frac is reported on the HBM peak for comparability only") if exact else
                      ("HBM traffic of the float32 intermediates between its five kernels (5.5 B per input sample against 2.08 algorithmic) and the "
                       "vector ALUs of the band-pass pair; DESIGN.md 4.7") if audio_channels == 2 else
                      ("front end: HBM (2.4 B per input sample with the discriminator's round trip); resampler: re-reading its overlapping windows; "
                       "DESIGN.md 4.7"))
        legs[nm] = d
        chs.close()
        del chs, d_pcm_all
        torch.cuda.empty_cache()
    try:
        bank_leg("stereo_channels", 0, 16384, 4, 5, exact=False)   # first: the exact banks' vector-ALU load pulls the clocks down for a while
        bank_leg("stereo_channels_exact", 0, 16384, 4, 3)
        bank_leg("stereo_channels_exact_65536", 0, 65536, 1, 3)
        bank_leg("stereo_channels_exact_mode1", 1, 16384, 4, 3)
        bank_leg("stereo_channels_exact_mode2", 2, 16384, 4, 3)        # 44.1 kHz out: the batched reference-order resampler behind the PLL
        bank_leg("mono_channels_mode2", 2, 16384, 4, 3, exact=False, audio_channels=1)
    except Exception as e:
        legs["stereo_channels_exact_error"] = {"error": str(e)}
    # (5) a live channel's regime: reference-size blocks (51,200 samples), one call per block, device-resident
    q = fmrx.Pipeline(0, 1, device=torch.cuda.current_device())
    d_pcm = torch.empty(1024, dtype=torch.int16, device="cuda")
    ms = event_ms(torch, lambda: q.process_dev(d_iq.data_ptr(), 102400, None, d_pcm.data_ptr(), wrap=True, stream=stream), 200, warm=20)
    legs["small_block"] = {"what": "mode 0 mono, one 51,200-sample reference block per call (what a live 2.4 MS/s channel delivers)",
                           "us_per_block": round(ms * 1e3, 2), "x_real_time": round(51200 / 2.4e6 / (ms * 1e-3), 1)}
    q.close()
    legs["small_block"]["fused_kernel_us_per_block"] = None
    q = fmrx.Pipeline(0, 1, device=torch.cuda.current_device())
    q.set_option("fused_min_audio", 0)
    ms = event_ms(torch, lambda: q.process_dev(d_iq.data_ptr(), 102400, None, d_pcm.data_ptr(), wrap=True, stream=stream), 200, warm=20)
    legs["small_block"]["fused_kernel_us_per_block"] = round(ms * 1e3, 2)
    q.close()
    # (5a) PCIe-inclusive figures (never `value`): host buffers through fmrx_pipeline_submit / _wait (two blocks in flight, page-locked
    #      memory), 1,024,000-sample blocks; and the drop-in CLI, stdin -> stdout, from and to /dev/shm
    try:   # a process of its own (started, not exec'ed): this one has created dozens of streams by now, and HIP deals streams
           # round-robin onto a few hardware queues -- the probe's two streams must not share one
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pcie_probe.py"), "--json"], capture_output=True, text=True, timeout=300)
        pj = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        two, one = pj["1024000_samples_two_in_flight"], pj["1024000_samples_synchronous"]
        legs["host_buffers"] = {
            "what": "mode 0 mono, 1,024,000-sample blocks from page-locked HOST memory, s16 back to host: fmrx_pipeline_submit / _wait "
                    "(two blocks in flight: a block's H2D under its neighbour's kernels and D2H); PCIe-inclusive, wall clock, own process "
                    "(tools/pcie_probe.py)",
            "ms_per_block": two["ms_per_block"], "value": two["MS_per_s"], "unit": "MS/s", "h2d_GBs": two["h2d_GBs"],
            "synchronous_process_ms_per_block": one["ms_per_block"], "synchronous_h2d_GBs": one["h2d_GBs"],
            "bare_pinned_h2d_2MB_GBs": pj.get("bare_pinned_h2d_2048000_bytes_GBs"), "larger_blocks": {k: v for k, v in pj.items() if k.startswith(("4096000", "16384000"))}}
    except Exception as e:
        legs["host_buffers"] = {"error": str(e)}
    try:
        cli = os.path.join(ROOT, "software-defined-radio_amd", "lib", "fmrx_project")
        path = "/dev/shm/fmrx_bench_iq.raw"
        nbytes = 0
        with open(path, "wb") as f:
            blk = d_iq[:2 * BLOCK_SAMPLES * 4].cpu().numpy().tobytes()
            for _ in range(64):
                f.write(blk); nbytes += len(blk)
        t0 = time.perf_counter()
        r = subprocess.run(f"{cli} 0 1 --blocks-per-call 20 < {path} > /dev/shm/fmrx_bench_pcm.raw", shell=True, capture_output=True)
        dt = time.perf_counter() - t0
        out_bytes = os.path.getsize("/dev/shm/fmrx_bench_pcm.raw")
        legs["cli_stdin_stdout"] = {
            "what": "the drop-in CLI: fmrx_project 0 1 --blocks-per-call 20 < /dev/shm/iq.raw > /dev/shm/pcm.raw (process start, device "
                    "initialisation and pipe I/O included), mode 0 mono", "input_bytes": nbytes, "output_bytes": out_bytes, "seconds": round(dt, 3),
            "value": round(nbytes / 2 / dt / 1e6, 1), "unit": "MS/s", "rc": r.returncode}
        os.remove(path); os.remove("/dev/shm/fmrx_bench_pcm.raw")
    except Exception as e:
        legs["cli_stdin_stdout"] = {"error": str(e)}
    # (5b) the same regime done properly: the current reference-size block of N live channels in ONE launch
    #      (fmrx_channels_*; every channel's state is its last bytes, csrc/channels.hip)
    nch = 4096
    chs = fmrx.Channels(0, nch, device=torch.cuda.current_device())
    blk = torch.from_numpy(synth.synth_fm_u8(51200, 2.4e6, seed=0x3D74 + 77)).cuda()
    src = blk.repeat(nch)                                   # every channel gets (a copy of) the block
    chs.load_dev(src.data_ptr(), stream)
    d_pcm_all = torch.empty(nch * chs.n_audio, dtype=torch.int16, device="cuda")
    ms = event_ms(torch, lambda: chs.process_dev(None, d_pcm_all.data_ptr(), wrap=True, stream=stream), 20, warm=5)
    legs["channels_batch"] = {
        "what": f"{nch} independent mode-0 mono channels, one 51,200-sample reference block each, per call (fmrx_channels_process_dev: one fused "
                "launch over all slots + one finishing kernel; inputs resident in HBM)",
        "ms_per_call": round(ms, 4), "value": round(nch * 51200 / (ms * 1e-3) / 1e6, 1), "unit": "MS/s",
        "us_per_channel_block": round(ms * 1e3 / nch, 4),
        "channels_at_real_time": int(nch * (51200 / 2.4e6) / (ms * 1e-3)),
        "note": "channels_at_real_time = how many 2.4 MS/s channels one GPU keeps up with when their blocks are already in HBM (PCIe would cap "
                "a live feed at ~13,000 channels per GPU: 64 GB/s / 4.8 MB/s)"}
    del chs, src, d_pcm_all
    # (5c) the RDS path (float64 chain on the discriminator output + host bit recovery), the model's block size
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from rds_signal import rds_demod_signal
        xr, _ = rds_demod_signal(8 * 9600, 240e3, seed=3, chip_offset=66.0)
        rds = fmrx.Rds(0)
        rds.process(xr[:9600])
        t0 = time.perf_counter()
        for b in range(1, 8):
            rds.process(xr[b * 9600:(b + 1) * 9600])
        dt = (time.perf_counter() - t0) / 7
        legs["rds"] = {"what": "RDS path (model/fmMonoBlock.py:238-296): 9,600 discriminator samples (40 ms of signal) per call, float64 "
                               "kernels (channel / carrier band-pass, serial PLL, mixers, 247/960 resampler, RRC) + host CDR / frame sync; "
                               "host buffers, wall clock", "ms_per_block": round(dt * 1e3, 3), "x_real_time": round(0.04 / dt, 1)}
        rds.close()
    except Exception as e:  # a side leg must not take the headline down
        legs["rds"] = {"error": str(e)}
    # (6) what this box's memory system gives a pure streaming read, by the access methods the kernels use
    reads = {}
    for method, label in ((0, "global_load_dwordx4 non-temporal to registers"), (1, "LDS-DMA ring (the matrix-core kernels' method)"),
                          (65, "LDS-DMA ring, 192 KiB chunks dealt round-robin over the waves (best pattern found, tools/stream_patterns.py)")):
        try:
            ms = event_ms(torch, lambda: fmrx.diagStreamRead(d_iq.data_ptr(), n_bytes, method, stream), 20, warm=20)
            reads[label] = round(n_bytes / (ms * 1e-3) / 1e9, 1)
        except Exception as e:  # diagnostics only
            reads[label] = f"unavailable: {e}"
    src = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    ms = event_ms(torch, lambda: dst.copy_(src), 20, warm=5)
    reads["torch d2d copy of 512 MiB, bytes read + written"] = round(2 * src.numel() * 4 / (ms * 1e-3) / 1e9, 1)
    del src, dst
    legs["measured_streaming_GBs"] = reads
    return legs


def run_stub(args) -> int:
    """Test hook (tests/test_multiproc_cpu.py): the launcher, the rank environment and the timing /
    aggregation logic without a GPU -- a step is a sleep of stub_ms * (rank + 1)."""
    rank, local_rank, world, dist = init_ranks("gloo")
    if world != args.gpus:
        print(json.dumps({"error": f"WORLD_SIZE {world} != --gpus {args.gpus}"}))
        return 2
    elapsed = timed_region(lambda: time.sleep(args.stub_ms * 1e-3 * (rank + 1)), lambda: None, args.steps, args.warmup, dist)
    seeds = [channel_seed(rank)]
    if dist is not None:
        seeds = [None] * world
        dist.all_gather_object(seeds, channel_seed(rank))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": job_throughput(world, 1000, args.steps, elapsed), "unit": "MS/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "data": "stub",
                          "seeds": seeds, "local_rank": local_rank}), flush=True)
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=1024,
                    help="1,024,000-sample blocks resident per step (default 1024 = 2.1 GB of u8 I/Q, 8 x the 256 MiB Infinity Cache; BASELINE asks for >= 256)")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed steps run for this long before the warm-up, so that the clocks the chip holds under "
                         "this load are reached (a step is ~0.12 ms; the first ~300 after idle run up to 30 %% slower)")
    ap.add_argument("--fe-variant", default="mfma", choices=["mfma", "valu"],
                    help="mfma (default): matrix-core kernels, whole mono chain fused; valu: the vector-ALU kernels "
                         "(front end + discriminator, then the audio kernel) -- the 'no MFMA' form of the north star")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the N=1 side measurements (other modes, S1/S2 kernels, copy probe)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"],
                    help="control-plane backend for N>1 (barrier + max of the elapsed time only; gloo: RCCL stays unused as the north star says)")
    ap.add_argument("--all-on-device0", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--stub-ms", type=float, default=0.0, help=argparse.SUPPRESS)   # test hook: see run_stub
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        print(json.dumps({"error": "--gpus must be >= 1"}))
        return 2
    if world_env is None and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])            # before anything touches the GPU
    if world_env is not None and int(world_env) != args.gpus:
        print(json.dumps({"error": f"WORLD_SIZE={world_env} but --gpus {args.gpus}: launch with --nproc-per-node {args.gpus}"}))
        return 2
    if args.stub_ms > 0:
        return run_stub(args)

    import numpy as np  # noqa: F401
    import torch

    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; libfmrx has no CPU fallback"}))
        return 2
    dev_index = 0 if args.all_on_device0 else int(os.environ.get("LOCAL_RANK", "0"))
    if dev_index >= torch.cuda.device_count():
        print(json.dumps({"error": f"rank needs GPU {dev_index} but only {torch.cuda.device_count()} visible (--gpus {args.gpus})"}))
        return 3
    torch.cuda.set_device(dev_index)
    rank, local_rank, world, dist = init_ranks(args.dist_backend)
    local_rank = dev_index

    fmrx = importlib.import_module("software-defined-radio_amd")
    synth = importlib.import_module("software-defined-radio_amd.synth")
    stray = sorted(k for k in os.environ if k.startswith("FMRX_") and k != "FMRX_NO_TORCH")
    # which library is being timed: path, SHA-256 of the file, and the identity it reports itself (hash of its sources)
    import hashlib
    lib_version = fmrx.lib.fmrx_version().decode()
    lib_id = {"path": os.path.relpath(fmrx.LIB_PATH, ROOT) if fmrx.LIB_PATH.startswith(ROOT) else fmrx.LIB_PATH,
              "sha256": hashlib.sha256(open(fmrx.LIB_PATH, "rb").read()).hexdigest(), "version": lib_version,
              "FMRX_LIB": os.environ.get("FMRX_LIB")}
    lib_src = lib_version.split("src:")[-1] if "src:" in lib_version else None

    # ---- device-resident synthetic stream: this rank's channel ----
    B = args.blocks
    base_blocks = 4 if B % 4 == 0 else 1
    iq_host = synth.synth_fm_u8(BLOCK_SAMPLES * base_blocks, 2.4e6, seed=channel_seed(rank))
    d_iq = torch.from_numpy(iq_host).cuda().repeat(B // base_blocks)       # [2 * B * 1,024,000] u8
    n_bytes = d_iq.numel()
    n_samples = n_bytes // 2
    pl = fmrx.Pipeline(0, 1, rf_taps=101, base_audio_taps=101, max_block_bytes=n_bytes, device=local_rank)
    pl.set_option("fe_variant", args.fe_variant)
    n_audio = pl.n_audio(n_bytes)
    d_pcm = torch.empty(n_audio, dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        pl.process_dev(d_iq.data_ptr(), n_bytes, None, d_pcm.data_ptr(), wrap=True, stream=stream)

    # untimed: bring the device from idle to the clocks it sustains under this load, then W warm-up steps
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    # HIP events on the launch stream bracket exactly the K timed launches (recorded inside the barrier
    # bracket): average launch duration of the dominant kernel = bracket / K, dispatch gaps included
    ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    elapsed = timed_region(step, torch.cuda.synchronize, args.steps, args.warmup, dist,
                           device="cuda" if args.dist_backend == "nccl" else "cpu",
                           on_start=ev[0].record, on_stop=ev[1].record)
    fe_ms = ev[0].elapsed_time(ev[1]) / args.steps
    fused = args.fe_variant == "mfma"
    if not fused:   # vector-ALU variant: the step is two kernels; the front end's share from the handle's event brackets
        pl.set_profiling(4)
        for _ in range(16):
            step()
        torch.cuda.synchronize()
        t4, c4 = pl.timing_sum(4)
        pl.set_profiling(False)
        fe_ms = max(fe_ms - t4["audio_ms"] / c4, 1e-6)

    legs = {}
    if world == 1 and fused and not args.no_side_legs:
        legs = side_legs(torch, fmrx, synth, args, pl, step, d_iq, n_bytes, stream)

    out = None
    if rank == 0:
        value = job_throughput(world, n_samples, args.steps, elapsed)
        bytes_per_sample = S3_BYTES if fused else S2_BYTES
        fe_bytes = bytes_per_sample * n_samples
        achieved = fe_bytes / (fe_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "fe_traffic.json")   # PMC-derived bytes per launch of this command
        if fused and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if int(tj.get("blocks", -1)) == B and tj.get("output") == "s16":   # same workload as the counters were collected on
                    if tj.get("lib_src") and tj.get("lib_src") == lib_src:
                        traffic = tj.get("hbm_bytes_per_launch")
                        traffic_source = ("profiles/fe_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this "
                                          f"build (src:{lib_src}, {tj.get('round', 'earlier round')}); a committed constant, NOT measured in this run")
                    else:
                        traffic_source = (f"profiles/fe_traffic.json was collected on build src:{tj.get('lib_src')}, this library is src:{lib_src}: "
                                          "stale, not reported")
            except Exception:
                traffic = None
        out = {
            "metric": "I/Q MS/s through front-end FIR+decimate+demod(+audio) per job; % HBM roofline",
            "value": round(value, 1), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i8+f32" if fused else "f32", "data": "synthetic",
            "config": {
                "workload": "configs[1]: mode 0 mono, 101-tap FE FIR+decimate(10) + FM discriminator + 101-tap audio "
                            "FIR+decimate(5) + s16 pack; synthetic 2.4 MS/s FM I/Q (u8), 1,024,000-sample blocks; "
                            "offline: one long stream per GPU, all blocks of a step in one block-parallel call",
                "blocks_per_step": B, "samples_per_step_per_gpu": n_samples,
                "sharding": f"{world} independent channel(s), one per GPU, no collective (control plane: {args.dist_backend})",
                "settle_ms": args.settle_ms, "fmrx_env": stray, "library": lib_id,
            },
            "roofline": {
                "kernel": ("mono_fused_kernel<101,10,101,5> (u8 I/Q -> 101-tap FIR, decimate 10 (int8 MFMA) -> FM "
                           "discriminator -> 101-tap audio FIR, decimate 5 (f32 MFMA) -> s16 PCM)") if fused else
                          "fe_demod_kernel<101,10,8> (u8 I/Q -> 101-tap FIR, decimate 10 (v_pk_fma_f32) -> FM discriminator -> f32 demod)",
                "algorithmic_bytes_per_sample": round(bytes_per_sample, 3),
                "algorithmic_bytes_note": "SURVEY 8d S3 (whole mono chain, s16 out: 2 + 2/50)" if fused else "SURVEY 8d S2 (2 + 4/10)",
                "fe_variant": args.fe_variant,
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(fe_bytes), "avg_launch_ms": round(fe_ms, 4),
                "launches_timed": args.steps,
                "timing": "HIP events on the launch stream around the K timed launches / K (dispatch gaps included, nothing subtracted)",
                "useful_tflops": round(FE_FLOP_PER_SAMPLE * n_samples / (fe_ms * 1e-3) / 1e12, 2),
            },
        }
        if legs:
            out["north_star_form"] = legs.pop("north_star_form")
            reads = legs.pop("measured_streaming_GBs", None)
            if reads:
                out["roofline"]["measured_streaming_GBs"] = reads
                best = max((v for v in reads.values() if isinstance(v, (int, float))), default=None)
                if best:
                    out["roofline"]["frac_of_best_measured_streaming"] = round(achieved / best, 4)
            out["legs"] = legs
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
