#!/usr/bin/env python3
"""Throughput of every receiver configuration (SURVEY 8 rows / BASELINE configs),
device-resident input, one channel on one GPU.  Reports MS/s (complex input
samples per second) and per-stage device time.  Also the host-buffer (PCIe
inclusive) rate and the stdin->stdout CLI rate for mode 0.
    python tools/mode_bench.py [blocks_mono=64] [blocks_stereo=4]
"""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")

BM = int(sys.argv[1]) if len(sys.argv) > 1 else 64
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
res = []
for mode in range(4):
    for ch in (1, 2):
        p = fmrx.modeParams(mode)
        per_blk = {0: 20, 1: 20, 2: 18, 3: 15}[mode] * p.block_bytes          # ~1M-sample blocks aligned to the mode's rules
        nb = max((BM if ch == 1 else BS) // 3 * 3, 3)
        # 3 blocks are a whole number of the multiplex's 1 ms periods in every mode, so tiling them (and
        # wrapping from the end of a step to the start of the next) keeps the pilot phase continuous
        iq = torch.from_numpy(synth.synth_fm_u8(per_blk // 2 * 3, p.rf_Fs, seed=0x3D74 + mode)).cuda()
        iq = iq.repeat(nb // 3)
        n_bytes = iq.numel()
        pl = fmrx.Pipeline(mode, ch, max_block_bytes=n_bytes)
        na = pl.n_audio(n_bytes)
        d_a = torch.empty(ch * na, dtype=torch.float32, device="cuda")
        d_p = torch.empty(ch * na, dtype=torch.int16, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        steps = 10 if ch == 1 else 2
        pl.process_dev(iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        pl.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            if ch == 2:
                pl.reset()   # the reference's float32 trigOffset stalls at 2^24 IF samples (70 s): keep each step a fresh stream
            pl.process_dev(iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ts, cnt = pl.timing_sum(steps)
        r = {"mode": mode, "channels": ch, "samples_per_step": n_bytes // 2, "ms_per_step": round(dt * 1e3, 3),
             "MS_per_s": round(n_bytes / 2 / dt / 1e6, 1), "x_realtime": round(n_bytes / 2 / dt / p.rf_Fs, 1),
             "stage_ms": {k: round(v / cnt, 3) for k, v in ts.items()}}
        if ch == 2:
            # the same stream continued (no reset): later blocks start locked, no serial PLL head; 3 more
            # steps keep the stream below the 2^24 IF samples where the reference's float32 trigOffset stalls
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                pl.process_dev(iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / 3
            r["streaming_ms_per_step"] = round(dts * 1e3, 3)
            r["streaming_MS_per_s"] = round(n_bytes / 2 / dts / 1e6, 1)
            r["streaming_x_realtime"] = round(n_bytes / 2 / dts / p.rf_Fs, 1)
            r["pll_repaired_segments"] = pl.pll_diagnostics()[0]
        print(json.dumps(r), flush=True)
        res.append(r)
        del pl, iq, d_a, d_p
        torch.cuda.empty_cache()

# host-buffer entry point (PCIe inclusive), mode 0 mono, one 1,024,000-sample block per call
blk = synth.synth_fm_u8(1_024_000)
pl = fmrx.Pipeline(0, 1, max_block_bytes=len(blk))
pl.process(blk)
t0 = time.perf_counter()
for _ in range(20):
    pl.process(blk)
dt = (time.perf_counter() - t0) / 20
print(json.dumps({"host_buffers_mode0_mono": {"ms_per_1M_block": round(dt * 1e3, 3), "MS_per_s": round(1.024 / dt, 1)}}), flush=True)
# reference-size blocks (51,200 samples = 21 ms of signal): latency per block
small = blk[:102400]
pl2 = fmrx.Pipeline(0, 1)
pl2.process(small)
t0 = time.perf_counter()
for _ in range(200):
    pl2.process(small)
dt = (time.perf_counter() - t0) / 200
print(json.dumps({"host_buffers_reference_block": {"us_per_block": round(dt * 1e6, 1), "x_realtime": round(0.0213333 / dt, 1)}}), flush=True)
# CLI, stdin -> stdout through /dev/shm
path = "/dev/shm/fmrx_cli_in.u8"
np.tile(blk, 48).tofile(path)      # ~49 M samples, 98 MB
exe = os.path.join(os.path.dirname(fmrx.LIB_PATH), "fmrx_project")
for per_call in (1, 20):
    t0 = time.perf_counter()
    with open(path, "rb") as f:
        r = subprocess.run([exe, "0", "1", "--blocks-per-call", str(per_call)], stdin=f, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    print(json.dumps({"cli_mode0_mono": {"blocks_per_call": per_call, "seconds": round(dt, 3), "MS_per_s": round(48 * 1.024 / dt, 1), "rc": r.returncode}}), flush=True)
os.remove(path)
