#!/usr/bin/env python3
"""Fused mono kernel: achieved TB/s vs the number of 1,024,000-sample blocks per call, i.e. vs the byte distance between the
2 048 waves' concurrently streamed runs (blocks x 1000 B).  Distances near multiples of 128 KiB are the slow ones
(tools/stream_patterns.py shows the same for a read-only kernel)."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
N = 1_024_000
sizes = [int(a) for a in sys.argv[1:]] or [224, 240, 248, 256, 264, 272, 288]
base = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda()
d_all = base.repeat(max(sizes) // 4 + 1)
s = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for B in sizes:
        n_bytes = 2 * N * B
        pl = fmrx.Pipeline(0, 1, max_block_bytes=n_bytes)
        d_p = torch.empty(pl.n_audio(n_bytes), dtype=torch.int16, device="cuda")
        def run(k):
            for _ in range(k):
                pl.process_dev(d_all.data_ptr(), n_bytes, 0, d_p.data_ptr(), stream=s)
            torch.cuda.synchronize()
        run(300 if rnd == 0 else 30)
        t0 = time.perf_counter(); run(200); dt = (time.perf_counter() - t0) / 200
        res.setdefault(B, []).append(2.04 * N * B / dt / 1e12)
        pl.close()
for B, v in res.items():
    print(f"{B} blocks, {B * 1000} B between neighbouring waves' runs: {np.median(v):.3f} TB/s (min {min(v):.3f} max {max(v):.3f})", flush=True)
