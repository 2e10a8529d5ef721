#!/usr/bin/env python3
"""A/B variants of the fused front-end kernel inside ONE process, interleaved
rounds, on the bench workload: env switches read per launch (FMRX_FE_R=8|12).
    python tools/fe_ab2.py [blocks=256] [rounds=6]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N = 1_024_000
d_iq = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda().repeat(B // 4)
n_bytes = d_iq.numel()
pl = fmrx.Pipeline(0, 1, max_block_bytes=n_bytes)
na = pl.n_audio(n_bytes)
d_a = torch.empty(na, dtype=torch.float32, device="cuda")
d_p = torch.empty(na, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
variants = {"R8": {"FMRX_FE_R": "8"}, "R12": {"FMRX_FE_R": "12"}}
res, outs = {}, {}
for rnd in range(ROUNDS):
    for name, env in variants.items():
        os.environ.update(env)
        pl.reset()
        pl.set_profiling(False)
        for _ in range(3):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        pl.set_profiling(True)
        for _ in range(10):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        t, c = pl.timing_sum(10)
        res.setdefault(name, []).append(t["front_end_ms"] / c)
        outs[name] = d_a.clone()
print("outputs identical:", bool(torch.equal(outs["R8"], outs["R12"])))
n = n_bytes // 2
for name, ts in res.items():
    ms = float(np.median(ts))
    print(f"{name}: median {ms:.4f} ms min {min(ts):.4f} max {max(ts):.4f} -> {2.4 * n / ms / 1e6:.0f} GB/s = {2.4 * n / ms / 1e6 / 8000:.3f} of HBM peak")
