#!/usr/bin/env python3
"""A/B the matrix-core front end over (workgroups per CU, tiles in flight), one process, interleaved rounds.
Needs the tuning build of the library (ablation kernels are not in the shipped libfmrx.so):
    make -C software-defined-radio_amd/csrc TUNING=1 && FMRX_LIB=software-defined-radio_amd/lib/libfmrx_tuning.so python tools/fe_mfma_tune.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
B, N = 256, 1_024_000
d_iq = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda().repeat(B // 4)
n_bytes = d_iq.numel()
pl = fmrx.Pipeline(0, 1, max_block_bytes=n_bytes)
na = pl.n_audio(n_bytes)
d_a = torch.empty(na, dtype=torch.float32, device="cuda"); d_p = torch.empty(na, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
pl.set_option("fused_min_audio", 10**12)   # the S2 kernel, not the fused one
variants = sys.argv[1:] or ["23", "22", "24", "33", "43", "13"]
res = {}
for rnd in range(5):
    for v in variants:
        pl.set_option("fe_mfma_tune", int(v))
        pl.set_profiling(False)
        for _ in range(3):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        pl.set_profiling(True)
        for _ in range(20):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        t, c = pl.timing_sum(20)
        res.setdefault(v, []).append(t["front_end_ms"] / c)
for v, ts in res.items():
    print(f"tune {v} (WG/CU, P): median {np.median(ts):.4f} ms min {min(ts):.4f} max {max(ts):.4f}", flush=True)
