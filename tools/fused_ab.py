#!/usr/bin/env python3
"""A/B fused mono kernel vs front end + audio kernels, one process, interleaved rounds; wall time per step."""
import importlib, os, sys, time
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
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
res = {}
for rnd in range(4):
    for name, v in (("fused", "0"), ("split", "1000000000000")):
        os.environ["FMRX_FUSED_MIN_AUDIO"] = v
        for _ in range(5):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=s)
        torch.cuda.synchronize()
        res.setdefault(name, []).append((time.perf_counter() - t0) / K * 1e3)
for name, ts in res.items():
    print(f"{name}: ms/step median {np.median(ts):.4f} min {min(ts):.4f} max {max(ts):.4f}", flush=True)
