#!/usr/bin/env python3
"""Per-step GPU time series (torch events) of the bench workload, to see clock / placement effects."""
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
st = torch.cuda.current_stream()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
ev[0].record(st)
for i in range(K):
    pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr(), d_p.data_ptr(), stream=st.cuda_stream)
    ev[i + 1].record(st)
torch.cuda.synchronize()
t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(K)])
print("steps", K, "median", np.median(t), "min", t.min(), "max", t.max())
print("series (x10 avg):", " ".join(f"{t[i:i+10].mean():.3f}" for i in range(0, K, 10)))
