#!/usr/bin/env python3
"""A/B variants of the fused mono kernel (option fused_tune), one process, interleaved rounds, after a settle phase.
Needs the tuning build of the library (ablation kernels are not in the shipped libfmrx.so):
    make -C software-defined-radio_amd/csrc TUNING=1 && FMRX_LIB=software-defined-radio_amd/lib/libfmrx_tuning.so python tools/fused_tune.py"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
MODE = int(os.environ.get("FUSED_TUNE_MODE", "0"))   # 1: mode 1 (variants 5000 + DBG)
B, N = 256, (1_024_000 if MODE == 0 else 614_400)
d_iq = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda().repeat(B // 4)
n_bytes = d_iq.numel()
pl = fmrx.Pipeline(MODE, 1, max_block_bytes=n_bytes)
na = pl.n_audio(n_bytes)
d_a = torch.empty(na, dtype=torch.float32, device="cuda"); d_p = torch.empty(na, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(k):
    global F32
    for _ in range(k):
        pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr() if F32 else 0, d_p.data_ptr(), stream=s)
    torch.cuda.synchronize()
F32 = os.environ.get("FUSED_TUNE_F32", "0") == "1"   # also write the f32 audio (round 1's output format)
variants = sys.argv[1:] or ["2", "12", "3", "13", "4"]
run(3000)   # settle
res = {}
for rnd in range(5):
    for v in variants:
        pl.set_option("fused_tune", int(v))
        run(20)
        t0 = time.perf_counter(); run(200); dt = (time.perf_counter() - t0) / 200 * 1e3
        res.setdefault(v, []).append(dt)
for v, ts in res.items():
    print(f"fused tune {v} (DBG,P): median {np.median(ts):.4f} ms min {min(ts):.4f} max {max(ts):.4f}", flush=True)
