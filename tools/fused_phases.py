#!/usr/bin/env python3
"""Where a wave of the fused mono kernel spends its time, phase by phase (tuning build, option fused_tune=162: s_memtime stamps).
    make -C software-defined-radio_amd/csrc TUNING=1 && FMRX_LIB=software-defined-radio_amd/lib/libfmrx_tuning.so python tools/fused_phases.py"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
MODE = int(os.environ.get("FUSED_TUNE_MODE", "0"))   # 1: mode 1 (variant 5016)
B, N = 256, (1_024_000 if MODE == 0 else 614_400)
d_iq = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda().repeat(B // 4)
n_bytes = d_iq.numel()
pl = fmrx.Pipeline(MODE, 1, max_block_bytes=n_bytes)
na = pl.n_audio(n_bytes)
d_a = torch.zeros(na, dtype=torch.float32, device="cuda"); d_p = torch.empty(na, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(k, audio=True):
    for _ in range(k):
        pl.process_dev(d_iq.data_ptr(), n_bytes, d_a.data_ptr() if audio else 0, d_p.data_ptr(), stream=s)
    torch.cuda.synchronize()
run(2000, audio=False)
t0 = time.perf_counter(); run(100, audio=False); base_ms = (time.perf_counter() - t0) * 10
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 162
pl.set_option("fused_tune", variant)
run(20)
t0 = time.perf_counter(); run(100); inst_ms = (time.perf_counter() - t0) * 10
d_a.zero_(); run(1)
rec = d_a.cpu().numpy().view(np.int64)
n_waves = 2048
rec = rec[: 8 * n_waves].reshape(n_waves, 8)
rec = rec[rec[:, 6] > 0]
tiles = rec[:, 6].astype(float)
names = ["dma issue + wait for the tile's bytes", "B fragments from LDS (+ slice operands)", "byte flip + 18 int8 MFMAs + unpack",
         "discriminator (shuffle, atan2 form) + ring write", "audio slice MFMAs issued + bookkeeping"]
print(f"kernel per launch: plain {base_ms:.4f} ms, instrumented {inst_ms:.4f} ms; {len(rec)} waves, {tiles.mean():.1f} tiles each")
span = rec[:, 5].astype(float)
print(f"wave run time (clock ticks): mean {span.mean():.0f} min {span.min():.0f} max {span.max():.0f}")
rt = rec[:, 7].astype(float)
print(f"  shader clock during the run (s_memtime / s_memrealtime x 100 MHz): median {np.median(span / rt) * 100:.0f} MHz; run = {np.median(rt) / 100:.1f} us")
tot = 0.0
for k, nme in enumerate(names):
    per = rec[:, k] / tiles
    tot += per.mean()
    print(f"  phase {k}: {per.mean():8.1f} ticks/tile (p10 {np.percentile(per,10):.1f}, p90 {np.percentile(per,90):.1f})  {nme}")
print(f"  sum {tot:.1f} ticks/tile; whole run / tiles {np.mean(span / tiles):.1f}")
