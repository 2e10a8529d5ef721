#!/usr/bin/env python3
"""Profiling target: one mono mode's bench-leg workload, device-resident, a few steps.
  rocprofv3 ... -- python3 tools/prof_mode.py <mode> [blocks] [steps]
mode 0: blocks x 1,024,000 samples; mode 1: blocks x 614,400; modes 2 / 3: blocks x 1,008,000 (bench.py's leg shapes)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
unit = {0: 2048000, 1: 1228800, 2: 2016000, 3: 2016000}[mode]
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else {0: 256, 1: 256, 2: 63, 3: 63}[mode]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
p = fmrx.modeParams(mode)
iq = synth.synth_fm_u8(unit // 2, float(p.rf_Fs), seed=0x3D74 + 10 + mode)
d_in = torch.from_numpy(iq).cuda().repeat(blocks)
nb = d_in.numel()
pl = fmrx.Pipeline(mode, 1, max_block_bytes=nb)
d_pcm = torch.empty(pl.n_audio(nb), dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(steps):
    pl.process_dev(d_in.data_ptr(), nb, None, d_pcm.data_ptr(), wrap=True, stream=s)
torch.cuda.synchronize()
print("done mode", mode, steps, "steps of", nb // 2, "samples")
