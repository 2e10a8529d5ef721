#!/usr/bin/env python3
"""Profiling target: mode 0 stereo, 12 x 1,024,000-sample blocks per step, fresh stream: 1 first call + 4 streaming
steps, s16 out.  rocprofv3 --kernel-trace --stats -- python3 tools/prof_stereo_r2.py [W] [L] [align]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
bb = 2048000
iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 2, 2.4e6, seed=0x3D74)).cuda().repeat(4)
nb = iq.numel()
pl = fmrx.Pipeline(0, 2, max_block_bytes=nb)
for k, i in (("pll_warmup", 1), ("pll_segment", 2), ("pll_align", 3)):
    if len(sys.argv) > i:
        pl.set_option(k, int(sys.argv[i]))
d_pcm = torch.empty(2 * pl.n_audio(nb), dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    pl.process_dev(iq.data_ptr(), nb, None, d_pcm.data_ptr(), stream=s)
torch.cuda.synchronize()
print("done", pl.pll_diagnostics())
