#!/usr/bin/env python3
"""Profiling target: fmrx_channels, 4096 mode-0 mono channels, one 51,200-sample block each per call, 30 calls.
rocprofv3 --kernel-trace --stats -- python3 tools/prof_channels_r2.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
nch = 4096
chs = fmrx.Channels(0, nch)
blk = torch.from_numpy(synth.synth_fm_u8(51200, 2.4e6, seed=0x3D74 + 77)).cuda()
src = blk.repeat(nch)
s = torch.cuda.current_stream().cuda_stream
chs.load_dev(src.data_ptr(), s)
d_pcm = torch.empty(nch * chs.n_audio, dtype=torch.int16, device="cuda")
for _ in range(30):
    chs.process_dev(None, d_pcm.data_ptr(), wrap=True, stream=s)
torch.cuda.synchronize()
print("done")
