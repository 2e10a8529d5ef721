#!/usr/bin/env python3
"""Profiling target: the bench workload (mode 0 mono, 256 x 1,024,000-sample
blocks resident) for a few steps, no CPU baseline, no torch.distributed.
Run under rocprofv3 as `rocprofv3 ... -- python3 tools/prof_target.py [steps] [blocks]`."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
N = 1_024_000
d_iq = torch.from_numpy(synth.synth_fm_u8(4 * N)).cuda().repeat(B // 4)
n_bytes = d_iq.numel()
pl = fmrx.Pipeline(0, 1, max_block_bytes=n_bytes)
na = pl.n_audio(n_bytes)
d_audio = torch.empty(na, dtype=torch.float32, device="cuda")
d_pcm = torch.empty(na, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(steps):
    pl.process_dev(d_iq.data_ptr(), n_bytes, d_audio.data_ptr(), d_pcm.data_ptr(), stream=s)
torch.cuda.synchronize()
print("done", steps, "steps of", n_bytes // 2, "samples")
