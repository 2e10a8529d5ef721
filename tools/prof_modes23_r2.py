#!/usr/bin/env python3
"""Profiling target: modes 2 and 3 mono, 63 reference blocks per step, 6 steps each, s16 out.
rocprofv3 --kernel-trace --stats -- python3 tools/prof_modes23_r2.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
for mode in (2, 3):
    bb = 2_016_000                                   # 10 reference blocks of modes 2 / 3 (100,800 samples each)
    fs = 2.4e6 if mode == 2 else 0.96e6
    iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 2, fs, seed=0x3D74)).cuda().repeat(42)
    nb = iq.numel()
    pl = fmrx.Pipeline(mode, 1, max_block_bytes=nb)
    d_pcm = torch.empty(pl.n_audio(nb), dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(6):
        pl.process_dev(iq.data_ptr(), nb, None, d_pcm.data_ptr(), stream=s)
    torch.cuda.synchronize()
    print("mode", mode, nb // 2, "samples per step")
