#!/usr/bin/env python3
"""Modes 2 / 3 mono step time: matrix-core resampler vs the bit-exact LDS-table kernel (resample_exact 1); 63.5 M-sample
steps, s16 out, interleaved rounds.  (A third variant was measured with this script and dropped: sixteen 4x4 blocks per
instruction, v_mfma_f32_4x4x1_16b_f32, every block of 4 outputs with its own K origin: K 118 / 123 instead of 192 / 224, but
2.2 x the tap traffic from L2 per output: 0.0454 vs 0.0403 ms in mode 2, 0.1061 vs 0.0864 ms in mode 3.)"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
bb = 2_016_000
for mode, fs in ((2, 2.4e6), (3, 0.96e6)):
    iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 4, fs, seed=0x3D74)).cuda().repeat(42)
    nb = iq.numel()
    pl = fmrx.Pipeline(mode, 1, max_block_bytes=nb)
    d_pcm = torch.empty(pl.n_audio(nb), dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    def run(k):
        for _ in range(k):
            pl.process_dev(iq.data_ptr(), nb, None, d_pcm.data_ptr(), stream=s)
        torch.cuda.synchronize()
    run(300)
    res = {}
    for rnd in range(4):
        for name, opts in (("matrix cores (16x16 tiles)", {"resample_exact": 0}), ("bit-exact LDS table", {"resample_exact": 1})):
            for k, v in opts.items():
                pl.set_option(k, v)
            run(10)
            t0 = time.perf_counter(); run(100); res.setdefault(name, []).append((time.perf_counter() - t0) * 10)
    for name, ts in res.items():
        print(f"mode {mode}, {nb // 2} samples per step, resampler {name:26s}: median {np.median(ts):.4f} ms min {min(ts):.4f}", flush=True)
