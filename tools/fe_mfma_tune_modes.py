#!/usr/bin/env python3
"""A/B of fe_mfma_tune variants on the modes whose front end decimates by 5 / 3 (tuning build; see fe_mfma_tune.py).
    FMRX_LIB=.../libfmrx_tuning.so python tools/fe_mfma_tune_modes.py 0 1600"""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
variants = sys.argv[1:] or ["0", "1600"]
for mode, fs, bb in ((3, 0.96e6, 2_016_000), (2, 2.4e6, 2_016_000), (1, 1.44e6, 1_228_800)):
    iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 4, fs, seed=0x3D74)).cuda().repeat(42)
    nb = iq.numel()
    pl = fmrx.Pipeline(mode, 1, max_block_bytes=nb)
    if mode == 1:
        pl.set_option("fused_min_audio", 10**12)   # the S2 kernel, not the fused one
    d_pcm = torch.empty(pl.n_audio(nb), dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    def run(k):
        for _ in range(k):
            pl.process_dev(iq.data_ptr(), nb, None, d_pcm.data_ptr(), stream=s)
        torch.cuda.synchronize()
    run(300)
    res = {}
    for rnd in range(5):
        for v in variants:
            pl.set_option("fe_mfma_tune", int(v))
            run(10)
            t0 = time.perf_counter(); run(100); res.setdefault(v, []).append((time.perf_counter() - t0) * 10)
    for v, ts in res.items():
        print(f"mode {mode}, {nb // 2} samples per step, whole step, fe_mfma_tune {v}: median {np.median(ts):.4f} ms min {min(ts):.4f}", flush=True)
