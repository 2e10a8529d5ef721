#!/usr/bin/env python3
"""Modes 2 / 3 mono step time by the matrix-core resampler's chain count (option resample_chains): how many workgroups per XCD
and tile group walk the period blocks; 0 = as many as are resident at once (default), a huge count = one block per workgroup."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
bb = 2_016_000
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 42
for mode, fs in ((2, 2.4e6), (3, 0.96e6)):
    iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 4, fs, seed=0x3D74)).cuda().repeat(REP)
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
    ref = None
    for rnd in range(4):
        for ch in ("1000000", "", "8", "16", "24", "32", "64", "96"):
            pl.set_option("resample_chains", int(ch or 0))
            run(10)
            if rnd == 0:
                got = d_pcm.clone()
                if ref is None:
                    ref = got
                elif not torch.equal(ref, got):
                    print(f"mode {mode} chains {ch}: output differs from the one-block-per-workgroup launch", flush=True)
            t0 = time.perf_counter(); run(100); res.setdefault(ch or "auto", []).append((time.perf_counter() - t0) * 10)
    for name, ts in res.items():
        print(f"mode {mode}, {nb // 2} samples per step, chains per XCD {name:>8s}: median {np.median(ts):.4f} ms min {min(ts):.4f}", flush=True)
