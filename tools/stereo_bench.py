#!/usr/bin/env python3
"""Stereo step time vs the lane shape of the parallel PLL: mode 0 stereo, 12 x 1,024,000-sample blocks per step
(a seamless stream: 3 synthesised blocks = 1280 periods of the multiplex, tiled), s16 L,R out, stream continued.
    python tools/stereo_bench.py [blocks=12]   -> one line per (warmup, segment, align)"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bb = 2048000
iq = torch.from_numpy(synth.synth_fm_u8(3 * bb // 2, 2.4e6, seed=0x3D74)).cuda().repeat(blocks // 3)
nb = iq.numel()
s = torch.cuda.current_stream().cuda_stream
# (warm-up, segment, align, start): start 1 = lanes start from the locked loop as a linear system of the input's signs
cfgs = [(512, 64, 0, 0), (768, 64, 0, 0), (256, 64, 1, 0), (0, 64, 0, 1), (64, 64, 0, 1), (128, 64, 0, 1), (64, 128, 0, 1), (256, 64, 0, 1)]
for W, L, A, S in cfgs:
    pl = fmrx.Pipeline(0, 2, max_block_bytes=nb)
    for k, v in (("pll_warmup", W), ("pll_segment", L), ("pll_align", A), ("pll_start", S)):
        pl.set_option(k, v)
    na = pl.n_audio(nb)
    d_pcm = torch.empty(2 * na, dtype=torch.int16, device="cuda")
    fn = lambda: pl.process_dev(iq.data_ptr(), nb, None, d_pcm.data_ptr(), stream=s)
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):          # 6 steps = 7.4 M IF samples = 31 s of stream: below the 2^24 where trigOffset stops
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    rep, dp, di = pl.pll_diagnostics()
    print(f"W={W:4d} L={L:4d} align={A} start={S}: {ms*1e3:8.1f} us per {nb//2} samples = {nb/2/ms/1e3:9.0f} MS/s = {2.08*nb/2/ms/1e6/8000:.4f} of HBM peak; "
          f"repaired segments {rep}, max accepted dphase {dp:.2e} dinteg {di:.2e}", flush=True)
    pl.close()
