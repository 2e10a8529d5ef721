#!/usr/bin/env python3
"""Stereo step time by option overlap_calls (0 = one stream; 1 = the front of the next call on an internal stream under the PLL
and output stage of this one; 2 = front / PLL / output stage on three internal streams): mode 0 stereo, `blocks` x 1,024,000-sample blocks per step of a seamless stream, s16 L,R out, a fresh
stream's first 31 s per measurement; the PCM of both forms is compared call by call.
    python tools/stereo_overlap_bench.py [blocks=12 ...]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
bb = 2048000
base = torch.from_numpy(synth.synth_fm_u8(3 * bb // 2, 2.4e6, seed=0x3D74)).cuda()
s = torch.cuda.current_stream().cuda_stream
for blocks in [int(a) for a in sys.argv[1:]] or [12, 24]:
    iq = base.repeat(blocks // 3)
    nb = iq.numel()
    n_if = nb // 2 // 10
    steps = max(2, min(8, 7_500_000 // n_if - 2))
    outs = {}
    for ovl in (0, 1, 2):
        pl = fmrx.Pipeline(0, 2, max_block_bytes=nb)
        pl.set_option("overlap_calls", ovl)
        na = pl.n_audio(nb)
        d_pcm = [torch.empty(2 * na, dtype=torch.int16, device="cuda") for _ in range(steps + 2)]
        best = None
        for rnd in range(3):
            pl.reset()
            torch.cuda.synchronize()
            pl.process_dev(iq.data_ptr(), nb, None, d_pcm[0].data_ptr(), stream=s)
            pl.process_dev(iq.data_ptr(), nb, None, d_pcm[1].data_ptr(), stream=s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for k in range(steps):
                pl.process_dev(iq.data_ptr(), nb, None, d_pcm[2 + k].data_ptr(), stream=s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            best = ms if best is None else min(best, ms)
        rep = pl.pll_diagnostics()[0]
        outs[ovl] = [d.clone() for d in d_pcm]
        print(f"{blocks} blocks per step, overlap_calls {ovl}: {best*1e3:7.1f} us per step = {nb/2/best/1e3:9.0f} MS/s = "
              f"{2.08*nb/2/best/1e6/8000:.4f} of HBM peak; repaired segments {rep}", flush=True)
        pl.close()
    same = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(outs[0], outs[1], outs[2]))
    print(f"{blocks} blocks per step: PCM of the three forms {'identical' if same else 'DIFFERENT'} over {steps + 2} calls", flush=True)
