#!/usr/bin/env python3
"""Does the headline step profit from the 256 MiB Infinity Cache?  The fused mono kernel and the pure streaming reads
(fmrx_diag_stream_read_dev) at 256, 512, 1024 and 2048 resident blocks of 1,024,000 samples (0.5 .. 4.2 GB)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
base = torch.from_numpy(synth.synth_fm_u8(4 * 1024000)).cuda()
stream = torch.cuda.current_stream().cuda_stream
def ev(fn, k, warm):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for B in (128, 256, 512, 1024, 2048):
    d = base.repeat(B // 4)
    nb = d.numel()
    pl = fmrx.Pipeline(0, 1, max_block_bytes=nb)
    pcm = torch.empty(pl.n_audio(nb), dtype=torch.int16, device="cuda")
    # settle the clocks first
    for _ in range(max(50, 4000 // B)): pl.process_dev(d.data_ptr(), nb, None, pcm.data_ptr(), stream=stream)
    ms = ev(lambda: pl.process_dev(d.data_ptr(), nb, None, pcm.data_ptr(), stream=stream), max(10, 2000 // B), 5)
    line = f"blocks {B:5d} ({nb/1e9:5.2f} GB): fused mono {ms:8.4f} ms  {nb/2/ms/1e3:10.1f} MS/s  frac {nb/2*2.04/(ms*1e-3)/8e12:6.4f}"
    for method, label in ((0, "nt loads"), (1, "LDS-DMA ring")):
        r = ev(lambda: fmrx.diagStreamRead(d.data_ptr(), nb, method, stream), max(10, 2000 // B), 10)
        line += f" | {label} {nb/(r*1e-3)/1e12:5.2f} TB/s"
    print(line, flush=True)
    pl.close(); del d, pcm
    torch.cuda.empty_cache()
