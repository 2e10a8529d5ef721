#!/usr/bin/env python3
"""The live regime, one 51,200-sample block per call: is a call bound by the host's launches or by the device?  Host time to
enqueue N calls (no sync) vs time until the device has finished them."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
d_iq = torch.from_numpy(synth.synth_fm_u8(51200, 2.4e6, seed=1)).cuda()
q = fmrx.Pipeline(0, 1)
d_pcm = torch.empty(1024, dtype=torch.int16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def run(n):
    for _ in range(n):
        q.process_dev(d_iq.data_ptr(), 102400, None, d_pcm.data_ptr(), wrap=True, stream=s)
run(500); torch.cuda.synchronize()
N = 5000
t0 = time.perf_counter(); run(N); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{N} calls: host enqueue {1e6 * (t1 - t0) / N:.2f} us per call, until the device is done {1e6 * (t2 - t0) / N:.2f} us per call")
