#!/usr/bin/env python3
"""Stereo, one long call (nb blocks of 1,024,000 samples = nb*0.43 s of signal) against the oracle: error over time,
PLL repairs, device time."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
from _oracle import Oracle
o = Oracle()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = nb * 1_024_000
iq = synth.synth_fm_u8(n, 2.4e6, seed=0x3D74)
t0 = time.perf_counter(); ref = o.pipeline(0, 2).process(iq); t_cpu = time.perf_counter() - t0
pl = fmrx.Pipeline(0, 2, max_block_bytes=2 * n)
if len(sys.argv) > 2 and sys.argv[2] == "generic":
    pl.set_force_generic(True)      # serial PLL, reference evaluation order, device libm
out = pl.process(iq)
pl.reset(); pl.set_profiling(True)
if len(sys.argv) > 2 and sys.argv[2] == 'generic': pl.set_force_generic(True)
t0 = time.perf_counter(); out = pl.process(iq); t_gpu = time.perf_counter() - t0
tm = pl.last_timing()
def rms(x): return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))
for k in ("audio_l", "audio_r"):
    e = out[k].astype(np.float64) - ref[k]
    seg = len(e) // 6
    print(k, "rms err total", f"{rms(e):.2e}", "by sixth of the stream:", " ".join(f"{rms(e[i*seg:(i+1)*seg]):.1e}" for i in range(6)), "signal rms", f"{rms(ref[k]):.3f}")
print(f"{nb} blocks = {n/2.4e6:.1f} s of signal: device time {tm}, host call {t_gpu*1e3:.1f} ms (PCIe incl.), oracle on 1 core {t_cpu:.1f} s; pll diag {pl.pll_diagnostics()}")
