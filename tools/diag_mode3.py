#!/usr/bin/env python3
"""Diagnose mode-3 stereo: front-end variants vs oracle on a large block, PLL repair counts."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
fmrx = importlib.import_module("software-defined-radio_amd")
synth = importlib.import_module("software-defined-radio_amd.synth")
from _oracle import Oracle
o = Oracle()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = fmrx.modeParams(mode)
per_blk = {0: 20, 1: 20, 2: 18, 3: 15}[mode] * p.block_bytes
nb = 12
iq = synth.synth_fm_u8(per_blk // 2 * 3, p.rf_Fs, seed=0x3D74 + mode)
iq = np.tile(iq, nb // 3)
po = o.pipeline(mode, 1)
ref = po.process(iq)
for var in ("mfma", "valu"):
    os.environ["FMRX_FE_VARIANT"] = var
    pl = fmrx.Pipeline(mode, 1, max_block_bytes=len(iq)); pl.set_keep_intermediates(True)
    out = pl.process(iq)
    d = pl.read_tap("demod")
    e = np.abs(d.astype(np.float64) - ref["demod"])
    print(var, "mono demod max err", e.max(), "at", int(e.argmax()), "rms", np.sqrt((e**2).mean()), "n", len(d))
    i_ = pl.read_tap("if_i"); ei = np.abs(i_.astype(np.float64) - ref["if_i"]); print(var, "if_i max err", ei.max(), "at", int(ei.argmax()))
    ps = fmrx.Pipeline(mode, 2, max_block_bytes=len(iq))
    t0 = time.perf_counter(); ps.process(iq); t1 = time.perf_counter(); ps.reset(); ps.process(iq); t2 = time.perf_counter()
    print(var, "stereo process s", round(t1 - t0, 4), round(t2 - t1, 4), "pll diag", ps.pll_diagnostics())
