#!/usr/bin/env python3
"""tests/golden/arctan.npz from the REFERENCE's own Python model (build container only).

fmDemodArctan (model/fmSupportLib.py:502-531) is imported where it lies (python -B: /root/reference is read-only) and run,
block by block with its state, on (a) the IF stream of three reference-size blocks of the synthetic FM signal (the IF samples
are the oracle's, i.e. the C++ front end's float32 output) and (b) a crafted sequence that exercises np.unwrap's branches
(steps near +-pi, more than pi, zero vectors).  Only data is written: no reference source.

    python3 -B tests/golden/make_golden_arctan.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.environ.get("FMRX_REFERENCE", "/root/reference") + "/model")
import fmSupportLib as L  # noqa: E402  the reference's model
from _oracle import Oracle  # noqa: E402

o = Oracle()
NBLK = 3
p = o.mode_params(0, 101, 101, 101)
iq = o.synth_fm_u8(p.block_bytes // 2 * NBLK, rf_Fs=p.rf_Fs, seed=0x3D74 + 31)
pl = o.pipeline(0, 1)
out = {"seed": np.array([0x3D74 + 31]), "nblk": np.array([NBLK])}
phase = 0.0
for b in range(NBLK):
    r = pl.process(iq[b * p.block_bytes:(b + 1) * p.block_bytes])
    d, phase = L.fmDemodArctan(r["if_i"].astype(np.float64), r["if_q"].astype(np.float64), phase)
    out[f"b{b}_if_i"], out[f"b{b}_if_q"] = r["if_i"], r["if_q"]
    out[f"b{b}_demod"] = d
    out[f"b{b}_phase"] = np.array([phase])
# np.unwrap's branches: angles walked in steps of 0.9 pi .. 1.1 pi (both directions), a zero vector, tiny vectors
rng = np.random.default_rng(17)
steps = np.concatenate([rng.uniform(-1.1 * np.pi, 1.1 * np.pi, 400), np.full(8, 0.999 * np.pi), np.full(8, -0.999 * np.pi),
                        rng.uniform(-0.2, 0.2, 100)])
ang = np.cumsum(steps)
mag = rng.uniform(0.05, 1.0, len(ang))
eI, eQ = mag * np.cos(ang), mag * np.sin(ang)
eI[50], eQ[50] = 0.0, 0.0
eI[51], eQ[51] = 0.0, -0.0
d, ph = L.fmDemodArctan(eI, eQ, 0.3)
out["edge_i"], out["edge_q"], out["edge_demod"], out["edge_phase"], out["edge_prev"] = eI, eQ, d, np.array([ph]), np.array([0.3])
np.savez_compressed(os.path.join(HERE, "arctan.npz"), **out)
print("wrote arctan.npz:", {k: v.shape for k, v in out.items()})
